// Detector pre/post-processing kernels.
//   preprocess : letterbox (tinygrad interpolate/lerp semantics, incl. the uint8 7-bit fixed point),
//                zero pad, BGR->RGB, /255            detection/yolov9.py:376-379,390-404; helpers.py:127-131
//   decode     : DFL softmax-expectation, dist2bbox, *stride, sigmoid, xywh->xyxy, class max/argmax,
//                confidence threshold                detection/yolov9.py:209-220,263-282,440-448
//   topk_nms   : stable top-300 by score (radix select + bitonic sort), 300x300 mask NMS,
//                scale_boxes/clip_boxes              detection/yolov9.py:406-458
// Integer/byte work (the uint8 resize) is bit-exact against the oracle; float work follows the
// reference's operation order in f32.
#include "kernels.h"

namespace cc {

// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int lerp_u8(int a, int b, int w7) {
  // tinygrad Tensor.lerp for uint8: int8-wrapped difference, 7-bit weight, +64, >>7, mod 256.
  const int d = (int)(int8_t)(uint8_t)(b - a);
  const int t = (int)(((uint16_t)(int16_t)(d * w7 + 64)) >> 7);
  return (a + t) & 0xff;
}

template <class T>
__global__ __launch_bounds__(256) void preprocess_kernel(const PreP p) {
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, b = blockIdx.z;      // grid = (row tiles, row, image)
  if (x >= p.Wn) return;
  const size_t idx = ((size_t)b * p.Hn + y) * p.Wn + x;
  float rgb[3] = {p.pad_val, p.pad_val, p.pad_val};
  const int yy = y - p.pad_y, xx = x - p.pad_x;
  if ((unsigned)yy < (unsigned)p.nh && (unsigned)xx < (unsigned)p.nw) {
    const int x0 = p.xlo[xx], x1 = p.xhi[xx], y0 = p.ylo[yy], y1 = p.yhi[yy];
    const float fx = p.xfr[xx], fy = p.yfr[yy];
    const size_t r0 = ((size_t)b * p.H + y0) * p.W, r1 = ((size_t)b * p.H + y1) * p.W;
    if (!p.frame_f32) {
      const uint8_t* f = reinterpret_cast<const uint8_t*>(p.frames);
      const int wx = (int)(int16_t)__fadd_rn(__fmul_rn(fx, 128.0f), 0.5f);
      const int wy = (int)(int16_t)__fadd_rn(__fmul_rn(fy, 128.0f), 0.5f);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int top = lerp_u8(f[(r0 + x0) * 3 + c], f[(r0 + x1) * 3 + c], wx);
        const int bot = lerp_u8(f[(r1 + x0) * 3 + c], f[(r1 + x1) * 3 + c], wx);
        rgb[p.flip ? 2 - c : c] = __fsub_rn(__fdiv_rn((float)lerp_u8(top, bot, wy), p.div), p.sub);
      }
    } else {
      const float* f = reinterpret_cast<const float*>(p.frames);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float a0 = f[(r0 + x0) * 3 + c], b0 = f[(r0 + x1) * 3 + c];
        const float a1 = f[(r1 + x0) * 3 + c], b1 = f[(r1 + x1) * 3 + c];
        const float top = __fadd_rn(a0, __fmul_rn(__fsub_rn(b0, a0), fx));
        const float bot = __fadd_rn(a1, __fmul_rn(__fsub_rn(b1, a1), fx));
        rgb[p.flip ? 2 - c : c] = __fsub_rn(__fdiv_rn(__fadd_rn(top, __fmul_rn(__fsub_rn(bot, top), fy)), p.div), p.sub);
      }
    }
  }
  T* o = reinterpret_cast<T*>(p.out) + idx * p.out_c;
  if (p.out_c * (int)sizeof(T) == 16) {                         // the stem's one 16-byte channel chunk: a single store
    uint4 v = make_uint4(0, 0, 0, 0);
    T* t = reinterpret_cast<T*>(&v);
#pragma unroll
    for (int c = 0; c < 3; ++c) t[c] = from_f32<T>(rgb[c]);
    *reinterpret_cast<uint4*>(o) = v;
  } else {
    for (int c = 0; c < p.out_c; ++c) o[c] = from_f32<T>(c < 3 ? rgb[c] : 0.f);
  }
}

void launch_preprocess(int dt, const PreP& p, hipStream_t stream) {
  const dim3 grid((unsigned)((p.Wn + 255) / 256), (unsigned)p.Hn, (unsigned)p.B), block(256);
  if (dt == F32) hipLaunchKernelGGL(preprocess_kernel<float>, grid, block, 0, stream, p);
  else if (dt == F16) hipLaunchKernelGGL(preprocess_kernel<f16_t>, grid, block, 0, stream, p);
  else hipLaunchKernelGGL(preprocess_kernel<bf16_t>, grid, block, 0, stream, p);
  CC_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void decode_kernel(const DecodeP p) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)p.B * p.A) return;
  const int b = (int)(idx / p.A);
  int a = (int)(idx - (size_t)b * p.A);
  int lvl = 0;
  while (lvl < 2 && a >= p.H[lvl] * p.W[lvl]) { a -= p.H[lvl] * p.W[lvl]; ++lvl; }
  const int H = p.H[lvl], W = p.W[lvl];
  const float stride = lvl == 0 ? 8.f : (lvl == 1 ? 16.f : 32.f);
  const float4* r = reinterpret_cast<const float4*>(p.raw[lvl] + ((size_t)b * H * W + a) * 144);
  float w16[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) w16[i] = p.dfl_w[i];
  float d[4];
#pragma unroll
  for (int side = 0; side < 4; ++side) {
    float v[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) { const float4 t = r[side * 4 + q]; v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w; }
    float mx = v[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) mx = fmaxf(mx, v[i]);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) { v[i] = expf(v[i] - mx); s += v[i]; }
    float e = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) e = __fadd_rn(e, __fmul_rn(__fdiv_rn(v[i], s), w16[i]));
    d[side] = e;
  }
  const float ax = (float)(a % W) + 0.5f, ay = (float)(a / W) + 0.5f;
  const float lx = ax - d[0], ly = ay - d[1], rx = ax + d[2], ry = ay + d[3];
  const float cx = ((lx + rx) / 2.f) * stride, cy = ((ly + ry) / 2.f) * stride;
  const float bw = (rx - lx) * stride, bh = (ry - ly) * stride;
  float best = -1.f; int bi = 0;
#pragma unroll 4
  for (int q = 0; q < 20; ++q) {
    const float4 t = r[16 + q];
    const float l[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float sg = 1.0f / (1.0f + expf(-l[e]));
      if (sg > best) { best = sg; bi = q * 4 + e; }
    }
  }
  float* o = p.det + idx * 6;
  o[0] = cx - bw / 2.f; o[1] = cy - bh / 2.f; o[2] = cx + bw / 2.f; o[3] = cy + bh / 2.f;
  o[4] = best >= p.conf ? best : 0.f;
  o[5] = (float)bi;
}

void launch_decode(const DecodeP& p, hipStream_t stream) {
  const size_t total = (size_t)p.B * p.A;
  hipLaunchKernelGGL(decode_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, p);
  CC_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------
// One 1024-thread workgroup per image.  key = score_bits<<32 | ~anchor  (unique; descending key order
// == descending score, ties by ascending anchor index == a stable descending sort).
constexpr int kMaxDet = 300;
constexpr int kSortN = 512;

__device__ __forceinline__ unsigned long long det_key(const float* det, int a) {
  return ((unsigned long long)__float_as_uint(det[(size_t)a * 6 + 4]) << 32) | (unsigned)(~(unsigned)a);
}

__global__ __launch_bounds__(1024) void topk_nms_kernel(const NmsP p) {
  __shared__ unsigned hist[256];
  __shared__ unsigned long long s_prefix;
  __shared__ int s_k;
  __shared__ unsigned s_cnt;
  __shared__ unsigned long long keys[kSortN];
  __shared__ float bx[kMaxDet][6];
  const int tid = threadIdx.x, b = blockIdx.x;
  const float* det = p.det + (size_t)b * p.A * 6;
  const int K = p.A < kMaxDet ? p.A : kMaxDet;

  if (tid == 0) { s_prefix = 0ull; s_k = K; s_cnt = 0; }
  // ---- radix select (MSB first, 8 bits per pass) of the K-th largest key
  for (int byte = 7; byte >= 0; --byte) {
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    const unsigned long long prefix = s_prefix;
    const int sh = 8 * (byte + 1);
    for (int a = tid; a < p.A; a += 1024) {
      const unsigned long long key = det_key(det, a);
      const bool match = byte == 7 ? true : ((key >> sh) == (prefix >> sh));
      if (match) atomicAdd(&hist[(unsigned)(key >> (8 * byte)) & 255u], 1u);
    }
    __syncthreads();
    if (tid < 64) {   // wave 0: suffix sums over buckets, 4 buckets per lane
      const unsigned h0 = hist[4 * tid], h1 = hist[4 * tid + 1], h2 = hist[4 * tid + 2], h3 = hist[4 * tid + 3];
      const unsigned s = h0 + h1 + h2 + h3;
      unsigned suf = s;   // inclusive suffix sum over lanes >= tid
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) { const unsigned t = __shfl_down(suf, off, 64); if (tid + off < 64) suf += t; }
      const unsigned above = suf - s;
      const unsigned kk = (unsigned)s_k;
      if (above < kk && kk <= suf) {
        unsigned cum = above; int v = 4 * tid + 3;
        const unsigned hh[4] = {h0, h1, h2, h3};
        for (int q = 3; q >= 0; --q) { if (cum + hh[q] >= kk) { v = 4 * tid + q; break; } cum += hh[q]; }
        s_prefix = prefix | ((unsigned long long)v << (8 * byte));
        s_k = (int)(kk - cum);
      }
    }
    __syncthreads();
  }
  // ---- collect the K keys >= threshold, pad, bitonic sort descending
  const unsigned long long thr = s_prefix;
  if (tid < kSortN) keys[tid] = 0ull;
  __syncthreads();
  if (K > 0)
    for (int a = tid; a < p.A; a += 1024) {
      const unsigned long long key = det_key(det, a);
      if (key >= thr) { const unsigned pos = atomicAdd(&s_cnt, 1u); if (pos < kSortN) keys[pos] = key; }
    }
  __syncthreads();
  for (int k = 2; k <= kSortN; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      if (tid < kSortN) {
        const int ixj = tid ^ j;
        if (ixj > tid) {
          const unsigned long long x = keys[tid], y = keys[ixj];
          const bool desc = (tid & k) == 0;
          if (desc ? (x < y) : (x > y)) { keys[tid] = y; keys[ixj] = x; }
        }
      }
      __syncthreads();
    }
  // ---- gather boxes
  if (tid < kMaxDet) {
    if (tid < K) {
      const unsigned a = ~(unsigned)(keys[tid] & 0xffffffffull);
      for (int c = 0; c < 6; ++c) bx[tid][c] = det[(size_t)a * 6 + c];
    } else
      for (int c = 0; c < 6; ++c) bx[tid][c] = 0.f;
  }
  __syncthreads();
  // ---- mask NMS: row j is suppressed by ANY earlier row i (suppressed or not) of the same class
  if (tid < kMaxDet) {
    const float x1 = bx[tid][0], y1 = bx[tid][1], x2 = bx[tid][2], y2 = bx[tid][3], cl = bx[tid][5];
    const float area = (x2 - x1) * (y2 - y1);
    bool sup = false;
    if (tid < K)
      for (int i = 0; i < tid; ++i) {
        const float ix1 = fmaxf(bx[i][0], x1), iy1 = fmaxf(bx[i][1], y1);
        const float ix2 = fminf(bx[i][2], x2), iy2 = fminf(bx[i][3], y2);
        const float inter = fmaxf(0.f, ix2 - ix1) * fmaxf(0.f, iy2 - iy1);
        const float ai = (bx[i][2] - bx[i][0]) * (bx[i][3] - bx[i][1]);
        const float iou = inter / (ai + area - inter);
        if (iou > p.iou_thr && bx[i][5] == cl) sup = true;
      }
    const float keep = (sup || tid >= K) ? 0.f : 1.f;
    float* o = p.out + ((size_t)b * kMaxDet + tid) * 6;
    // scale_boxes + clip_boxes (applied to zeroed rows too)
    o[0] = fminf(fmaxf((x1 * keep - p.pad_x) / p.gain, 0.f), p.src_w);
    o[1] = fminf(fmaxf((y1 * keep - p.pad_y) / p.gain, 0.f), p.src_h);
    o[2] = fminf(fmaxf((x2 * keep - p.pad_x) / p.gain, 0.f), p.src_w);
    o[3] = fminf(fmaxf((y2 * keep - p.pad_y) / p.gain, 0.f), p.src_h);
    o[4] = bx[tid][4] * keep;
    o[5] = cl * keep;
  }
}

void launch_topk_nms(const NmsP& p, hipStream_t stream) {
  hipLaunchKernelGGL(topk_nms_kernel, dim3(p.B), dim3(1024), 0, stream, p);
  CC_HIP(hipGetLastError());
}

}  // namespace cc

// Generic direct convolution (one thread per output element) and pooling kernels.
//
// conv_direct is the correctness fallback for shapes the MFMA kernel rejects (channel counts that are
// not multiples of 16 bytes: YOLOv9-m's 60/90/184-wide layers).  Same ConvP contract as conv_mfma.
// Pools: avg_pool2d(k2,s1,p0) and max_pool2d(k3 s2 p1 / k5 s1 p2, -inf padding) of ADown/AConv/SPPELAN
// (detection/yolov9.py:45-63,127-149), reading/writing channel slices of NHWC buffers.
#include "kernels.h"

namespace cc {

template <class T>
__global__ __launch_bounds__(256) void conv_direct_kernel(const ConvP p) {
  const size_t total = (size_t)p.B * p.Ho * p.Wo * p.Cout;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int n = (int)(idx % p.Cout);
  const size_t m = idx / p.Cout;
  const int hw = p.Ho * p.Wo;
  const int b = (int)(m / hw), rem = (int)(m - (size_t)b * hw), ho = rem / p.Wo, wo = rem - ho * p.Wo;
  const T* w = reinterpret_cast<const T*>(p.w) + (size_t)n * p.Kw;
  float acc = 0.f;
  for (int r = 0; r < p.ks; ++r) {
    const int ih = ho * p.stride - p.pad + r;
    if ((unsigned)ih >= (unsigned)p.Hin) continue;
    for (int s = 0; s < p.ks; ++s) {
      const int iw = wo * p.stride - p.pad + s;
      if ((unsigned)iw >= (unsigned)p.Win) continue;
      for (int h = 0; h <= p.split; ++h) {               // split weights: the tap's channels again against the low plane
        const T* wk = w + ((r * p.ks + s) * (1 + p.split) + h) * p.Cin;
        {
          const Src& S = p.s0;
          const T* x = reinterpret_cast<const T*>(S.ptr) + ((size_t)(b * S.H + (ih >> S.shift)) * S.W + (iw >> S.shift)) * S.cstride + S.coff;
          for (int c = 0; c < S.C; ++c) acc = fmaf(to_f32<T>(x[c]), to_f32<T>(wk[c]), acc);
        }
        if (p.s1.C > 0) {
          const Src& S = p.s1;
          const T* x = reinterpret_cast<const T*>(S.ptr) + ((size_t)(b * S.H + (ih >> S.shift)) * S.W + (iw >> S.shift)) * S.cstride + S.coff;
          const T* wk1 = wk + p.s0.C;
          for (int c = 0; c < S.C; ++c) acc = fmaf(to_f32<T>(x[c]), to_f32<T>(wk1[c]), acc);
        }
      }
    }
  }
  float t = fmaf(acc, p.split ? p.oscale : 1.0f, p.bias ? p.bias[n] : 0.f);
  if (p.act == 1) t = t / (1.0f + expf(-t));
  else if (p.act == 2) t = 0.5f * t * (1.0f + tanhf(0.7978845608028654f * (t + 0.044715f * t * t * t)));
  else if (p.act == 3) t = t > 0.f ? t : p.slope[n] * t;
  if (p.res) {
    const size_t ri = m * p.res_cstride + p.res_coff + n;
    t = (p.res_f32 ? reinterpret_cast<const float*>(p.res)[ri] : to_f32<T>(reinterpret_cast<const T*>(p.res)[ri])) + t;
  }
  if (p.act == 4) t = fmaxf(t, 0.f);
  const size_t oi = m * p.out_cstride + p.out_coff + n;
  if (p.out_f32) reinterpret_cast<float*>(p.out)[oi] = t;
  else reinterpret_cast<T*>(p.out)[oi] = from_f32<T>(t);
}

void launch_conv_direct(int dt, const ConvP& p, hipStream_t stream) {
  const size_t total = (size_t)p.B * p.Ho * p.Wo * p.Cout;
  const dim3 grid((unsigned)((total + 255) / 256)), block(256);
  if (dt == F32) hipLaunchKernelGGL(conv_direct_kernel<float>, grid, block, 0, stream, p);
  else if (dt == F16) hipLaunchKernelGGL(conv_direct_kernel<f16_t>, grid, block, 0, stream, p);
  else hipLaunchKernelGGL(conv_direct_kernel<bf16_t>, grid, block, 0, stream, p);
  CC_HIP(hipGetLastError());
}

// ---- pooling ---------------------------------------------------------------------------------------
// Vector path: one thread per (pixel, 16-byte channel chunk): 8 halfs / 4 floats per load, coalesced
// along NHWC channels (HBM-bound op: read k*k-overlapping windows through L1/L2, write once).
// Grid = (chunks of a row, output row, image).  The window is a compile-time size and every load is unconditional
// (coordinates clamped into the image, the value replaced afterwards when the tap is padding): with a branch per tap
// the compiler emitted load / s_waitcnt vmcnt(0) pairs, i.e. k*k dependent round trips per thread (2 TB/s); now all
// taps of a thread are in flight together.
template <class T> __device__ inline void unpack_chunk(const uint4& u, float (&v)[16 / sizeof(T)]) {
  const T* t = reinterpret_cast<const T*>(&u);
#pragma unroll
  for (int e = 0; e < 16 / (int)sizeof(T); ++e) v[e] = to_f32<T>(t[e]);
}

template <class T, int K, int MODE>
__global__ __launch_bounds__(256) void pool_vec_kernel(const PoolP p) {
  constexpr int E = 16 / (int)sizeof(T);
  const int CV = p.C / E;
  const int x = blockIdx.x * 256 + threadIdx.x;
  if (x >= p.Wo * CV) return;
  const int wo = x / CV, cv = x - wo * CV, ho = blockIdx.y, b = blockIdx.z;
  const T* in = reinterpret_cast<const T*>(p.in) + p.in_coff + cv * E;
  float acc[E];
#pragma unroll
  for (int e = 0; e < E; ++e) acc[e] = MODE ? -INFINITY : 0.f;
#pragma unroll
  for (int r = 0; r < K; ++r) {
    const int ih = ho * p.stride - p.pad + r, ihc = min(max(ih, 0), p.H - 1);
    uint4 u[K];
#pragma unroll
    for (int s = 0; s < K; ++s) {
      const int iwc = min(max(wo * p.stride - p.pad + s, 0), p.W - 1);
      u[s] = *reinterpret_cast<const uint4*>(in + ((size_t)(b * p.H + ihc) * p.W + iwc) * p.in_cstride);
    }
#pragma unroll
    for (int s = 0; s < K; ++s) {
      const int iw = wo * p.stride - p.pad + s;
      const bool ok = ih == ihc && (unsigned)iw < (unsigned)p.W;
      float v[E];
      unpack_chunk<T>(u[s], v);
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const float t = ok ? v[e] : (MODE ? -INFINITY : 0.f);         // padding: -inf for max, 0 (counted) for avg
        acc[e] = MODE ? fmaxf(acc[e], t) : acc[e] + t;
      }
    }
  }
  uint4 o;
  T* t = reinterpret_cast<T*>(&o);
  const float inv = 1.0f / (float)(K * K);
#pragma unroll
  for (int e = 0; e < E; ++e) t[e] = from_f32<T>(MODE ? acc[e] : acc[e] * inv);
  const size_t m = ((size_t)b * p.Ho + ho) * p.Wo + wo;
  *reinterpret_cast<uint4*>(reinterpret_cast<T*>(p.out) + m * p.out_cstride + p.out_coff + cv * E) = o;
}

// ADown's avg_pool2d(2, stride 1, pad 0) (detection/yolov9.py:45) with R output rows per thread (round 5): the R + 1 input rows of a
// thread's two columns are loaded once - 2 (R + 1) loads for R outputs instead of 4 R, all in flight together - and every output sums its
// four taps in the order of pool_vec_kernel<T, 2, 0> ((r0,s0) + (r0,s1) + (r1,s0) + (r1,s1), then x 0.25): the same bits.
template <class T, int R>
__global__ __launch_bounds__(256) void avg2_rows_kernel(const PoolP p) {
  constexpr int E = 16 / (int)sizeof(T);
  const int CV = p.C / E;
  const int x = blockIdx.x * 256 + threadIdx.x;
  if (x >= p.Wo * CV) return;
  const int wo = x / CV, cv = x - wo * CV, ho0 = blockIdx.y * R, b = blockIdx.z;
  const T* in = reinterpret_cast<const T*>(p.in) + p.in_coff + cv * E;
  uint4 u[R + 1][2];
#pragma unroll
  for (int r = 0; r <= R; ++r) {
    const int ih = min(ho0 + r, p.H - 1);                  // rows past the map are loaded (clamped) and never used
#pragma unroll
    for (int c = 0; c < 2; ++c) u[r][c] = *reinterpret_cast<const uint4*>(in + ((size_t)(b * p.H + ih) * p.W + (wo + c)) * p.in_cstride);
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int ho = ho0 + r;
    if (ho >= p.Ho) break;
    float a0[E], a1[E], b0[E], b1[E];
    unpack_chunk<T>(u[r][0], a0); unpack_chunk<T>(u[r][1], a1); unpack_chunk<T>(u[r + 1][0], b0); unpack_chunk<T>(u[r + 1][1], b1);
    uint4 o;
    T* t = reinterpret_cast<T*>(&o);
#pragma unroll
    for (int e = 0; e < E; ++e) t[e] = from_f32<T>((((0.f + a0[e]) + a1[e]) + b0[e] + b1[e]) * 0.25f);
    const size_t m = ((size_t)b * p.Ho + ho) * p.Wo + wo;
    *reinterpret_cast<uint4*>(reinterpret_cast<T*>(p.out) + m * p.out_cstride + p.out_coff + cv * E) = o;
  }
}

// ADown's second branch (detection/yolov9.py:47-51): avg_pool2d(2, stride 1) followed by max_pool2d(3, stride 2, pad 1)
// on the same channels, in one pass: out(ho,wo) = max over the 3x3 window of pooled positions (2ho-1+r, 2wo-1+s) inside
// [0,H-2]x[0,W-2] of the 2x2 average there.  The full-resolution averaged tensor is never written or read back.
// Each average is summed in the order of the unfused kernel and rounding to T is monotone, so max-then-round gives
// exactly what round-then-max gave.  The 4x4 input window is 16 unconditional loads (clamped coordinates; a pooled
// position outside the averaged map is skipped by index, so what a clamped tap holds never matters).
template <class T>
__global__ __launch_bounds__(256) void avgmax_pool_kernel(const PoolP p) {
  constexpr int E = 16 / (int)sizeof(T);
  const int CV = p.C / E;
  const int x = blockIdx.x * 256 + threadIdx.x;
  if (x >= p.Wo * CV) return;
  const int wo = x / CV, cv = x - wo * CV, ho = blockIdx.y, b = blockIdx.z;
  const T* in = reinterpret_cast<const T*>(p.in) + p.in_coff + cv * E;
  const int i0 = 2 * ho - 1, j0 = 2 * wo - 1;               // top-left input pixel of the 4x4 window
  uint4 u[4][4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int ic = min(max(i0 + r, 0), p.H - 1);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int jc = min(max(j0 + c, 0), p.W - 1);
      u[r][c] = *reinterpret_cast<const uint4*>(in + ((size_t)(b * p.H + ic) * p.W + jc) * p.in_cstride);
    }
  }
  float acc[E];
#pragma unroll
  for (int e = 0; e < E; ++e) acc[e] = -INFINITY;
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const int ph = i0 + r;                                   // pooled row: averages input rows ph, ph+1
    float top[4][E], bot[4][E];
#pragma unroll
    for (int c = 0; c < 4; ++c) { unpack_chunk<T>(u[r][c], top[c]); unpack_chunk<T>(u[r + 1][c], bot[c]); }
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const int pw = j0 + s;
      const bool ok = ph >= 0 && ph <= p.H - 2 && pw >= 0 && pw <= p.W - 2;
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const float a = (((top[s][e] + top[s + 1][e]) + bot[s][e]) + bot[s + 1][e]) * 0.25f;
        acc[e] = fmaxf(acc[e], ok ? a : -INFINITY);
      }
    }
  }
  uint4 o;
  T* t = reinterpret_cast<T*>(&o);
#pragma unroll
  for (int e = 0; e < E; ++e) t[e] = from_f32<T>(acc[e]);
  const size_t m = ((size_t)b * p.Ho + ho) * p.Wo + wo;
  *reinterpret_cast<uint4*>(reinterpret_cast<T*>(p.out) + m * p.out_cstride + p.out_coff + cv * E) = o;
}

// avgmax_pool_kernel with two output rows per thread (round 5): the 4x4 windows of vertically adjacent outputs share two input rows, so
// six rows x four columns are loaded once (24 loads for two outputs instead of 32).  Same window walk and summation order per output.
template <class T>
__global__ __launch_bounds__(256) void avgmax_rows2_kernel(const PoolP p) {
  constexpr int E = 16 / (int)sizeof(T);
  const int CV = p.C / E;
  const int x = blockIdx.x * 256 + threadIdx.x;
  if (x >= p.Wo * CV) return;
  const int wo = x / CV, cv = x - wo * CV, ho0 = blockIdx.y * 2, b = blockIdx.z;
  const T* in = reinterpret_cast<const T*>(p.in) + p.in_coff + cv * E;
  const int i0 = 2 * ho0 - 1, j0 = 2 * wo - 1;             // top-left input pixel of the first output's window
  uint4 u[6][4];
#pragma unroll
  for (int r = 0; r < 6; ++r) {
    const int ic = min(max(i0 + r, 0), p.H - 1);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int jc = min(max(j0 + c, 0), p.W - 1);
      u[r][c] = *reinterpret_cast<const uint4*>(in + ((size_t)(b * p.H + ic) * p.W + jc) * p.in_cstride);
    }
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int ho = ho0 + q;
    if (ho >= p.Ho) break;
    float acc[E];
#pragma unroll
    for (int e = 0; e < E; ++e) acc[e] = -INFINITY;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int ph = i0 + 2 * q + r;
      float top[4][E], bot[4][E];
#pragma unroll
      for (int c = 0; c < 4; ++c) { unpack_chunk<T>(u[2 * q + r][c], top[c]); unpack_chunk<T>(u[2 * q + r + 1][c], bot[c]); }
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const int pw = j0 + s;
        const bool ok = ph >= 0 && ph <= p.H - 2 && pw >= 0 && pw <= p.W - 2;
#pragma unroll
        for (int e = 0; e < E; ++e) {
          const float a = (((top[s][e] + top[s + 1][e]) + bot[s][e]) + bot[s + 1][e]) * 0.25f;
          acc[e] = fmaxf(acc[e], ok ? a : -INFINITY);
        }
      }
    }
    uint4 o;
    T* t = reinterpret_cast<T*>(&o);
#pragma unroll
    for (int e = 0; e < E; ++e) t[e] = from_f32<T>(acc[e]);
    const size_t m = ((size_t)b * p.Ho + ho) * p.Wo + wo;
    *reinterpret_cast<uint4*>(reinterpret_cast<T*>(p.out) + m * p.out_cstride + p.out_coff + cv * E) = o;
  }
}

// Scalar fallback: one thread per (pixel, channel), any channel count / alignment.
template <class T>
__global__ __launch_bounds__(256) void pool_kernel(const PoolP p) {
  const size_t total = (size_t)p.B * p.Ho * p.Wo * p.C;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int c = (int)(idx % p.C);
  const size_t m = idx / p.C;
  const int hw = p.Ho * p.Wo;
  const int b = (int)(m / hw), rem = (int)(m - (size_t)b * hw), ho = rem / p.Wo, wo = rem - ho * p.Wo;
  const T* in = reinterpret_cast<const T*>(p.in);
  float acc = p.mode ? -INFINITY : 0.f;
  for (int r = 0; r < p.k; ++r) {
    const int ih = ho * p.stride - p.pad + r;
    if ((unsigned)ih >= (unsigned)p.H) continue;
    for (int s = 0; s < p.k; ++s) {
      const int iw = wo * p.stride - p.pad + s;
      if ((unsigned)iw >= (unsigned)p.W) continue;
      const float v = to_f32<T>(in[((size_t)(b * p.H + ih) * p.W + iw) * p.in_cstride + p.in_coff + c]);
      acc = p.mode ? fmaxf(acc, v) : acc + v;
    }
  }
  if (!p.mode) acc = acc * (1.0f / (float)(p.k * p.k));
  reinterpret_cast<T*>(p.out)[m * p.out_cstride + p.out_coff + c] = from_f32<T>(acc);
}

template <class T> static void launch_pool_t(const PoolP& p, hipStream_t stream) {
  constexpr int E = 16 / (int)sizeof(T);
  const bool vec = p.C % E == 0 && p.in_coff % E == 0 && p.in_cstride % E == 0 && p.out_coff % E == 0 && p.out_cstride % E == 0 &&
                   ((uintptr_t)p.in & 15) == 0 && ((uintptr_t)p.out & 15) == 0;
  const dim3 vgrid((unsigned)((p.Wo * (p.C / E) + 255) / 256), (unsigned)p.Ho, (unsigned)p.B);
  if (p.mode == 2) {
    CC_CHECK(vec && p.k == 3 && p.stride == 2 && p.pad == 1, "avg-max pool: needs 16-byte channel chunks, k=3 s=2 p=1");
    static const int rows2 = [] { const char* e = getenv("CLEARCAM_POOL_ROWS"); return e ? atoi(e) : 4; }();
    if (rows2 != 1 && p.Ho >= 32) hipLaunchKernelGGL(avgmax_rows2_kernel<T>, dim3(vgrid.x, (unsigned)((p.Ho + 1) / 2), vgrid.z), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL(avgmax_pool_kernel<T>, vgrid, dim3(256), 0, stream, p);
  } else if (vec && p.mode == 0 && p.k == 2) {
    // stride 1, no padding, at least 64 rows: eight output rows per thread (CLEARCAM_POOL_ROWS=4 / 1: four / one).  Pools per 64-frame step:
    // 0.726 ms with one row, 0.660 with four (+ two rows in the avg-max kernel), 0.654 with eight - same bits (profiles/r05t_pool_rows.txt)
    static const int rows = [] { const char* e = getenv("CLEARCAM_POOL_ROWS"); return e ? atoi(e) : 8; }();
    const bool plain = p.stride == 1 && p.pad == 0 && p.Ho >= 64 && p.Wo == p.W - 1 && p.Ho == p.H - 1;
    if (rows == 8 && plain)
      hipLaunchKernelGGL((avg2_rows_kernel<T, 8>), dim3(vgrid.x, (unsigned)((p.Ho + 7) / 8), vgrid.z), dim3(256), 0, stream, p);
    else if (rows == 4 && plain)
      hipLaunchKernelGGL((avg2_rows_kernel<T, 4>), dim3(vgrid.x, (unsigned)((p.Ho + 3) / 4), vgrid.z), dim3(256), 0, stream, p);
    else
    hipLaunchKernelGGL((pool_vec_kernel<T, 2, 0>), vgrid, dim3(256), 0, stream, p);
  } else if (vec && p.mode == 1 && p.k == 1) {                       // AdaFace's MaxPool2d(1, stride) shortcut: a strided copy
    hipLaunchKernelGGL((pool_vec_kernel<T, 1, 1>), vgrid, dim3(256), 0, stream, p);
  } else if (vec && p.mode == 1 && p.k == 2) {                       // BlazeFace's max_pool2d(2, 2)
    hipLaunchKernelGGL((pool_vec_kernel<T, 2, 1>), vgrid, dim3(256), 0, stream, p);
  } else if (vec && p.mode == 1 && p.k == 3) {
    hipLaunchKernelGGL((pool_vec_kernel<T, 3, 1>), vgrid, dim3(256), 0, stream, p);
  } else if (vec && p.mode == 1 && p.k == 5) {
    hipLaunchKernelGGL((pool_vec_kernel<T, 5, 1>), vgrid, dim3(256), 0, stream, p);
  } else {
    const size_t total = (size_t)p.B * p.Ho * p.Wo * p.C;
    hipLaunchKernelGGL(pool_kernel<T>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, p);
  }
  CC_HIP(hipGetLastError());
}

// ---- CBFuse: sum of nearest-upsampled channel slices, 16 bytes of channels per thread ----------------
template <class T>
__global__ __launch_bounds__(256) void fuse_kernel(const FuseP p) {
  constexpr int E = 16 / (int)sizeof(T);
  const int CV = p.C / E;
  const size_t total = (size_t)p.B * p.Ho * p.Wo * CV;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int cv = (int)(idx % CV);
  const size_t m = idx / CV;
  const int hw = p.Ho * p.Wo;
  const int b = (int)(m / hw), rem = (int)(m - (size_t)b * hw), ho = rem / p.Wo, wo = rem - ho * p.Wo;
  float acc[E];
#pragma unroll
  for (int e = 0; e < E; ++e) acc[e] = 0.f;
  for (int k = 0; k < p.n; ++k) {
    const T* in = reinterpret_cast<const T*>(p.in[k]) +
                  ((size_t)(b * p.H[k] + (ho >> p.shift[k])) * p.W[k] + (wo >> p.shift[k])) * p.cstride[k] + p.coff[k] + cv * E;
    const uint4 u = *reinterpret_cast<const uint4*>(in);
    const T* t = reinterpret_cast<const T*>(&u);
#pragma unroll
    for (int e = 0; e < E; ++e) acc[e] += to_f32<T>(t[e]);
  }
  uint4 o;
  T* t = reinterpret_cast<T*>(&o);
#pragma unroll
  for (int e = 0; e < E; ++e) t[e] = from_f32<T>(acc[e]);
  *reinterpret_cast<uint4*>(reinterpret_cast<T*>(p.out) + m * p.out_cstride + p.out_coff + cv * E) = o;
}

void launch_fuse(int dt, const FuseP& p, hipStream_t stream) {
  const int E = dt == F32 ? 4 : 8;
  CC_CHECK(p.C % E == 0 && p.out_coff % E == 0 && p.out_cstride % E == 0, "fuse: channel alignment");
  for (int k = 0; k < p.n; ++k) CC_CHECK(p.coff[k] % E == 0 && p.cstride[k] % E == 0, "fuse: input channel alignment");
  const size_t total = (size_t)p.B * p.Ho * p.Wo * (p.C / E);
  const dim3 grid((unsigned)((total + 255) / 256)), block(256);
  if (dt == F32) hipLaunchKernelGGL(fuse_kernel<float>, grid, block, 0, stream, p);
  else if (dt == F16) hipLaunchKernelGGL(fuse_kernel<f16_t>, grid, block, 0, stream, p);
  else hipLaunchKernelGGL(fuse_kernel<bf16_t>, grid, block, 0, stream, p);
  CC_HIP(hipGetLastError());
}

void launch_pool(int dt, const PoolP& p, hipStream_t stream) {
  if (dt == F32) launch_pool_t<float>(p, stream);
  else if (dt == F16) launch_pool_t<f16_t>(p, stream);
  else launch_pool_t<bf16_t>(p, stream);
}

}  // namespace cc

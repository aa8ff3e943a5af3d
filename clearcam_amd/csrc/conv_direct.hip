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
      const T* wk = w + (r * p.ks + s) * p.Cin;
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
  float t = acc + (p.bias ? p.bias[n] : 0.f);
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
template <class T>
__global__ __launch_bounds__(256) void pool_vec_kernel(const PoolP p) {
  constexpr int E = 16 / (int)sizeof(T);
  const int CV = p.C / E;
  const size_t total = (size_t)p.B * p.Ho * p.Wo * CV;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int cv = (int)(idx % CV);
  const size_t m = idx / CV;
  const int hw = p.Ho * p.Wo;
  const int b = (int)(m / hw), rem = (int)(m - (size_t)b * hw), ho = rem / p.Wo, wo = rem - ho * p.Wo;
  const T* in = reinterpret_cast<const T*>(p.in) + p.in_coff + cv * E;
  float acc[E];
#pragma unroll
  for (int e = 0; e < E; ++e) acc[e] = p.mode ? -INFINITY : 0.f;
  for (int r = 0; r < p.k; ++r) {
    const int ih = ho * p.stride - p.pad + r;
    if ((unsigned)ih >= (unsigned)p.H) continue;
    for (int s = 0; s < p.k; ++s) {
      const int iw = wo * p.stride - p.pad + s;
      if ((unsigned)iw >= (unsigned)p.W) continue;
      const uint4 u = *reinterpret_cast<const uint4*>(in + ((size_t)(b * p.H + ih) * p.W + iw) * p.in_cstride);
      const T* t = reinterpret_cast<const T*>(&u);
#pragma unroll
      for (int e = 0; e < E; ++e) { const float v = to_f32<T>(t[e]); acc[e] = p.mode ? fmaxf(acc[e], v) : acc[e] + v; }
    }
  }
  uint4 o;
  T* t = reinterpret_cast<T*>(&o);
  const float inv = 1.0f / (float)(p.k * p.k);
#pragma unroll
  for (int e = 0; e < E; ++e) t[e] = from_f32<T>(p.mode ? acc[e] : acc[e] * inv);
  *reinterpret_cast<uint4*>(reinterpret_cast<T*>(p.out) + m * p.out_cstride + p.out_coff + cv * E) = o;
}

// ADown's second branch (detection/yolov9.py:47-51): avg_pool2d(2, stride 1) followed by max_pool2d(3, stride 2, pad 1)
// on the same channels, in one pass: out(ho,wo) = max over the 3x3 window of pooled positions (2ho-1+r, 2wo-1+s) inside
// [0,H-2]x[0,W-2] of the 2x2 average there.  The full-resolution averaged tensor is never written or read back.
// Each average is summed in the order of the unfused kernel and rounding to T is monotone, so max-then-round gives
// exactly what round-then-max gave.  The 4x4 input window lives in registers (two rows at a time).
template <class T>
__global__ __launch_bounds__(256) void avgmax_pool_kernel(const PoolP p) {
  constexpr int E = 16 / (int)sizeof(T);
  const int CV = p.C / E;
  const size_t total = (size_t)p.B * p.Ho * p.Wo * CV;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int cv = (int)(idx % CV);
  const size_t m = idx / CV;
  const int hw = p.Ho * p.Wo;
  const int b = (int)(m / hw), rem = (int)(m - (size_t)b * hw), ho = rem / p.Wo, wo = rem - ho * p.Wo;
  const T* in = reinterpret_cast<const T*>(p.in) + p.in_coff + cv * E;
  const int i0 = 2 * ho - 1, j0 = 2 * wo - 1;               // top-left input pixel of the 4x4 window
  float prev[4][E], cur[4][E], acc[E];
#pragma unroll
  for (int e = 0; e < E; ++e) acc[e] = -INFINITY;
  auto load_row = [&](int i, float (&row)[4][E]) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int j = j0 + c;
      uint4 u = make_uint4(0, 0, 0, 0);
      if ((unsigned)i < (unsigned)p.H && (unsigned)j < (unsigned)p.W)
        u = *reinterpret_cast<const uint4*>(in + ((size_t)(b * p.H + i) * p.W + j) * p.in_cstride);
      const T* t = reinterpret_cast<const T*>(&u);
#pragma unroll
      for (int e = 0; e < E; ++e) row[c][e] = to_f32<T>(t[e]);
    }
  };
  load_row(i0, prev);
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    load_row(i0 + r + 1, cur);
    const int ph = i0 + r;                                   // pooled row: averages input rows ph, ph+1
    if (ph >= 0 && ph <= p.H - 2) {
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const int pw = j0 + s;
        if (pw < 0 || pw > p.W - 2) continue;
#pragma unroll
        for (int e = 0; e < E; ++e) {
          const float a = (((prev[s][e] + prev[s + 1][e]) + cur[s][e]) + cur[s + 1][e]) * 0.25f;
          acc[e] = fmaxf(acc[e], a);
        }
      }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int e = 0; e < E; ++e) prev[c][e] = cur[c][e];
  }
  uint4 o;
  T* t = reinterpret_cast<T*>(&o);
#pragma unroll
  for (int e = 0; e < E; ++e) t[e] = from_f32<T>(acc[e]);
  *reinterpret_cast<uint4*>(reinterpret_cast<T*>(p.out) + m * p.out_cstride + p.out_coff + cv * E) = o;
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
  if (p.mode == 2) {
    CC_CHECK(vec && p.k == 3 && p.stride == 2 && p.pad == 1, "avg-max pool: needs 16-byte channel chunks, k=3 s=2 p=1");
    const size_t total = (size_t)p.B * p.Ho * p.Wo * (p.C / E);
    hipLaunchKernelGGL(avgmax_pool_kernel<T>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, p);
  } else if (vec) {
    const size_t total = (size_t)p.B * p.Ho * p.Wo * (p.C / E);
    hipLaunchKernelGGL(pool_vec_kernel<T>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, p);
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

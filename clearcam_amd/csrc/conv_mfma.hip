// NHWC implicit-GEMM convolution / linear layer on MFMA (gfx950), fused bias + SiLU/GELU + residual.
//
// Replaces what tinygrad codegen emits for `nn.Conv2d(...)(x).silu()` (detection/yolov9.py:33-38) and
// `x @ W.T + b` (models/objects.py:109,120,124-126).  One kernel covers 1x1 (a plain GEMM), 3x3 s1/s2,
// two-source channel concat and nearest-x2 upsample-on-load; im2col is never materialised.
//
// GEMM view: D[n][m] = sum_k W[n][k] * X[m][k];  m = output pixel, n = output channel, k = (tap, channel).
// Weights are the MFMA A operand (rows n), activations the B operand (cols m), so each lane ends up
// with 4 consecutive output channels of one pixel -> 8/16-byte NHWC stores.
//
// Tile: 128 pixels x BN channels x 128 bytes of K (64 halfs / 32 floats) per step, 256 threads = 4 waves.
// Staging: global -> registers (next tile, issued before the MFMAs of the current one) -> LDS rows of
// 128 B whose eight 16-B chunks are XOR-swizzled by (row>>1)&7, which makes both the ds_write_b128
// (8 lanes = one row) and the MFMA-fragment ds_read_b128 (16 rows x one chunk column) conflict-free.
// f32 mode uses v_mfma_f32_16x16x4_f32 (exact f32) with the k-slots of a 16-B chunk spread over 4 MFMAs.
#include "kernels.h"
#include "mfma.h"

namespace cc {

template <class T> __device__ __forceinline__ float act_silu(float x) {
  if constexpr (sizeof(T) == 4) return x / (1.0f + expf(-x));
  else return x * __frcp_rn(1.0f + __expf(-x));
}
template <class T> __device__ __forceinline__ float act_gelu_tanh(float x) {
  // tinygrad Tensor.gelu(): 0.5*x*(1+tanh(sqrt(2/pi)*(x+0.044715*x^3)))  
  const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
  if constexpr (sizeof(T) == 4) return 0.5f * x * (1.0f + tanhf(u));
  else { const float e = __expf(2.0f * u); return 0.5f * x * (1.0f + (1.0f - 2.0f * __frcp_rn(e + 1.0f))); }
}

template <class T> __device__ __forceinline__ void store4(void* base, size_t idx, const float (&v)[4]) {
  if constexpr (sizeof(T) == 4) {
    *reinterpret_cast<float4*>(reinterpret_cast<float*>(base) + idx) = make_float4(v[0], v[1], v[2], v[3]);
  } else {
    alignas(8) T t[4] = {from_f32<T>(v[0]), from_f32<T>(v[1]), from_f32<T>(v[2]), from_f32<T>(v[3])};
    *reinterpret_cast<uint2*>(reinterpret_cast<T*>(base) + idx) = *reinterpret_cast<uint2*>(t);
  }
}
template <class T> __device__ __forceinline__ void load4(const void* base, size_t idx, float (&v)[4]) {
  if constexpr (sizeof(T) == 4) {
    const float4 f = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + idx);
    v[0] = f.x; v[1] = f.y; v[2] = f.z; v[3] = f.w;
  } else {
    uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const T*>(base) + idx);
    const T* t = reinterpret_cast<const T*>(&u);
    for (int i = 0; i < 4; ++i) v[i] = to_f32<T>(t[i]);
  }
}

template <class T, int BN, int WM>
__global__ __launch_bounds__(256) void conv_mfma_kernel(const ConvP p) {
  constexpr int BM = 128, WN = 4 / WM;
  constexpr int MI = BM / WM / 16, NJ = BN / WN / 16;
  constexpr int E = 16 / (int)sizeof(T);   // elements per 16-byte chunk
  constexpr int BK = 8 * E;                // elements per K step (128 bytes)
  constexpr int WR = BN / 32;              // weight rows staged per thread
  __shared__ uint4 lds[(BM + BN) * 8];
  uint4* ldsX = lds;
  uint4* ldsW = lds + BM * 8;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int M = p.B * p.Ho * p.Wo;
  const int nt = (p.Cout + BN - 1) / BN;

  // XCD-aware tile order: consecutive tiles (same pixel rows, all channel tiles) share one XCD's L2.
  int wg;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int m0 = (wg / nt) * BM, n0 = (wg % nt) * BN;

  // ---- per-thread staging assignment: 16-B chunk column `chunk`, rows rowb + 32*i
  const int chunk = tid & 7, rowb = tid >> 3;
  int pb[4], ph0[4], pw0[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + rowb + 32 * i;
    if (m < M) {
      const int hw = p.Ho * p.Wo;
      const int b = m / hw, rem = m - b * hw, ho = rem / p.Wo, wo = rem - ho * p.Wo;
      pb[i] = b; ph0[i] = ho * p.stride - p.pad; pw0[i] = wo * p.stride - p.pad;
    } else { pb[i] = 0; ph0[i] = -(1 << 28); pw0[i] = 0; }
  }
  int k0 = chunk * E;                      // this thread's k index inside the current K step
  int kc, kr, ks_;                         // channel, tap row, tap col of k0
  { const int tap = k0 / p.Cin; kc = k0 - tap * p.Cin; kr = tap / p.ks; ks_ = tap - kr * p.ks; }

  uint4 xr[4], wr[WR];
  auto issue_loads = [&]() {
    const bool kok = k0 < p.Ktot;
    const bool first = kc < p.s0.C;
    const T* sp = reinterpret_cast<const T*>(first ? p.s0.ptr : p.s1.ptr);
    const int sH = first ? p.s0.H : p.s1.H, sW = first ? p.s0.W : p.s1.W;
    const int scs = first ? p.s0.cstride : p.s1.cstride, sco = first ? p.s0.coff : p.s1.coff;
    const int ssh = first ? p.s0.shift : p.s1.shift;
    const int cc = first ? kc : kc - p.s0.C;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int ih = ph0[i] + kr, iw = pw0[i] + ks_;
      const bool ok = kok && (unsigned)ih < (unsigned)p.Hin && (unsigned)iw < (unsigned)p.Win;
      const size_t off = ((size_t)(pb[i] * sH + (ih >> ssh)) * sW + (iw >> ssh)) * scs + sco + cc;
      xr[i] = ok ? *reinterpret_cast<const uint4*>(sp + off) : make_uint4(0, 0, 0, 0);
    }
    const T* wp = reinterpret_cast<const T*>(p.w);
#pragma unroll
    for (int i = 0; i < WR; ++i) {
      const int n = n0 + rowb + 32 * i;
      wr[i] = (kok && n < p.Cout) ? *reinterpret_cast<const uint4*>(wp + (size_t)n * p.Ktot + k0) : make_uint4(0, 0, 0, 0);
    }
  };
  auto advance_k = [&]() {
    k0 += BK; kc += BK;
    while (kc >= p.Cin) { kc -= p.Cin; if (++ks_ == p.ks) { ks_ = 0; ++kr; } }
  };

  f32x4 acc[NJ][MI];
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int i = 0; i < MI; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int wm0 = (wave % WM) * (BM / WM), wn0 = (wave / WM) * (BN / WN);
  const int fr = lane & 15, fg = lane >> 4;
  const int nkt = (p.Ktot + BK - 1) / BK;

  issue_loads();
  for (int kt = 0; kt < nkt; ++kt) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int row = rowb + 32 * i; ldsX[row * 8 + (chunk ^ ((row >> 1) & 7))] = xr[i]; }
#pragma unroll
    for (int i = 0; i < WR; ++i) { const int row = rowb + 32 * i; ldsW[row * 8 + (chunk ^ ((row >> 1) & 7))] = wr[i]; }
    __syncthreads();
    if (kt + 1 < nkt) { advance_k(); issue_loads(); }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      uint4 xf[MI], wf[NJ];
#pragma unroll
      for (int i = 0; i < MI; ++i) { const int row = wm0 + i * 16 + fr; xf[i] = ldsX[row * 8 + ((h * 4 + fg) ^ ((row >> 1) & 7))]; }
#pragma unroll
      for (int j = 0; j < NJ; ++j) { const int row = wn0 + j * 16 + fr; wf[j] = ldsW[row * 8 + ((h * 4 + fg) ^ ((row >> 1) & 7))]; }
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int i = 0; i < MI; ++i) Mma<T>::run(wf[j], xf[i], acc[j][i]);
    }
  }

  // ---- epilogue: bias -> activation -> (+residual) -> store 4 consecutive channels of one pixel
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int n = n0 + wn0 + j * 16 + fg * 4;
    if (n >= p.Cout) continue;
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) { const float4 b4 = *reinterpret_cast<const float4*>(p.bias + n); bv[0] = b4.x; bv[1] = b4.y; bv[2] = b4.z; bv[3] = b4.w; }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int m = m0 + wm0 + i * 16 + fr;
      if (m >= M) continue;
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float t = acc[j][i][e] + bv[e];
        if (p.act == 1) t = act_silu<T>(t); else if (p.act == 2) t = act_gelu_tanh<T>(t);
        v[e] = t;
      }
      if (p.res) {
        float rv[4];
        const size_t ri = (size_t)m * p.res_cstride + p.res_coff + n;
        if (p.res_f32) load4<float>(p.res, ri, rv); else load4<T>(p.res, ri, rv);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = rv[e] + v[e];
      }
      const size_t oi = (size_t)m * p.out_cstride + p.out_coff + n;
      if (p.out_f32) store4<float>(p.out, oi, v); else store4<T>(p.out, oi, v);
    }
  }
}

template <class T> static bool supported_t(const ConvP& p) {
  const int E = 16 / (int)sizeof(T);
  auto src_ok = [&](const Src& s) { return s.C == 0 || (s.C % E == 0 && s.coff % E == 0 && s.cstride % E == 0 && ((uintptr_t)s.ptr & 15) == 0); };
  if (!src_ok(p.s0) || !src_ok(p.s1)) return false;
  if (p.Cin % E || p.Cout % 4 || p.out_coff % 4 || p.out_cstride % 4) return false;
  if (p.res && (p.res_coff % 4 || p.res_cstride % 4)) return false;
  if (p.s0.C + p.s1.C != p.Cin || p.Ktot != p.ks * p.ks * p.Cin) return false;
  return true;
}

bool conv_mfma_supported(int dt, const ConvP& p) {
  return dt == F32 ? supported_t<float>(p) : supported_t<f16_t>(p);
}

template <class T> static void launch_t(const ConvP& p, hipStream_t stream) {
  const int M = p.B * p.Ho * p.Wo;
  const int mt = (M + 127) / 128;
  auto padded = [&](int bn) { return (p.Cout + bn - 1) / bn * bn; };
  int bn = 128;
  if (padded(64) < padded(bn)) bn = 64;
  if (padded(32) < padded(bn)) bn = 32;
  const int nt = (p.Cout + bn - 1) / bn;
  const dim3 grid(mt * nt), block(256);
  if (bn == 128) hipLaunchKernelGGL((conv_mfma_kernel<T, 128, 2>), grid, block, 0, stream, p);
  else if (bn == 64) hipLaunchKernelGGL((conv_mfma_kernel<T, 64, 2>), grid, block, 0, stream, p);
  else hipLaunchKernelGGL((conv_mfma_kernel<T, 32, 4>), grid, block, 0, stream, p);
  CC_HIP(hipGetLastError());
}

void launch_conv_mfma(int dt, const ConvP& p, hipStream_t stream) {
  CC_CHECK(conv_mfma_supported(dt, p), "conv_mfma: unsupported shape/alignment");
  if (dt == F32) launch_t<float>(p, stream);
  else if (dt == F16) launch_t<f16_t>(p, stream);
  else launch_t<bf16_t>(p, stream);
}

void launch_conv(int dt, const ConvP& p, hipStream_t stream) {
  if (conv_mfma_supported(dt, p)) launch_conv_mfma(dt, p, stream);
  else launch_conv_direct(dt, p, stream);
}

}  // namespace cc

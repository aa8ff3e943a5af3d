// Device-side building blocks shared by the conv / GEMM kernels (conv_mfma.hip, conv_phase.hip): activations, LDS-DMA issue,
// chunk swizzle, the fused epilogue (bias + activation + residual, staged through LDS for 16-byte stores).
#pragma once
#include <algorithm>
#include <type_traits>
#include "kernels.h"
#include "mfma.h"

namespace cc {

static __device__ const uint4 g_zero16 = {0u, 0u, 0u, 0u};   // 16-byte zero page: the source of halo / out-of-range DMA rows

struct ConvAux {
  float inv_hw, inv_wo;   // 1/(Ho*Wo), 1/Wo for divide-free pixel decomposition
  int nt;                 // channel tiles per pixel tile
  int is1x1;              // 1x1, stride 1, no pad, no upsample: input pixel index == output pixel index
  int flags;              // tuning switches of conv_phase_kernel (CLEARCAM_PHASE_FLAGS)
  unsigned x_bytes, w_bytes;   // conv_phase_kernel with buffer-descriptor DMA: sizes of the input tensor and of the weights
  int ntiles;             // conv_phase_persist_kernel: tiles in the layer
  int two;                // conv_phase_kernel: 1x1 conv over the channel concat of two sources (either may be read nearest-upsampled)
};

// accumulator -> pre-activation value: fma(acc, out_scale, bias).  1 unless the layer's weights are split (ConvP::split), where it is
// the exact power of two that undoes the weight scale; fma(acc, 1, b) IS acc + b, so the other modes keep their bits.
__device__ __forceinline__ float out_scale(const ConvP& p) { return p.split ? p.oscale : 1.0f; }
// K walk over "virtual taps" (ConvP::split): leaving virtual tap vt - 1 for vt moves to the next filter tap unless vt is the
// low-plane pass of the same tap.
__device__ __forceinline__ bool next_filter_tap(int vt, int split) { return !(vt & split); }

template <class T, int ACT> __device__ __forceinline__ float activate(float x) {
  if constexpr (ACT == 1) {            // SiLU
    if constexpr (sizeof(T) == 4) return x / (1.0f + expf(-x));
    else return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f));   // v_exp_f32 + v_rcp_f32 (1 ulp): plenty for a 16-bit result
  } else if constexpr (ACT == 2) {     // tinygrad Tensor.gelu(): 0.5*x*(1+tanh(sqrt(2/pi)*(x+0.044715*x^3)))  
    const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
    if constexpr (sizeof(T) == 4) return 0.5f * x * (1.0f + tanhf(u));
    else {
      // 0.5 x (1 + tanh u) = x * sigmoid(2u) = x / (1 + 2^(-2u log2 e)): one fma chain, one v_exp_f32, one v_rcp_f32.
      // The epilogue is not overlapped with MFMA work in the one-block-per-CU kernels, so its VALU count is paid in full.
      const float t = x * x;
      const float z = x * __builtin_fmaf(t, -0.044715f * 2.0f * 0.7978845608028654f * 1.4426950408889634f, -2.0f * 0.7978845608028654f * 1.4426950408889634f);
      return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z));
    }
  } else return x;
}
template <class T> __device__ __forceinline__ float activate_rt(float x, int act) {
  return act == 1 ? activate<T, 1>(x) : (act == 2 ? activate<T, 2>(x) : x);
}

template <class T> __device__ __forceinline__ void store4(void* base, size_t idx, const float (&v)[4]) {
  if constexpr (sizeof(T) == 4) {
    *reinterpret_cast<float4*>(reinterpret_cast<float*>(base) + idx) = make_float4(v[0], v[1], v[2], v[3]);
  } else {
    *reinterpret_cast<uint2*>(reinterpret_cast<T*>(base) + idx) = make_uint2(pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3]));
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

// LDS-DMA issue in inline asm: hipcc otherwise counts the DMA as an LDS write that may alias the next
// ds_read and drains it (s_waitcnt vmcnt(0)) before the MFMA phase, serialising load and compute.  Hidden here,
// the only wait is the explicit vmcnt(0) in front of each K-step barrier.  M0 = wave-uniform LDS byte address;
// the hardware adds lane*16.  (cdna_hip_programming.md §5.7: M0 written in the same statement that uses it.)
__device__ __forceinline__ void glds16(const void* src, unsigned lds_wave_byte_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(src), "s"(lds_wave_byte_addr) : "memory");
}
// Same, for kernels that use M0 for nothing else: declared clobbered instead of saved/restored (2 SALU less per DMA).
__device__ __forceinline__ void glds16_m0(const void* src, unsigned lds_wave_byte_addr) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(lds_wave_byte_addr) : "memory", "m0");
}
__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)(size_t)(__attribute__((address_space(3))) const void*)p;
}
// n / d for 0 <= n < 2^31 with quotient < 2^22, given inv ~ 1/d (float reciprocal + one-step correction)
__device__ __forceinline__ int fdiv(int n, int d, float inv) {
  int q = (int)((float)n * inv);
  const int r = n - q * d;
  q += (r >= d) - (r < 0);
  return q;
}

// CPRW = 16-byte chunks per LDS row: 8 -> 128-byte rows (K step 64 halfs), 4 -> 64-byte rows (K step 32 halfs).
template <int CPRW> __device__ __forceinline__ int swz(int row) {
  if constexpr (CPRW == 8) return (row >> 1) & 7;
  else return (-(row >> 2)) & 3;        // conflict-free for the ds_read_b128 lane groups with 64-byte rows
}
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// bias + activation (compile-time) + pack, fragment -> chunk-swizzled LDS tile (row = pixel, BN channels per row)
template <class T, int ACT, int BN, int MI, int NJ>
__device__ __forceinline__ void stage_tile(const ConvP& p, const f32x4 (&acc)[NJ][MI], char* tilep, int n0, int wm0, int wn0, int fr, int fg) {
  constexpr int ROWB = BN * 2, CPR = BN / 8;
  auto rowswz = [](int row) { if constexpr (CPR <= 16) return row / (16 / CPR); else return row * (CPR / 16); };
  const float osc = out_scale(p);
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int nl = wn0 + j * 16 + fg * 4, n = n0 + nl;
    float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f), s4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias && n < p.Cout) b4 = *reinterpret_cast<const float4*>(p.bias + n);
    if constexpr (ACT == 3) { if (n < p.Cout) s4 = *reinterpret_cast<const float4*>(p.slope + n); }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const f32x4 av = acc[j][i];
      float v0 = activate<T, ACT>(__builtin_fmaf(av[0], osc, b4.x)), v1 = activate<T, ACT>(__builtin_fmaf(av[1], osc, b4.y));
      float v2 = activate<T, ACT>(__builtin_fmaf(av[2], osc, b4.z)), v3 = activate<T, ACT>(__builtin_fmaf(av[3], osc, b4.w));
      if constexpr (ACT == 3) { v0 = v0 > 0.f ? v0 : s4.x * v0; v1 = v1 > 0.f ? v1 : s4.y * v1; v2 = v2 > 0.f ? v2 : s4.z * v2; v3 = v3 > 0.f ? v3 : s4.w * v3; }
      if constexpr (ACT == 4) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }   // staged path has no residual
      const int row = wm0 + i * 16 + fr;
      const int ch = (nl >> 3) ^ (rowswz(row) & (CPR - 1));
      *reinterpret_cast<uint2*>(tilep + row * ROWB + ch * 16 + (nl & 4) * 2) = make_uint2(pack2<T>(v0, v1), pack2<T>(v2, v3));
    }
  }
}

// bias + activation (compile-time: a runtime switch per value keeps the 64-fragment accumulator of the 256x256 kernel
// from being promoted to registers) + optional residual, 4 consecutive channels of one pixel stored from registers
template <class T, int ACT, int MI, int NJ>
__device__ __forceinline__ void direct_tile(const ConvP& p, const f32x4 (&acc)[NJ][MI], const long (&mrow)[MI], int nbase) {
  const float osc = out_scale(p);
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int n = nbase + j * 16;
    const bool nok = n < p.Cout;
    float bv[4] = {0.f, 0.f, 0.f, 0.f}, sv[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.bias && nok) { const float4 b4 = *reinterpret_cast<const float4*>(p.bias + n); bv[0] = b4.x; bv[1] = b4.y; bv[2] = b4.z; bv[3] = b4.w; }
    if constexpr (ACT == 3) { if (nok) { const float4 s4 = *reinterpret_cast<const float4*>(p.slope + n); sv[0] = s4.x; sv[1] = s4.y; sv[2] = s4.z; sv[3] = s4.w; } }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const f32x4 av = acc[j][i];
      const long m = mrow[i];
      if (nok && m >= 0) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] = activate<T, ACT>(__builtin_fmaf(av[e], osc, bv[e]));
          if constexpr (ACT == 3) v[e] = v[e] > 0.f ? v[e] : sv[e] * v[e];
        }
        if (p.res) {
          float rv[4];
          const size_t ri = (size_t)m * p.res_cstride + p.res_coff + n;
          if (p.res_f32) load4<float>(p.res, ri, rv); else load4<T>(p.res, ri, rv);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = rv[e] + v[e];
        }
        if constexpr (ACT == 4) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        const size_t oi = (size_t)m * p.out_cstride + p.out_coff + n;
        if (p.out_f32) store4<float>(p.out, oi, v); else store4<T>(p.out, oi, v);
      }
    }
  }
}

// Epilogue shared by the conv kernels.  acc[j][i] = 4 consecutive output channels (n0 + wn0 + 16j + 4*(lane>>4) + e) of
// tile pixel row = wm0 + 16i + (lane&15);  pix(row) -> flat output pixel index (b*Ho + ho)*Wo + wo, or -1 outside the image.
// 16-bit outputs without residual go through LDS (chunk-swizzled, the K stages are dead by now) so that every pixel's BN
// channels leave as 16-byte-per-lane, line-contiguous stores; f32 outputs / residual adds store from registers.
// MMAJOR: waves are numbered pixel-group-major (wave / WN = pixel group) instead of wave % WM (conv_phase_kernel).
template <class T, int BM, int BN, int WM, int MI, int NJ, int NT = 2 * BM, bool MMAJOR = false, class PixFn>
__device__ __forceinline__ void conv_epilogue(const ConvP& p, f32x4 (&acc)[NJ][MI], int n0, uint4* lds, PixFn pix) {
  constexpr int WN = NT / 64 / WM;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm0 = (MMAJOR ? wave / WN : wave % WM) * (BM / WM), wn0 = (MMAJOR ? wave % WN : wave / WM) * (BN / WN);
  const int fr = lane & 15, fg = lane >> 4;
  if constexpr (sizeof(T) == 2) {
    constexpr int ROWB = BN * 2;                       // epilogue tile row (bytes), chunk-swizzled, no padding
    constexpr int CPR = BN / 8;                        // 16-byte chunks per row
    auto rowswz = [](int row) { if constexpr (CPR <= 16) return row / (16 / CPR); else return row * (CPR / 16); };
    if (!p.res && !p.out_f32 && (p.Cout % 8 == 0) && (p.out_coff % 8 == 0) && (p.out_cstride % 8 == 0)) {
      __syncthreads();                                 // every wave is done reading the K stages
      char* tilep = reinterpret_cast<char*>(lds);
      if (p.act == 1) stage_tile<T, 1, BN, MI, NJ>(p, acc, tilep, n0, wm0, wn0, fr, fg);
      else if (p.act == 2) stage_tile<T, 2, BN, MI, NJ>(p, acc, tilep, n0, wm0, wn0, fr, fg);
      else if (p.act == 3) stage_tile<T, 3, BN, MI, NJ>(p, acc, tilep, n0, wm0, wn0, fr, fg);
      else if (p.act == 4) stage_tile<T, 4, BN, MI, NJ>(p, acc, tilep, n0, wm0, wn0, fr, fg);
      else stage_tile<T, 0, BN, MI, NJ>(p, acc, tilep, n0, wm0, wn0, fr, fg);
      __syncthreads();
      T* outp = reinterpret_cast<T*>(p.out) + p.out_coff + n0;
#pragma unroll
      for (int q = 0; q < (BM * CPR + NT - 1) / NT; ++q) {
        const int idx = tid + NT * q, row = idx / CPR, ch = idx - row * CPR;
        if ((BM * CPR) % NT != 0 && idx >= BM * CPR) break;   // tiles with fewer 16-byte chunks than threads (32 x 32 on 256 threads)
        const long m = pix(row);
        if (m >= 0 && n0 + ch * 8 < p.Cout)
          *reinterpret_cast<uint4*>(outp + (size_t)m * p.out_cstride + ch * 8) =
              *reinterpret_cast<const uint4*>(tilep + row * ROWB + (ch ^ (rowswz(row) & (CPR - 1))) * 16);
      }
      return;
    }
  }
  // direct path: bias -> activation -> (+residual) -> store 4 consecutive channels of one pixel
  long mrow[MI];
#pragma unroll
  for (int i = 0; i < MI; ++i) mrow[i] = pix(wm0 + i * 16 + fr);
  if (p.act == 1) direct_tile<T, 1, MI, NJ>(p, acc, mrow, n0 + wn0 + fg * 4);
  else if (p.act == 2) direct_tile<T, 2, MI, NJ>(p, acc, mrow, n0 + wn0 + fg * 4);
  else if (p.act == 3) direct_tile<T, 3, MI, NJ>(p, acc, mrow, n0 + wn0 + fg * 4);
  else if (p.act == 4) direct_tile<T, 4, MI, NJ>(p, acc, mrow, n0 + wn0 + fg * 4);
  else direct_tile<T, 0, MI, NJ>(p, acc, mrow, n0 + wn0 + fg * 4);
}

}  // namespace cc

// CLIP tower kernels other than the GEMMs: LayerNorm, patchify, token assembly, embedding gather,
// fused attention (MFMA for f16/bf16, exact-f32 VALU for parity mode), L2 normalise.
// Reference: models/objects.py:94-186 (tinygrad tensor ops); GEMMs go through conv_mfma.hip.
#include "kernels.h"
#include "mfma.h"

namespace cc {

ConvP gemm_params(const void* A, int lda, int M, int K, const void* W, int ldw, const float* bias, int N, void* out, int ldc,
                  int out_f32, int act, const void* res, int ldres, int res_f32) {
  ConvP c{};
  c.s0 = Src{A, 1, M, lda, 0, K, 0};
  c.s1 = Src{A, 1, 1, 0, 0, 0, 0};
  c.B = 1; c.Hin = 1; c.Win = M; c.Cin = K; c.Ho = 1; c.Wo = M; c.Cout = N;
  c.ks = 1; c.stride = 1; c.pad = 0; c.Ktot = K; c.Kw = ldw;
  c.w = W; c.bias = bias; c.out = out; c.out_cstride = ldc; c.out_coff = 0; c.out_f32 = out_f32;
  c.res = res; c.res_cstride = ldres; c.res_coff = 0; c.res_f32 = res_f32; c.act = act;
  return c;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---- LayerNorm: one wave per row, D <= 1024 ------------------------------------------------------
// VEC: D is a multiple of 4 and the rows are 16-byte aligned - the lane owns four float4 groups, 256 elements apart: every load is one
// 1 KB line-contiguous wave access (4 instead of 16 load instructions per row), gamma / beta likewise, and the row leaves as 8-byte
// (16-bit storage) or 16-byte stores instead of sixteen 2-byte ones.  Same arithmetic per element, same reduction order across the lanes'
// partial sums is NOT the same as the scalar form's (lane l sums elements 4l.., not l + 64 i) - an f32 rounding-level difference.
template <class T, bool VEC>
__global__ __launch_bounds__(256) void layernorm_kernel(const LnP p) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= p.rows) return;
  const float* x = p.in + (long)(p.row_index ? p.row_index[row] : row) * p.in_row_stride;
  float v[16];
  float s = 0.f;
  if constexpr (VEC) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int j = (lane + 64 * i) * 4;
      const float4 t = j < p.D ? *reinterpret_cast<const float4*>(x + j) : make_float4(0.f, 0.f, 0.f, 0.f);
      v[4 * i] = t.x; v[4 * i + 1] = t.y; v[4 * i + 2] = t.z; v[4 * i + 3] = t.w;
      s += (t.x + t.y) + (t.z + t.w);
    }
  } else {
#pragma unroll
    for (int i = 0; i < 16; ++i) { const int j = lane + 64 * i; v[i] = j < p.D ? x[j] : 0.f; s += v[i]; }
  }
  const float mean = wave_sum(s) / (float)p.D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) { const int j = VEC ? (lane + 64 * (i >> 2)) * 4 + (i & 3) : lane + 64 * i; const float d = j < p.D ? v[i] - mean : 0.f; q += d * d; }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)p.D + 1e-5f);
  if constexpr (VEC) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int j = (lane + 64 * i) * 4;
      if (j < p.D) {
        const float4 w4 = *reinterpret_cast<const float4*>(p.w + j), b4 = *reinterpret_cast<const float4*>(p.b + j);
        const float y0 = (v[4 * i] - mean) * rstd * w4.x + b4.x, y1 = (v[4 * i + 1] - mean) * rstd * w4.y + b4.y;
        const float y2 = (v[4 * i + 2] - mean) * rstd * w4.z + b4.z, y3 = (v[4 * i + 3] - mean) * rstd * w4.w + b4.w;
        if (p.out_f32) *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.out) + (size_t)row * p.D + j) = make_float4(y0, y1, y2, y3);
        else if constexpr (sizeof(T) == 2) *reinterpret_cast<uint2*>(reinterpret_cast<T*>(p.out) + (size_t)row * p.D + j) = make_uint2(pack2<T>(y0, y1), pack2<T>(y2, y3));
        else *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.out) + (size_t)row * p.D + j) = make_float4(y0, y1, y2, y3);
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int j = lane + 64 * i;
      if (j < p.D) {
        const float y = (v[i] - mean) * rstd * p.w[j] + p.b[j];
        if (p.out_f32) reinterpret_cast<float*>(p.out)[(size_t)row * p.D + j] = y;
        else reinterpret_cast<T*>(p.out)[(size_t)row * p.D + j] = from_f32<T>(y);
      }
    }
  }
}
void launch_layernorm(int dt, const LnP& p, hipStream_t stream) {
  CC_CHECK(p.D <= 1024, "layernorm: D > 1024");
  const dim3 grid((p.rows + 3) / 4), block(256);
  const bool vec = p.D % 4 == 0 && p.in_row_stride % 4 == 0 && (((uintptr_t)p.in | (uintptr_t)p.out | (uintptr_t)p.w | (uintptr_t)p.b) & 15) == 0;
  if (dt == F32) { if (vec) hipLaunchKernelGGL((layernorm_kernel<float, true>), grid, block, 0, stream, p); else hipLaunchKernelGGL((layernorm_kernel<float, false>), grid, block, 0, stream, p); }
  else if (dt == F16) { if (vec) hipLaunchKernelGGL((layernorm_kernel<f16_t, true>), grid, block, 0, stream, p); else hipLaunchKernelGGL((layernorm_kernel<f16_t, false>), grid, block, 0, stream, p); }
  else { if (vec) hipLaunchKernelGGL((layernorm_kernel<bf16_t, true>), grid, block, 0, stream, p); else hipLaunchKernelGGL((layernorm_kernel<bf16_t, false>), grid, block, 0, stream, p); }
  CC_HIP(hipGetLastError());
}

// ---- patchify: conv 14x14/s14 as a GEMM operand (objects.py:95-97) --------------------------------
template <class T>
__global__ __launch_bounds__(256) void patchify_kernel(const PatchP p) {
  const int g = p.S / p.patch, pp = p.patch * p.patch;
  const size_t total = (size_t)p.B * g * g * p.Kpad;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int k = (int)(idx % p.Kpad);
  const size_t row = idx / p.Kpad;
  float v = 0.f;
  if (k < 3 * pp) {
    const int c = k / pp, r = k - c * pp, kh = r / p.patch, kw = r - kh * p.patch;
    const int b = (int)(row / (g * g)), pi = (int)(row - (size_t)b * g * g), py = pi / g, px = pi - py * g;
    v = p.x[(((size_t)b * 3 + c) * p.S + py * p.patch + kh) * p.S + px * p.patch + kw];
  }
  reinterpret_cast<T*>(p.out)[idx] = from_f32<T>(v);
}
void launch_patchify(int dt, const PatchP& p, hipStream_t stream) {
  const int g = p.S / p.patch;
  const size_t total = (size_t)p.B * g * g * p.Kpad;
  const dim3 grid((unsigned)((total + 255) / 256)), block(256);
  if (dt == F32) hipLaunchKernelGGL(patchify_kernel<float>, grid, block, 0, stream, p);
  else if (dt == F16) hipLaunchKernelGGL(patchify_kernel<f16_t>, grid, block, 0, stream, p);
  else hipLaunchKernelGGL(patchify_kernel<bf16_t>, grid, block, 0, stream, p);
  CC_HIP(hipGetLastError());
}

// ---- token assembly + ln_pre ------------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(256) void assemble_ln_kernel(const AssembleP p) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= p.B * p.L) return;
  const int b = row / p.L, t = row - b * p.L;
  const T* patch = reinterpret_cast<const T*>(p.patches) + ((size_t)b * (p.L - 1) + (t > 0 ? t - 1 : 0)) * p.D;
  float v[16];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int j = lane + 64 * i;
    v[i] = j < p.D ? (t == 0 ? p.cls[j] : to_f32<T>(patch[j])) + p.pos[(size_t)t * p.D + j] : 0.f;
    s += v[i];
  }
  const float mean = wave_sum(s) / (float)p.D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) { const int j = lane + 64 * i; const float d = j < p.D ? v[i] - mean : 0.f; q += d * d; }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)p.D + 1e-5f);
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int j = lane + 64 * i;
    if (j < p.D) p.out[(size_t)row * p.D + j] = (v[i] - mean) * rstd * p.w[j] + p.b[j];
  }
}
void launch_assemble_ln(int dt, const AssembleP& p, hipStream_t stream) {
  CC_CHECK(p.D <= 1024, "assemble: D > 1024");
  const dim3 grid((p.B * p.L + 3) / 4), block(256);
  if (dt == F32) hipLaunchKernelGGL(assemble_ln_kernel<float>, grid, block, 0, stream, p);
  else if (dt == F16) hipLaunchKernelGGL(assemble_ln_kernel<f16_t>, grid, block, 0, stream, p);
  else hipLaunchKernelGGL(assemble_ln_kernel<bf16_t>, grid, block, 0, stream, p);
  CC_HIP(hipGetLastError());
}

// ---- token embedding gather + positional -----------------------------------------------------------
__global__ __launch_bounds__(256) void embed_kernel(const EmbedP p) {
  const size_t total = (size_t)p.B * p.L * p.D;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int j = (int)(idx % p.D);
  const size_t row = idx / p.D;
  const int t = (int)(row % p.L);
  p.out[idx] = p.table[(size_t)p.tokens[row] * p.D + j] + p.pos[(size_t)t * p.D + j];
}
void launch_embed(const EmbedP& p, hipStream_t stream) {
  const size_t total = (size_t)p.B * p.L * p.D;
  hipLaunchKernelGGL(embed_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, p);
  CC_HIP(hipGetLastError());
}

// ---- L2 normalise -------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void l2norm_kernel(const NormP p) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= p.rows) return;
  float* x = p.x + (size_t)row * p.D;
  float s = 0.f;
  for (int j = lane; j < p.D; j += 64) s += x[j] * x[j];
  const float n = sqrtf(wave_sum(s)) + p.eps;
  for (int j = lane; j < p.D; j += 64) x[j] = x[j] / n;
}
void launch_l2norm(const NormP& p, hipStream_t stream) {
  hipLaunchKernelGGL(l2norm_kernel, dim3((p.rows + 3) / 4), dim3(256), 0, stream, p);
  CC_HIP(hipGetLastError());
}

// ---- attention, 16-bit storage: S^T = K Q^T and O^T = V^T P^T on MFMA --------------------------------
// One workgroup = (image b, head h); wave w owns the 16-query tiles w, w+4, ...  The whole K (rows of 128 B, chunk-swizzled like
// the GEMM tiles) and V (ROW-MAJOR, 160-byte rows) of the head sit in LDS: L <= 288 keys (ViT-L/14 has 257, the text tower 77) -> no
// online-softmax tiling over keys is needed.  Computing the TRANSPOSED scores puts one query per lane column (lane&15), so the
// softmax reduction is 4*NF register values + two cross-lane steps, and the exponentiated P is already in MFMA B-operand layout.
//
// Round 4: (1) V stays row-major and the PV A operand (V^T: eight keys of one output dimension per lane) is gathered by gfx950's
// transposing LDS read - ds_read_b64_tr_b16: the sixteen lanes of a group point at the rows of a [4 keys][16 dims] block (lane i:
// key i>>2, dims 4(i&3)..+3) and lane i receives dimension i of the four keys (checked on the chip: tools/dev/tr_read_test.hip).
// Round 3 built V^T with 72 two-byte LDS writes per thread per (image, head) - banks 8-way conflicted (SQ_LDS_BANK_CONFLICT 8.8 % of
// the kernel's wave cycles) - now nine 16-byte writes.  Row pitch 160 B: the eight rows a 32-lane half touches start 40 dwords
// apart = eight distinct multiples of 8 banks (mod 64), i.e. conflict-free.  (2) K holds NFK <= NF fragments (17 for 257 keys), so
// that two blocks still share a CU (80.9 KB each).  (3) With 257 = 16 x 16 + 1 queries the seventeenth tile holds ONE query, and as
// wave 0's fifth tile it made every block 25 % longer than its waves' average: when tiles % 4 == 1 the last tile is computed by all
// four waves, each over a quarter of the keys (local max / sum / partial O), merged through LDS (the K region, dead by then).
template <class T> __device__ __forceinline__ uint2 lds_read_tr16(const T* p) {
  typedef short s16x4 __attribute__((ext_vector_type(4)));
  const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
  return __builtin_bit_cast(uint2, v);
}

// LFULL >= 0: the token count is known at compile time to the extent that matters - fragments below LFULL hold real keys only, fragment
// LFULL is the partial one (keys >= L masked), fragments above it are padding - and CAUSAL is a compile-time flag: the score / softmax
// section is then straight-line code.  With L and causal as run-time values it compiled to ~130 wave-uniform branches and 220 s_waitcnt
// per tile pair (every fragment asks "do I need masking?", "am I padding?") and ran at a third of the speed (round 4: the softmax section
// alone took 124 of the kernel's 225 us at ViT-L/14's shape, cc_attn_bench ablations).  LFULL = -1 keeps the run-time form for other lengths.
// NQ = query tiles a wave works on at once (1 or 2): with two, every K / V fragment read from LDS feeds two MFMAs and the two tiles'
// dependent chains (36 MFMAs -> max -> 72 exp -> 36 MFMAs) interleave - the kernel is bound by those latencies, not by instruction issue.
template <class T, int NF, int NFK, int LFULL = -1, int CAUSAL = -1, int NQ = 1>    // NF = padded keys / 16 (even: PV walks 32 keys per step); NFK = key fragments that can hold a real key
__global__ __launch_bounds__(256, 2) void attn_mfma_kernel(const AttnP p) {   // two blocks per CU: one wave can run MFMAs while the other does its softmax
  constexpr int LP = NF * 16, VP = 80;                                   // padded keys; V row pitch in elements (64 dims + 16: see above)
  static_assert(NF % 2 == 0 && NFK <= NF && NFK >= NF - 1, "PV walks pairs of key fragments");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint4* ldsK = reinterpret_cast<uint4*>(smem);                         // [NFK * 16][8 chunks]
  T* ldsV = reinterpret_cast<T*>(smem + (size_t)NFK * 16 * 128);        // [LP][VP]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.z, h = blockIdx.y;
  const int D3 = 3 * p.D;
  const T* base = reinterpret_cast<const T*>(p.qkv) + (size_t)b * p.L * D3 + h * 64;

  // stage K and V of this (image, head) ONCE; the workgroup then walks all query tiles.  All loads of a thread are issued before the
  // first LDS write and none is conditional (keys past L are clamped and zeroed afterwards): with a load inside an `if (key < L)` per
  // iteration hipcc branched around every load and waited for it before the next one (cdna_hip_programming.md, section 5, trap (c)).
  if (!(p.abl & 2)) {
    constexpr int NIT = LP * 8 / 256;
    static_assert(LP * 8 % 256 == 0, "whole passes of the 256 threads");
    uint4 kv[NIT], vv[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = tid + 256 * it, key = idx >> 3, chunk = idx & 7, kc = key < p.L ? key : p.L - 1;
      kv[it] = *reinterpret_cast<const uint4*>(base + (size_t)kc * D3 + p.D + chunk * 8);
      vv[it] = *reinterpret_cast<const uint4*>(base + (size_t)kc * D3 + 2 * p.D + chunk * 8);
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = tid + 256 * it, key = idx >> 3, chunk = idx & 7;
      if (key >= p.L) { kv[it] = make_uint4(0, 0, 0, 0); vv[it] = make_uint4(0, 0, 0, 0); }
      if (key < NFK * 16) ldsK[key * 8 + (chunk ^ ((key >> 1) & 7))] = kv[it];
      *reinterpret_cast<uint4*>(ldsV + key * VP + chunk * 8) = vv[it];
    }
  }
  __syncthreads();
  if (p.abl & 1) { if (tid == 0 && ldsK[0].x == 0x12345678u) reinterpret_cast<unsigned*>(p.ctx)[0] = 1u; return; }
  if (p.abl & 32) return;
  const int ql = lane & 15, g = lane >> 4;
  const float c = p.scale * 1.4426950408889634f;
  // PV A operand of (32-key block f2, output-dimension block d): k slot (g, j): j < 4 -> key 32 f2 + 4g + j, j >= 4 -> key 32 f2 + 16 + 4g
  // + (j - 4) - the order the scores sit in the registers - as two transposing reads of [4 keys][16 dims] blocks
  const T* vbase = ldsV + (4 * g + (ql >> 2)) * VP + 4 * (ql & 3);
  auto vfrag = [&](int f2, int d) {
    const uint2 lo = lds_read_tr16(vbase + (f2 * 32) * VP + d * 16), hi = lds_read_tr16(vbase + (f2 * 32 + 16) * VP + d * 16);
    return make_uint4(lo.x, lo.y, hi.x, hi.y);
  };
  // scores of key fragments [F0, F1) against the query fragment qf: masked, exponentiated relative to the (returned) maximum over those
  // keys; `sum` = the row sum of the exponentials.  s[f][r] = <k_{16f+4g+r}, q_{ql}>.  The softmax is the VALU-bound part of this
  // kernel (72 values per lane against 72 MFMAs per tile), so it is kept to max / fma / v_exp_f32 / add per value: the row maximum is
  // taken on the raw scores (scale > 0 keeps the order), scale*log2(e) is folded into one fma feeding the hardware exp2, and masking
  // code runs only for fragments that actually contain padded or future keys.
  // (the lambdas below take the number of query tiles NT <= NQ they work on as a compile-time argument: NQ in the main loop, 1 in the split tile)
  auto scores = [&](auto nt_c, const uint4 (&qf)[NQ][2], const int (&q)[NQ], auto f0_c, auto f1_c, f32x4 (&s)[NQ][NF], float (&mx)[NQ], float (&sum)[NQ]) {
    constexpr int NT = decltype(nt_c)::value, F0 = decltype(f0_c)::value, F1 = decltype(f1_c)::value;
    constexpr bool RT = LFULL < 0;                              // run-time token count / causal flag
    const bool causal = RT ? (p.causal != 0) : (CAUSAL != 0);
#pragma unroll
    for (int f = F0; f < F1; ++f) {
#pragma unroll
      for (int t = 0; t < NT; ++t) s[t][f] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (f < NFK && (RT || f <= LFULL)) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const int row = f * 16 + ql;
          const uint4 kf = ldsK[row * 8 + ((ks * 4 + g) ^ ((row >> 1) & 7))];
#pragma unroll
          for (int t = 0; t < NT; ++t) Mma<T>::run(kf, qf[t][ks], s[t][f]);
        }
      }
      if ((f - F0) % 3 == 2) asm volatile("" ::: "memory");    // keep at most 6 K fragments in flight: hoisting all 36 reads costs 144 VGPRs
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      float m_ = -INFINITY;
#pragma unroll
      for (int f = F0; f < F1; ++f) {
        if (!RT && f > LFULL) continue;                        // padding only: out of the maximum and the sum, p = 0
        if (RT ? (causal || f * 16 + 16 > p.L) : (CAUSAL != 0 || f == LFULL)) {   // run time: wave-uniform branch; compile time: no branch
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int key = f * 16 + g * 4 + r;
            const bool dead = ((RT || f == LFULL) && key >= p.L) || (causal && key > q[t]);
            s[t][f][r] = dead ? -INFINITY : s[t][f][r];
          }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) m_ = fmaxf(m_, s[t][f][r]);
      }
      m_ = fmaxf(m_, __shfl_xor(m_, 16, 64));
      m_ = fmaxf(m_, __shfl_xor(m_, 32, 64));
      mx[t] = m_;
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const float mc = mx[t] == -INFINITY ? 0.f : mx[t] * c;    // a key range that is masked out entirely (split tile, causal): every p = 0
      float l_ = 0.f;
#pragma unroll
      for (int f = F0; f < F1; ++f) {
        if (RT ? (f * 16 >= p.L) : (f > LFULL)) { s[t][f] = f32x4{0.f, 0.f, 0.f, 0.f}; continue; }      // fragment of padding only: p = 0
#pragma unroll
        for (int r = 0; r < 4; ++r) { const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(s[t][f][r], c, -mc)); s[t][f][r] = e; l_ += e; }
      }
      l_ += __shfl_xor(l_, 16, 64);
      l_ += __shfl_xor(l_, 32, 64);
      sum[t] = l_;
    }
  };
  // O^T += V^T P^T over the 32-key blocks [B0, B1)
  auto pv = [&](auto nt_c, auto b0_c, auto b1_c, const f32x4 (&s)[NQ][NF], f32x4 (&o)[NQ][4]) {
    constexpr int NT = decltype(nt_c)::value, B0 = decltype(b0_c)::value, B1 = decltype(b1_c)::value;
#pragma unroll
    for (int f2 = B0; f2 < B1; ++f2) {
      uint4 pf[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t)
        pf[t] = make_uint4(pack2<T>(s[t][2 * f2][0], s[t][2 * f2][1]), pack2<T>(s[t][2 * f2][2], s[t][2 * f2][3]),
                           pack2<T>(s[t][2 * f2 + 1][0], s[t][2 * f2 + 1][1]), pack2<T>(s[t][2 * f2 + 1][2], s[t][2 * f2 + 1][3]));
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const uint4 vf = vfrag(f2, d);
#pragma unroll
        for (int t = 0; t < NT; ++t) Mma<T>::run(vf, pf[t], o[t][d]);
      }
      if (f2 & 1) asm volatile("" ::: "memory");
    }
  };
  auto store_row = [&](int q, const f32x4 (&o)[4], float inv) {
    T* out = reinterpret_cast<T*>(p.ctx) + ((size_t)b * p.L + q) * p.D + h * 64;
#pragma unroll
    for (int d = 0; d < 4; ++d)
      *reinterpret_cast<uint2*>(out + d * 16 + g * 4) = make_uint2(pack2<T>(o[d][0] * inv, o[d][1] * inv), pack2<T>(o[d][2] * inv, o[d][3] * inv));
  };

  // Q fragments are loaded one round AHEAD (unconditionally: rows past L re-read row L-1, their results are never stored): a
  // wave otherwise opens every tile with a dependent global load and nothing to do while it is in flight.
  uint4 qf[NQ][2], qn[NQ][2];
  auto load_q = [&](int q0, uint4 (&dst)[2]) {
    const int qc = q0 + ql < p.L ? q0 + ql : p.L - 1;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) dst[ks] = *reinterpret_cast<const uint4*>(base + (size_t)qc * D3 + (ks * 4 + g) * 8);
  };
  const int tiles = (p.L + 15) >> 4;
  const bool split_last = (tiles & 3) == 1 && tiles > 1;         // block-uniform: the last tile is shared by the four waves
  const int own_end = (split_last ? tiles - 1 : tiles) * 16;     // queries below this belong to whole-tile rounds
  // round i of wave w: tiles w + 4 (NQ i + t), t < NQ (the launcher picks NQ = 2 only when the whole-tile rounds come in pairs)
#pragma unroll
  for (int t = 0; t < NQ; ++t) load_q(wave * 16 + 64 * t, qf[t]);
  for (int q0 = wave * 16; q0 < own_end; q0 += 64 * NQ) {
    int q[NQ];
#pragma unroll
    for (int t = 0; t < NQ; ++t) q[t] = q0 + 64 * t + ql;
    const bool more = q0 + 64 * NQ < own_end;
#pragma unroll
    for (int t = 0; t < NQ; ++t) load_q(more ? q0 + 64 * (NQ + t) : (split_last ? (tiles - 1) * 16 : q0), qn[t]);
    f32x4 s[NQ][NF];
    float mx[NQ], sum[NQ];
    scores(std::integral_constant<int, NQ>{}, qf, q, std::integral_constant<int, 0>{}, std::integral_constant<int, NF>{}, s, mx, sum);
    f32x4 o[NQ][4];
#pragma unroll
    for (int t = 0; t < NQ; ++t)
#pragma unroll
      for (int d = 0; d < 4; ++d) o[t][d] = f32x4{0.f, 0.f, 0.f, 0.f};
    pv(std::integral_constant<int, NQ>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, NF / 2>{}, s, o);
#pragma unroll
    for (int t = 0; t < NQ; ++t) {
      if (q[t] < p.L) store_row(q[t], o[t], 1.0f / sum[t]);
      qf[t][0] = qn[t][0]; qf[t][1] = qn[t][1];
    }
  }
  if (!split_last) return;
  // ---- the last tile, key range split four ways: wave w takes the 32-key blocks [BLK0(w), BLK0(w+1)) ------------------------------------
  constexpr int NB = NF / 2, BQ = NB / 4, BR = NB % 4;             // 9 blocks -> 3, 2, 2, 2
  {
    const int q0 = (tiles - 1) * 16;
    int q[NQ];
#pragma unroll
    for (int t = 0; t < NQ; ++t) q[t] = q0 + ql;
    if (wave * 16 >= own_end) load_q(q0, qf[0]);                   // a wave that ran no whole tile (L <= 64) has not prefetched it
    f32x4 s[NQ][NF], o[NQ][4];
    float mx[NQ], sum[NQ];
    mx[0] = -INFINITY; sum[0] = 0.f;
#pragma unroll
    for (int d = 0; d < 4; ++d) o[0][d] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto part = [&](auto w_c) {
      constexpr int W = decltype(w_c)::value, B0 = W * BQ + (W < BR ? W : BR), B1 = B0 + BQ + (W < BR ? 1 : 0);
      scores(std::integral_constant<int, 1>{}, qf, q, std::integral_constant<int, 2 * B0>{}, std::integral_constant<int, 2 * B1>{}, s, mx, sum);
      pv(std::integral_constant<int, 1>{}, std::integral_constant<int, B0>{}, std::integral_constant<int, B1>{}, s, o);
    };
    if (wave == 0) part(std::integral_constant<int, 0>{});
    else if (wave == 1) part(std::integral_constant<int, 1>{});
    else if (wave == 2) part(std::integral_constant<int, 2>{});
    else part(std::integral_constant<int, 3>{});
    __syncthreads();                                               // nobody reads K any more: its region takes the partial results
    float* part_o = reinterpret_cast<float*>(smem);                // [4 waves][4 d][64 lanes] float4  = 16 KB
    float* part_ms = part_o + 4 * 4 * 64 * 4;                      // [4 waves][16 queries][2]
#pragma unroll
    for (int d = 0; d < 4; ++d) *reinterpret_cast<float4*>(part_o + ((wave * 4 + d) * 64 + lane) * 4) = make_float4(o[0][d][0], o[0][d][1], o[0][d][2], o[0][d][3]);
    if (g == 0) { part_ms[(wave * 16 + ql) * 2] = mx[0]; part_ms[(wave * 16 + ql) * 2 + 1] = sum[0]; }
    __syncthreads();
    // wave w merges output-dimension block d = w of every query of the tile: o = sum_w o_w 2^((m_w - m) c) / sum_w l_w 2^((m_w - m) c)
    float m = -INFINITY;
#pragma unroll
    for (int w = 0; w < 4; ++w) m = fmaxf(m, part_ms[(w * 16 + ql) * 2]);
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    float l = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float mw = part_ms[(w * 16 + ql) * 2];
      const float fac = mw == -INFINITY ? 0.f : __builtin_amdgcn_exp2f((mw - m) * c);
      l += part_ms[(w * 16 + ql) * 2 + 1] * fac;
      const float4 ow = *reinterpret_cast<const float4*>(part_o + ((w * 4 + wave) * 64 + lane) * 4);
      acc[0] += ow.x * fac; acc[1] += ow.y * fac; acc[2] += ow.z * fac; acc[3] += ow.w * fac;
    }
    if (q[0] < p.L) {
      const float inv = 1.0f / l;
      T* out = reinterpret_cast<T*>(p.ctx) + ((size_t)b * p.L + q[0]) * p.D + h * 64;
      *reinterpret_cast<uint2*>(out + wave * 16 + g * 4) = make_uint2(pack2<T>(acc[0] * inv, acc[1] * inv), pack2<T>(acc[2] * inv, acc[3] * inv));
    }
  }
}

// ---- attention, any storage type, f32 VALU math: one wave per query (parity mode / fallback) ------------
template <class T>
__global__ __launch_bounds__(256) void attn_simple_kernel(const AttnP p) {
  constexpr int LMAX = 320;
  __shared__ float qs[4][64];
  __shared__ float ps[4][LMAX];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.z, h = blockIdx.y;
  const int q = min(blockIdx.x * 4 + wave, p.L - 1);
  const bool live = blockIdx.x * 4 + wave < p.L;
  const int D3 = 3 * p.D;
  const T* base = reinterpret_cast<const T*>(p.qkv) + (size_t)b * p.L * D3 + h * 64;
  qs[wave][lane] = to_f32<T>(base[(size_t)q * D3 + lane]);
  __syncthreads();
  float mx = -INFINITY;
  for (int j = lane; j < p.L; j += 64) {
    const T* k = base + (size_t)j * D3 + p.D;
    float dot = 0.f;
#pragma unroll 8
    for (int d = 0; d < 64; ++d) dot = fmaf(qs[wave][d], to_f32<T>(k[d]), dot);
    float v = dot * p.scale;
    if (p.causal && j > q) v = -INFINITY;
    ps[wave][j] = v;
    mx = fmaxf(mx, v);
  }
  mx = wave_max(mx);
  float sum = 0.f;
  for (int j = lane; j < p.L; j += 64) { const float e = expf(ps[wave][j] - mx); ps[wave][j] = e; sum += e; }
  sum = wave_sum(sum);
  __syncthreads();
  float o = 0.f;
  for (int j = 0; j < p.L; ++j) o = fmaf(ps[wave][j] / sum, to_f32<T>(base[(size_t)j * D3 + 2 * p.D + lane]), o);
  if (live) reinterpret_cast<T*>(p.ctx)[((size_t)b * p.L + q) * p.D + h * 64 + lane] = from_f32<T>(o);
}

template <class T, int NF, int NFK, int LFULL = -1, int CAUSAL = -1, int NQ = 1> static void launch_attn_mfma(const AttnP& p, hipStream_t stream) {
  const size_t lds = std::max((size_t)NFK * 16 * 128, (size_t)(4 * 4 * 64 * 4 + 4 * 16 * 2) * 4) + (size_t)NF * 16 * 80 * sizeof(T);
  static PerDevice once;                               // the attribute is per device (common.h)
  if (once.first(once.index())) {
    CC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_mfma_kernel<T, NF, NFK, LFULL, CAUSAL, NQ>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  }
  if (p.abl & 64) {                                    // development: how many blocks share a CU
    int nb = 0; CC_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(attn_mfma_kernel<T, NF, NFK, LFULL, CAUSAL, NQ>), 256, lds));
    fprintf(stderr, "[clearcam] attn_mfma_kernel<NF=%d, NFK=%d, LFULL=%d>: %zu bytes of LDS, %d blocks per CU\n", NF, NFK, LFULL, lds, nb);
  }
  hipLaunchKernelGGL((attn_mfma_kernel<T, NF, NFK, LFULL, CAUSAL, NQ>), dim3(1, p.H, p.B), dim3(256), lds, stream, p);
}
template <class T> static void launch_attn_t(const AttnP& p, hipStream_t stream) {
  // the shapes the reference runs get straight-line instantiations: ViT-L/14's 257 tokens, ViT-B/32's 50, the text tower's 77 causal ones
  if (p.L == 257 && !p.causal) {                       // sixteen whole tiles = two pairs per wave + the split seventeenth
    if (p.abl & 128) launch_attn_mfma<T, 18, 17, 16, 0, 1>(p, stream); else launch_attn_mfma<T, 18, 17, 16, 0, 2>(p, stream);
    return;
  }
  if (p.L == 50 && !p.causal) { launch_attn_mfma<T, 6, 5, 3, 0>(p, stream); return; }
  if (p.L == 77 && p.causal) { launch_attn_mfma<T, 6, 5, 4, 1>(p, stream); return; }
  // any other length: K keeps only the fragments that can hold a real key (17 of 18 up to 272 keys: two blocks per CU need <= 80 KB each)
  if (p.L <= 80) launch_attn_mfma<T, 6, 5>(p, stream);
  else if (p.L <= 96) launch_attn_mfma<T, 6, 6>(p, stream);
  else if (p.L <= 272) launch_attn_mfma<T, 18, 17>(p, stream);
  else launch_attn_mfma<T, 18, 18>(p, stream);
}

void launch_attention(int dt, const AttnP& p, hipStream_t stream) {
  CC_CHECK(p.D == p.H * 64, "attention: head dim must be 64");
  CC_CHECK(p.L <= 288, "attention: more than 288 tokens");
  if (dt == F32) {
    hipLaunchKernelGGL(attn_simple_kernel<float>, dim3((p.L + 3) / 4, p.H, p.B), dim3(256), 0, stream, p);
  } else if (dt == F16) launch_attn_t<f16_t>(p, stream);
  else launch_attn_t<bf16_t>(p, stream);
  CC_HIP(hipGetLastError());
}

}  // namespace cc

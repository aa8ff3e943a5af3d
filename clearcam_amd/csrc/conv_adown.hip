// ADown's first branch in one launch: the 3x3 stride-2 convolution reads the 2x2 stride-1 AVERAGE of its source
// (detection/yolov9.py:45-49: `x.avg_pool2d(2, 1, 0)` -> chunk -> `cv1`, a 3x3 s2 p1 Conv) without the averaged map ever being
// written to or read back from HBM.
//
// Same implicit GEMM as conv_mfma_kernel (128 pixels x BN channels x 64 K per step, 4 waves of 64 x BN/2, two LDS stages, weights by
// LDS-DMA, chunk-swizzled 128-byte rows, the shared epilogue) with ONE difference: the activation half of a stage is not DMA'd but
// built in registers.  A thread owns one 16-byte channel chunk of four tile rows per K step; for each it loads the four source
// pixels of the 2x2 window (16 B each), sums them in the order pool_vec_kernel<T, 2, 0> does (((0 + x00) + x01) + x10) + x11 in
// f32, scales by 1/4, rounds to T and writes the chunk where the DMA would have put it.  The values the MFMAs see are therefore
// bit for bit the ones the two-launch path reads back from HBM, the K order is (tap, channel) like every kernel on the default
// path: results are IDENTICAL to pool + conv (tests/test_gpu_yolo.py::test_fused_adown_equals_unfused), whatever the batch size.
//
// Schedule per K step t (the loads of step t+1 were issued a step earlier):
//     s_waitcnt vmcnt(16) ; barrier     weights of step t have landed (the 16 window loads of step t+1 may still be in flight)
//     weight DMA of step t+1 -> the other stage
//     MFMAs of step t
//     average the windows of step t+1 -> ds_write into the other stage; issue the window loads of step t+2
// Measured (MI355X, batch 64, bf16): SLOWER than pool + conv on every ADown of YOLOv9-C - 0.497 ms against 0.182 + 0.199 for the
// 128-channel half at 160x160, 0.332 against 0.093 + 0.135 for 256 channels at 80x80 - because the loader pulls each source chunk
// through L2 nine times (four window pixels x 2.25 taps per pixel) where the conv over the materialised map pulls each averaged chunk
// 2.25 times and the pool runs at HBM speed; a patch-resident form would have to hold 561 averaged pixels x all channels (144 KB at
// 128 channels) to keep the (tap, channel) order.  So the builder leaves it OFF (CLEARCAM_FUSE_ADOWN=1 enables it); it stays as the
// tested answer to "fold the average into the stride-2 conv's loader".
// Pooled positions outside the averaged map (the conv's zero padding) and rows past the last pixel read a 16-byte zero page four
// times: no branch in the loop.  Needs Cin % 64 == 0 (a K step never straddles two taps), 16-bit storage, one source.
#include "conv_tile.h"

namespace cc {

template <class T> __device__ __forceinline__ uint4 avg2x2(const uint4& a, const uint4& b, const uint4& c, const uint4& d) {
  const T* ta = reinterpret_cast<const T*>(&a); const T* tb = reinterpret_cast<const T*>(&b);
  const T* tc = reinterpret_cast<const T*>(&c); const T* td = reinterpret_cast<const T*>(&d);
  uint4 o;
  T* to = reinterpret_cast<T*>(&o);
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    float acc = 0.f;                                   // as pool_vec_kernel: the sum starts from +0 (a window of four -0 gives +0)
    acc = acc + to_f32<T>(ta[e]); acc = acc + to_f32<T>(tb[e]); acc = acc + to_f32<T>(tc[e]); acc = acc + to_f32<T>(td[e]);
    to[e] = from_f32<T>(acc * 0.25f);
  }
  return o;
}

// explicit global-address-space load: the pointer comes out of a by-value struct, where hipcc may otherwise fall back to flat_load
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 ldg16(const char* s) {
  const u32x4 v = *reinterpret_cast<const __attribute__((address_space(1))) u32x4*>((unsigned long)s);
  return make_uint4(v.x, v.y, v.z, v.w);
}

template <class T, int BN>
__global__ __launch_bounds__(256, 2) void conv_avg_s2_kernel(const ConvP p, const ConvAux a) {
  constexpr int BM = 128, NT = 256, WM = 2, WN = 2, MI = BM / WM / 16, NJ = BN / WN / 16;
  constexpr int E = 8, BK = 64, RPP = 32, XR = BM / RPP, WR = BN / RPP, STAGE = (BM + BN) * 8;
  static_assert(sizeof(T) == 2 && BN % 32 == 0 && 2 * STAGE * 16 >= BM * BN * 2, "16-bit storage; the epilogue tile must fit in the stages");
  extern __shared__ __attribute__((aligned(16))) uint4 lds[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int M = p.B * p.Ho * p.Wo, hw = p.Ho * p.Wo;
  const unsigned lds_base = lds_addr(lds);
  int wg;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int mt_ = wg / a.nt;
  const int m0 = mt_ * BM, n0 = (wg - mt_ * a.nt) * BN;

  // this thread stages LDS position `ppos` of rows prow + 32 i from source chunk `chunk` (the chunk swizzle, applied at the source)
  const int ppos = tid & 7, prow = tid >> 3;
  const int chunk = ppos ^ swz<8>(prow);
  const char* rowp[XR]; unsigned vmask[XR];              // window (2ho-1, 2wo-1) of the pixel's first tap, channel coff + chunk*8; taps inside the averaged map
#pragma unroll
  for (int i = 0; i < XR; ++i) {
    const int m = m0 + prow + RPP * i, mm = m < M ? m : 0;
    const int b = fdiv(mm, hw, a.inv_hw), rem = mm - b * hw, ho = fdiv(rem, p.Wo, a.inv_wo), wo = rem - ho * p.Wo;
    const int h0 = ho * 2 - 1, w0 = wo * 2 - 1;
    unsigned hm = 0, wm = 0;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      hm |= (unsigned)((unsigned)(h0 + r) < (unsigned)p.Hin) << r;       // Hin x Win = the averaged map: (H-1) x (W-1)
      wm |= (unsigned)((unsigned)(w0 + r) < (unsigned)p.Win) << r;
    }
    const unsigned vm = ((hm & 1u) ? wm : 0u) | ((hm & 2u) ? wm << 3 : 0u) | ((hm & 4u) ? wm << 6 : 0u);
    vmask[i] = m < M ? vm : 0u;
    rowp[i] = reinterpret_cast<const char*>(p.s0.ptr) +
              ((((long)b * p.s0.H + h0) * p.s0.W + w0) * (long)p.s0.cstride + p.s0.coff + chunk * E) * (long)sizeof(T);
  }
  const char* wcur[WR]; unsigned winc[WR];
#pragma unroll
  for (int i = 0; i < WR; ++i) {
    const int n = n0 + prow + RPP * i;
    const bool ok = n < p.Cout;
    wcur[i] = ok ? reinterpret_cast<const char*>(p.w) + ((size_t)n * p.Kw + chunk * E) * sizeof(T) : reinterpret_cast<const char*>(&g_zero16);
    winc[i] = ok ? (unsigned)(BK * sizeof(T)) : 0u;
  }
  const unsigned col_b = (unsigned)p.s0.cstride * (unsigned)sizeof(T), row_b = (unsigned)p.s0.W * col_b;   // next column / next row of the source
  const int spt = p.Cin / BK;                           // K steps per tap
  const int nkt = 9 * spt;

  uint4 win[XR][4];                                     // the 2x2 windows of the step being staged
  int tap = 0, cs = 0;                                  // (tap, channel step) of the NEXT window load
  auto load_windows = [&]() {
    const long delta = ((long)((tap / 3) * p.s0.W + (tap % 3)) * p.s0.cstride + cs * BK) * (long)sizeof(T);
#pragma unroll
    for (int i = 0; i < XR; ++i) {
      const bool ok = (vmask[i] >> tap) & 1u;
      const char* s = ok ? rowp[i] + delta : reinterpret_cast<const char*>(&g_zero16);
      const unsigned cb = ok ? col_b : 0u, rb = ok ? row_b : 0u;
      win[i][0] = ldg16(s); win[i][1] = ldg16(s + cb); win[i][2] = ldg16(s + rb); win[i][3] = ldg16(s + rb + cb);
    }
    if (++cs == spt) { cs = 0; ++tap; }
  };
  auto write_windows = [&](int stage) {
    uint4* dst = lds + stage * STAGE + tid;
#pragma unroll
    for (int i = 0; i < XR; ++i) dst[i * NT] = avg2x2<T>(win[i][0], win[i][1], win[i][2], win[i][3]);
  };
  auto issue_weights = [&](int stage) {
    const unsigned sbase = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(stage * STAGE + BM * 8 + wave * 64) * 16u);
#pragma unroll
    for (int i = 0; i < WR; ++i) { glds16(wcur[i], sbase + i * (NT * 16u)); wcur[i] += winc[i]; }
  };

  const int wm0 = (wave % WM) * (BM / WM), wn0 = (wave / WM) * (BN / WN);
  const int fr = lane & 15, fg = lane >> 4;
  f32x4 acc[NJ][MI];
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int i = 0; i < MI; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};

  // prologue: stage 0 <- step 0 (weights by DMA, windows through registers); the windows of step 1 in flight
  issue_weights(0);
  load_windows();
  write_windows(0);
  if (nkt > 1) load_windows();
  for (int kt = 0; kt < nkt; ++kt) {
    // the weight DMA of step kt was issued BEFORE the window loads of step kt+1: at most those 4*XR loads may stay outstanding
    if (kt + 1 < nkt) wait_vmcnt<4 * XR>(); else wait_vmcnt<0>();
    __syncthreads();                                   // stage kt complete (ds_writes of every wave drained by the barrier's lgkmcnt(0)); stage kt^1 free
    const int st = kt & 1;
    if (kt + 1 < nkt) issue_weights(st ^ 1);
    const uint4* ldsX = lds + st * STAGE;
    const uint4* ldsW = ldsX + BM * 8;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      uint4 xf[MI], wf[NJ];
#pragma unroll
      for (int i = 0; i < MI; ++i) { const int row = wm0 + i * 16 + fr; xf[i] = ldsX[row * 8 + ((h * 4 + fg) ^ swz<8>(row))]; }
#pragma unroll
      for (int j = 0; j < NJ; ++j) { const int row = wn0 + j * 16 + fr; wf[j] = ldsW[row * 8 + ((h * 4 + fg) ^ swz<8>(row))]; }
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int i = 0; i < MI; ++i) Mma<T>::run(wf[j], xf[i], acc[j][i]);
    }
    if (kt + 1 < nkt) {
      write_windows(st ^ 1);                           // (the compiler waits for the window loads here; the weight DMA issued above is younger)
      if (kt + 2 < nkt) load_windows();
    }
  }
  conv_epilogue<T, BM, BN, WM, MI, NJ>(p, acc, n0, lds, [&](int row) { const int m = m0 + row; return m < M ? (long)m : -1L; });
}

bool conv_adown_supported(int dt, const ConvP& p) {
  return dt != F32 && !p.split && p.s0.shift == -1 && p.s1.C == 0 && p.ks == 3 && p.stride == 2 && p.pad == 1 && p.Cin % 64 == 0 && p.Cin == p.s0.C &&
         p.Hin == p.s0.H - 1 && p.Win == p.s0.W - 1 && p.s0.coff % 8 == 0 && p.s0.cstride % 8 == 0 && p.Kw == 9 * p.Cin && !p.res;
}

template <class T, int BN> static void launch_adown_k(const ConvP& p, hipStream_t stream) {
  constexpr size_t lds = (size_t)2 * (128 + BN) * 8 * 16;
  static PerDevice once;                               // the attribute is per device (common.h)
  if (once.first(once.index())) {
    CC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_avg_s2_kernel<T, BN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  }
  const int M = p.B * p.Ho * p.Wo;
  ConvAux a{};
  a.nt = (p.Cout + BN - 1) / BN;
  a.inv_hw = 1.0f / (float)(p.Ho * p.Wo); a.inv_wo = 1.0f / (float)p.Wo;
  hipLaunchKernelGGL((conv_avg_s2_kernel<T, BN>), dim3((unsigned)(((M + 127) / 128) * a.nt)), dim3(256), lds, stream, p, a);
  CC_HIP(hipGetLastError());
}

void launch_conv_adown(int dt, const ConvP& p, hipStream_t stream) {
  CC_CHECK(conv_adown_supported(dt, p), "fused average + stride-2 conv: unsupported shape");
  auto padded = [&](int bn) { return (p.Cout + bn - 1) / bn * bn; };
  const bool narrow = padded(64) < padded(128);
  if (dt == F16) { if (narrow) launch_adown_k<f16_t, 64>(p, stream); else launch_adown_k<f16_t, 128>(p, stream); }
  else { if (narrow) launch_adown_k<bf16_t, 64>(p, stream); else launch_adown_k<bf16_t, 128>(p, stream); }
}

}  // namespace cc

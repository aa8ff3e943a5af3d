// 3x3 stride-1 convolution for the narrow layers (Cin, Cout in {32, 64}): weights stationary in LDS, one AUTONOMOUS wave per
// output sub-tile - no barrier after the weights have landed.
//
// These are the RepNBottleneck convs of the RepNCSP blocks (detection/yolov9.py:82-90; 32 channels at 160x160, 64 at 80x80
// in YOLOv9-C).  K = 9*Cin is at most 576, i.e. nine tiny K steps: in a cooperative tile kernel every step is a barrier and
// every tile ends in a block-wide epilogue during which the matrix pipes idle (conv3x3_ws_kernel measured ~31 % MFMA
// utilisation, one block of four waves per CU because the weights take half the LDS).  Here the block still holds ONE copy
// of the weights (<= 72 KB), but each wave runs its own pipeline on a 2 x 16-pixel sub-tile:
//     DMA the 4 x 18 input patch into the wave's private LDS buffer  ->  s_waitcnt vmcnt(0)  (only this wave waits)
//     -> 9 taps x Cin/32 MFMA steps straight from LDS (2 pixel fragments x Cout/16 weight fragments)
//     -> issue the NEXT sub-tile's patch DMA  ->  bias + activation (+ residual) and 8-byte stores from registers.
// Eight waves (64 channels) or sixteen (32 channels) share a CU, two or four per SIMD: while one wave waits for its patch or
// runs its epilogue VALU the others keep the matrix pipe busy, with no s_barrier anywhere in the loop.  A wave orders its own
// LDS-DMA writes against its own ds_reads with its vmcnt (MI355X_MICROARCH.md: nothing else is needed for same-wave data).
#include "conv_tile.h"

namespace cc {

struct WaveAux { float inv_tiles, inv_tx; int tiles, tx, total; };

template <class T, int CIN, int COUT, int NW>
__global__ __launch_bounds__(NW * 64) void conv3x3_wave_kernel(const ConvP p, const WaveAux a) {
  constexpr int E = 8, CPRW = CIN / E, ROWB = CIN * 2;          // 16-byte chunks / bytes per pixel row
  constexpr int MI = 2, NJ = COUT / 16, HS = CPRW / 4;           // pixel fragments (2 rows x 16 px), channel fragments, 32-wide k steps per tap
  constexpr int PW = 18, RPI = 64 / CPRW, NPI = (4 * PW + RPI - 1) / RPI, PROWS = NPI * RPI;   // patch: 4 x 18 pixels in NPI DMA pieces
  constexpr int NT = NW * 64, RPP = NT / CPRW, WROWS = 9 * COUT, WPASS = (WROWS + RPP - 1) / RPP;
  constexpr int WLB = WPASS * RPP * ROWB;                         // bytes of resident weights
  constexpr int PB = PROWS * ROWB;                                // bytes of one wave's patch buffer
  static_assert(sizeof(T) == 2 && (CIN == 32 || CIN == 64) && (COUT == 32 || COUT == 64), "narrow 16-bit layers only");
  static_assert(WLB + NW * PB <= 160 * 1024, "LDS budget");
  extern __shared__ __attribute__((aligned(16))) uint4 lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned lds_base = lds_addr(lds);

  // ---- resident weights: LDS row R = tap*COUT + n holds w[n][tap*CIN .. +CIN), chunk-swizzled -----------------------------
  {
    const int ppos = tid % CPRW, prow = tid / CPRW;
#pragma unroll
    for (int i = 0; i < WPASS; ++i) {
      const int R = prow + RPP * i;
      const int tap = R / COUT, n = R - tap * COUT;
      const int chunk = ppos ^ swz<CPRW>(R);
      const void* src = (R < WROWS && n < p.Cout) ? static_cast<const void*>(reinterpret_cast<const char*>(p.w) +
                            ((size_t)n * p.Kw + tap * CIN + chunk * E) * sizeof(T)) : static_cast<const void*>(&g_zero16);
      glds16(src, __builtin_amdgcn_readfirstlane(lds_base + (unsigned)((RPP * i) * CPRW + wave * 64) * 16u));
    }
  }

  // ---- per-wave constants ----------------------------------------------------------------------------------------------------
  const int fr = lane & 15, fg = lane >> 4;
  const unsigned patch_base = lds_base + (unsigned)(WLB + wave * PB);
  const char* patch = reinterpret_cast<const char*>(lds) + WLB + wave * PB;
  // pixel-fragment byte offsets inside the patch for patch row q = i + r (0..3), column shift s (0..2), k step h
  unsigned poff[4][3][HS];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
      for (int h = 0; h < HS; ++h) {
        const int row = q * PW + s + fr;
        poff[q][s][h] = (unsigned)(row * ROWB + (((h * 4 + fg) ^ swz<CPRW>(row)) & (CPRW - 1)) * 16);
      }
  // weight fragments: row = tap*COUT + j*16 + fr with a 16-aligned base, so the swizzle depends on the lane only
  const char* wlo[HS]; const char* whi[HS];                       // taps 0..3 / taps 4..8 (ds_read offset field is 16 bits)
#pragma unroll
  for (int h = 0; h < HS; ++h) {
    wlo[h] = reinterpret_cast<const char*>(lds) + fr * ROWB + (((h * 4 + fg) ^ swz<CPRW>(fr)) & (CPRW - 1)) * 16;
    whi[h] = wlo[h] + 4 * COUT * ROWB;
  }
  // patch DMA lane roles: piece i covers patch rows i*RPI + lane / CPRW
  const int dpos = lane % CPRW, drow = lane / CPRW;

  auto issue_patch = [&](int st) {
    const int b = fdiv(st, a.tiles, a.inv_tiles), trem = st - b * a.tiles, ty = fdiv(trem, a.tx, a.inv_tx), tx = trem - ty * a.tx;
    const int h0 = ty * 2 - 1, w0 = tx * 16 - 1;
#pragma unroll
    for (int i = 0; i < NPI; ++i) {
      const int pr = i * RPI + drow;
      const int py = pr / PW, px = pr - py * PW;
      const int ih = h0 + py, iw = w0 + px;
      const bool ok = pr < 4 * PW && (unsigned)ih < (unsigned)p.Hin && (unsigned)iw < (unsigned)p.Win;
      const int chunk = dpos ^ swz<CPRW>(pr);
      const void* src = ok ? static_cast<const void*>(reinterpret_cast<const char*>(p.s0.ptr) +
                                 ((((long)b * p.s0.H + ih) * p.s0.W + iw) * (long)p.s0.cstride + p.s0.coff + chunk * E) * (long)sizeof(T))
                           : static_cast<const void*>(&g_zero16);
      glds16(src, patch_base + (unsigned)(i * 1024));
    }
  };

  const int step = gridDim.x * NW;
  int st = blockIdx.x * NW + wave;
  if (st < a.total) issue_patch(st);
  wait_vmcnt<0>();
  __syncthreads();                                                // the weights have landed for everybody: the only barrier

  for (; st < a.total; st += step) {
    wait_vmcnt<0>();                                              // this wave's patch (and its previous stores) are done
    f32x4 acc[NJ][MI];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int i = 0; i < MI; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int r = tap / 3, s = tap % 3;
#pragma unroll
      for (int h = 0; h < HS; ++h) {
        uint4 xf[MI], wf[NJ];
#pragma unroll
        for (int i = 0; i < MI; ++i) xf[i] = *reinterpret_cast<const uint4*>(patch + poff[i + r][s][h]);
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          wf[j] = *reinterpret_cast<const uint4*>((tap < 4 ? wlo[h] : whi[h]) + ((tap < 4 ? tap : tap - 4) * COUT + j * 16) * ROWB);
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int i = 0; i < MI; ++i) Mma<T>::run(wf[j], xf[i], acc[j][i]);
      }
    }
    // every LDS read of this patch has been consumed by an MFMA that was issued: the buffer can take the next patch while the
    // epilogue runs (same-wave ordering: DMA issue follows the ds_reads in program order, and they have returned)
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int nst = st + step;
    if (nst < a.total) issue_patch(nst);
    __builtin_amdgcn_sched_barrier(0);

    const int b = fdiv(st, a.tiles, a.inv_tiles), trem = st - b * a.tiles, ty = fdiv(trem, a.tx, a.inv_tx), tx = trem - ty * a.tx;
    long mrow[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int ho = ty * 2 + i, wo = tx * 16 + fr;
      mrow[i] = (ho < p.Ho && wo < p.Wo) ? ((long)b * p.Ho + ho) * p.Wo + wo : -1L;
    }
    if (p.act == 1) direct_tile<T, 1, MI, NJ>(p, acc, mrow, fg * 4);
    else if (p.act == 2) direct_tile<T, 2, MI, NJ>(p, acc, mrow, fg * 4);
    else if (p.act == 3) direct_tile<T, 3, MI, NJ>(p, acc, mrow, fg * 4);
    else if (p.act == 4) direct_tile<T, 4, MI, NJ>(p, acc, mrow, fg * 4);
    else direct_tile<T, 0, MI, NJ>(p, acc, mrow, fg * 4);
  }
}

template <class T, int CIN, int COUT, int NW> static void launch_wave(const ConvP& p, hipStream_t stream) {
  constexpr int CPRW = CIN / 8, ROWB = CIN * 2, RPI = 64 / CPRW, NPI = (72 + RPI - 1) / RPI;
  constexpr int RPP = NW * 64 / CPRW, WPASS = (9 * COUT + RPP - 1) / RPP;
  constexpr size_t lds = (size_t)WPASS * RPP * ROWB + (size_t)NW * NPI * RPI * ROWB;
  static PerDevice pd;                                 // attribute and CU count per device ordinal (common.h)
  const int pdi = pd.index();
  if (pd.first(pdi))
    CC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_wave_kernel<T, CIN, COUT, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int cus = pd.cu_count(pdi);
  WaveAux a{};
  a.tx = (p.Wo + 15) / 16; a.tiles = ((p.Ho + 1) / 2) * a.tx; a.total = p.B * a.tiles;
  a.inv_tiles = 1.0f / (float)a.tiles; a.inv_tx = 1.0f / (float)a.tx;
  const int blocks = std::min((a.total + NW - 1) / NW, cus);
  note_launch("conv3x3_wave", conv3x3_wave_kernel<T, CIN, COUT, NW>, (long)(a.total + NW - 1) / NW, NW * 64, lds, blocks);
  hipLaunchKernelGGL((conv3x3_wave_kernel<T, CIN, COUT, NW>), dim3(blocks), dim3(NW * 64), lds, stream, p, a);
}

bool conv_wave_legal(const ConvP& p) {
  return !p.split && p.ks == 3 && p.stride == 1 && p.pad == 1 && p.s1.C == 0 && p.s0.shift == 0 && (p.Cin == 32 || p.Cin == 64) &&
         (p.Cout == 32 || p.Cout == 64) && p.Hin == p.Ho && p.Win == p.Wo && (long)p.B * ((p.Ho + 1) / 2) * ((p.Wo + 15) / 16) < (1L << 22);
}

template <class T> static void launch_wave_t(const ConvP& p, hipStream_t stream) {
  if (p.Cin == 64) { if (p.Cout == 64) launch_wave<T, 64, 64, 8>(p, stream); else launch_wave<T, 64, 32, 8>(p, stream); }
  else { if (p.Cout == 64) launch_wave<T, 32, 64, 16>(p, stream); else launch_wave<T, 32, 32, 16>(p, stream); }
}

void launch_conv_wave(int dt, const ConvP& p, hipStream_t stream) {
  CC_CHECK(conv_wave_legal(p) && dt != F32, "wave-autonomous 3x3: shape not eligible");
  if (dt == F16) launch_wave_t<f16_t>(p, stream); else launch_wave_t<bf16_t>(p, stream);
  CC_HIP(hipGetLastError());
}

}  // namespace cc

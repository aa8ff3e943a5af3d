// NHWC implicit-GEMM convolution / linear layer on MFMA (gfx950), fused bias + SiLU/GELU + residual.
//
// Replaces what tinygrad codegen emits for `nn.Conv2d(...)(x).silu()` (detection/yolov9.py:33-38) and
// `x @ W.T + b` (models/objects.py:109,120,124-126).  One kernel covers 1x1 (a plain GEMM), 3x3 s1/s2,
// two-source channel concat and nearest-x2 upsample-on-load; im2col is never materialised.
//
// GEMM view: D[n][m] = sum_k W[n][k] * X[m][k];  m = output pixel, n = output channel, k = (tap, channel).
// Weights are the MFMA A operand (rows n), activations the B operand (cols m), so each lane ends up
// with 4 consecutive output channels of one pixel.
//
// Tile: 128 pixels x BN channels x 128 bytes of K (64 halfs / 32 floats) per step, 256 threads = 4 waves.
// * Staging: global_load_lds (16 B per lane, straight into LDS, no VGPR round trip), two LDS stages, one
//   barrier per K step: step t+1 streams in while the MFMAs of step t run.  LDS rows are 128 B; the eight
//   16-B chunks of a row are XOR-swizzled by (row>>1)&7.  The DMA writes lane-linear (wave base + lane*16),
//   so the swizzle is applied to the per-lane SOURCE chunk and again on the fragment read; DMA writes
//   (8 lanes = one row) and ds_read_b128 (16 rows x one chunk column) are both conflict-free.
// * Loader state is one 64-bit pointer per staged row that simply advances by the K step:  cur += inc.
//   Halo / out-of-range taps point at a 16-byte zero page with inc = 0.  Pointers are re-targeted only when
//   the thread's chunk crosses into the next filter tap (every Cin/64 steps), from a per-row base pointer and a
//   9-bit tap-validity mask computed once per tile.  (rocprofv3 PMC showed the first version VALU-bound:
//   ~170 VALU per K step per wave against 32 MFMA; this form needs ~2 VALU per DMA.)
// * Weight rows are zero padded to a whole number of K steps at load time, so the K tail needs no checks.
// * Epilogue: bias -> activation (compile-time branch) in registers, v_cvt_pk to bf16/f16, then (16-bit
//   outputs) through a consumed LDS stage (chunk-swizzled) so that every pixel's BN channels leave as
//   16-byte-per-lane, line-contiguous stores; f32 outputs / residual adds store directly from registers.
// * Tiles are issued in an XCD-aware order (bijective remap of blockIdx): all channel tiles of a pixel tile and
//   neighbouring pixel tiles run on one XCD's L2.
// * f32 mode uses v_mfma_f32_16x16x4_f32 (exact f32) with the k-slots of a 16-B chunk spread over 4 MFMAs.
//
// Kernels in this file, chosen per layer by launch_t():
//   conv_mfma_kernel     the generic one described above (every shape, every dtype)
//   conv_big_kernel      256x256 tile / 4 waves / one barrier per K step, for deep GEMM-shaped 16-bit layers
//   conv3x3_halo_kernel  3x3 s1 with the input patch resident in LDS across the nine taps (128/256-channel layers)
//   conv3x3_ws_kernel    3x3 s1 for narrow layers: persistent blocks, all nine taps of the weights resident in LDS
#include "conv_tile.h"

namespace cc {

// SIMPLE: one source, no upsample, ks <= 3 -> incremental row pointers.  !SIMPLE: general two-source / upsample path.
// NS = LDS stages: the DMA of step t+NS-1 is issued while step t computes (prefetch distance NS-1 steps).
// BM = pixels per tile (128 -> 4 waves, 256 -> 8 waves: half the L2->LDS weight traffic per flop).
// (A ping-pong schedule and a register-staged loader were measured and dropped: DESIGN.md §4.)
// NTH = threads per block: 2 * BM by default; the few-tile configuration runs 64-pixel tiles on 256 threads (half the DMA issues per wave and step).
template <class T, int BM, int BN, int WM, bool SIMPLE, int CPRW, int NS, int NTH = 2 * BM>
__global__ __launch_bounds__(NTH) void conv_mfma_kernel(const ConvP p, const ConvAux a) {
  constexpr int NT = NTH, WN = NT / 64 / WM;
  constexpr int MI = BM / WM / 16, NJ = BN / WN / 16;
  constexpr int E = 16 / (int)sizeof(T);   // elements per 16-byte chunk
  constexpr int BK = CPRW * E;             // elements per K step
  constexpr int RPP = NT / CPRW;           // rows staged per pass of the NT threads
  constexpr int XR = BM / RPP;             // pixel rows staged per thread
  constexpr int WR = BN / RPP;             // weight rows staged per thread
  constexpr int LPS = XR + WR;             // DMA instructions per thread per K step
  constexpr int STAGE = (BM + BN) * CPRW;  // uint4 per stage
  static_assert(BN % RPP == 0, "BN too small for this row width");
  static_assert(NS * STAGE * 16 >= BM * BN * 2, "epilogue tile must fit in the stages");
  static_assert(MI >= 1 && NJ >= 1, "bad wave layout");
  static_assert((NS - 2) * LPS < 64, "vmcnt is a 6-bit counter");
  extern __shared__ __attribute__((aligned(16))) uint4 lds[];   // NS stages; the first BM*BN*2 bytes double as the epilogue tile

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

  // ---- loader setup: LDS position `ppos` of rows prow + 32*i  <-  global chunk `chunk`
  const int ppos = tid % CPRW, prow = tid / CPRW;
  const int chunk = ppos ^ swz<CPRW>(prow);            // swz(prow + RPP*i) is the same for every i
  const char* rowp[XR]; unsigned vmask[XR];            // SIMPLE: byte address of (pixel (h0,w0), channel coff); tap-in-range bits
  const char* cur[XR]; unsigned inc[XR];               // SIMPLE: running source pointer and its per-step increment
  int pb[XR], ph0[XR], pw0[XR];                        // general: batch index, top-left input coordinate
#pragma unroll
  for (int i = 0; i < XR; ++i) {
    const int m = m0 + prow + RPP * i;
    if constexpr (SIMPLE) {
      if (a.is1x1) {
        rowp[i] = reinterpret_cast<const char*>(p.s0.ptr) + ((size_t)m * p.s0.cstride + p.s0.coff) * sizeof(T);
        vmask[i] = m < M ? 1u : 0u;
      } else {
        const int mm = m < M ? m : 0;
        const int b = fdiv(mm, hw, a.inv_hw), rem = mm - b * hw, ho = fdiv(rem, p.Wo, a.inv_wo), wo = rem - ho * p.Wo;
        const int h0 = ho * p.stride - p.pad, w0 = wo * p.stride - p.pad;
        unsigned hm = 0, wm = 0;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          hm |= (unsigned)(r < p.ks && (unsigned)(h0 + r) < (unsigned)p.Hin) << r;
          wm |= (unsigned)(r < p.ks && (unsigned)(w0 + r) < (unsigned)p.Win) << r;
        }
        const unsigned vm = ((hm & 1u) ? wm : 0u) | ((hm & 2u) ? wm << p.ks : 0u) | ((hm & 4u) ? wm << (2 * p.ks) : 0u);
        vmask[i] = m < M ? vm : 0u;
        rowp[i] = reinterpret_cast<const char*>(p.s0.ptr) +
                  ((((long)b * p.s0.H + h0) * p.s0.W + w0) * (long)p.s0.cstride + p.s0.coff) * (long)sizeof(T);
      }
    } else {
      if (m < M) {
        const int b = fdiv(m, hw, a.inv_hw), rem = m - b * hw, ho = fdiv(rem, p.Wo, a.inv_wo), wo = rem - ho * p.Wo;
        pb[i] = b; ph0[i] = ho * p.stride - p.pad; pw0[i] = wo * p.stride - p.pad;
      } else { pb[i] = 0; ph0[i] = -(1 << 28); pw0[i] = 0; }
    }
  }
  // weight rows: zero padded to whole K steps (p.Kw), rows past Cout read the zero page
  const char* wcur[WR]; unsigned winc[WR];
#pragma unroll
  for (int i = 0; i < WR; ++i) {
    const int n = n0 + prow + RPP * i;
    const bool ok = n < p.Cout;
    wcur[i] = ok ? reinterpret_cast<const char*>(p.w) + ((size_t)n * p.Kw + chunk * E) * sizeof(T) : reinterpret_cast<const char*>(&g_zero16);
    winc[i] = ok ? (unsigned)(BK * sizeof(T)) : 0u;
  }
  int k0 = chunk * E, kc, kr, ks_, vt;                 // this thread's k index; its channel, tap row, tap col; virtual tap (split weights: two per tap)
  if (k0 < p.Cin) { kc = k0; kr = 0; ks_ = 0; vt = 0; }
  else { vt = k0 / p.Cin; kc = k0 - vt * p.Cin; const int tap = vt >> p.split; kr = tap / p.ks; ks_ = tap - kr * p.ks; }

  auto retarget = [&]() {                              // SIMPLE: (kr, ks_, kc) changed tap -> new pointers
    const bool kok = kr < p.ks;                        // k0 < Ktot
    const int tbit = kr * p.ks + ks_;
    const long delta = ((long)(kr * p.s0.W + ks_) * p.s0.cstride + kc) * (long)sizeof(T);
#pragma unroll
    for (int i = 0; i < XR; ++i) {
      const bool ok = kok && ((vmask[i] >> tbit) & 1u);
      cur[i] = ok ? rowp[i] + delta : reinterpret_cast<const char*>(&g_zero16);
      inc[i] = ok ? (unsigned)(BK * sizeof(T)) : 0u;
    }
  };
  if constexpr (SIMPLE) retarget();

  auto issue_loads = [&](int stage) {
    // wave-uniform LDS byte address of this wave's 1 KiB slice of the stage (+ lane*16 added by the DMA)
    const unsigned sbase = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(stage * STAGE + wave * 64) * 16u);
    if constexpr (SIMPLE) {
#pragma unroll
      for (int i = 0; i < XR; ++i) glds16(cur[i], sbase + i * (NT * 16u));
    } else {
      const bool kok = k0 < p.Ktot;
      const bool first = kc < p.s0.C;
      const T* sp = reinterpret_cast<const T*>(first ? p.s0.ptr : p.s1.ptr);
      const int scs = first ? p.s0.cstride : p.s1.cstride, sco = first ? p.s0.coff : p.s1.coff;
      const int cc = first ? kc : kc - p.s0.C;
      const int sH = first ? p.s0.H : p.s1.H, sW = first ? p.s0.W : p.s1.W, ssh = first ? p.s0.shift : p.s1.shift;
#pragma unroll
      for (int i = 0; i < XR; ++i) {
        const int ih = ph0[i] + kr, iw = pw0[i] + ks_;
        const bool ok = kok && (unsigned)ih < (unsigned)p.Hin && (unsigned)iw < (unsigned)p.Win;
        const size_t off = ((size_t)(pb[i] * sH + (ih >> ssh)) * sW + (iw >> ssh)) * scs + sco + cc;
        glds16(ok ? static_cast<const void*>(sp + off) : static_cast<const void*>(&g_zero16), sbase + i * (NT * 16u));
      }
    }
#pragma unroll
    for (int i = 0; i < WR; ++i) glds16(wcur[i], sbase + (BM * CPRW + i * NT) * 16u);
  };
  auto advance_k = [&]() {
    k0 += BK; kc += BK;
#pragma unroll
    for (int i = 0; i < WR; ++i) wcur[i] += winc[i];
    if (kc >= p.Cin) {
      do { kc -= p.Cin; ++vt; if (next_filter_tap(vt, p.split)) { if (++ks_ == p.ks) { ks_ = 0; ++kr; } } } while (kc >= p.Cin);
      if constexpr (SIMPLE) retarget();
    } else if constexpr (SIMPLE) {
#pragma unroll
      for (int i = 0; i < XR; ++i) cur[i] += inc[i];
    }
  };

  const int wm0 = (wave % WM) * (BM / WM), wn0 = (wave / WM) * (BN / WN);
  const int fr = lane & 15, fg = lane >> 4;
  const int nkt = (p.Ktot + BK - 1) / BK;

  f32x4 acc[NJ][MI];
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int i = 0; i < MI; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};

  // prologue: steps 0 .. NS-2 in flight
  int issued = 0;
#pragma unroll
  for (int s = 0; s < NS - 1; ++s)
    if (s < nkt) { if (s > 0) advance_k(); issue_loads(s); ++issued; }
  int st_c = 0, st_i = NS - 1;                          // stage being consumed / stage the next DMA goes to
  for (int kt = 0; kt < nkt; ++kt) {
    // step kt must have landed; later steps (at most NS-2 of them) may stay in flight
    const int ahead = issued - kt - 1;
    if (NS > 5 && ahead >= 4) wait_vmcnt<4 * LPS>();
    else if (NS > 4 && ahead >= 3) wait_vmcnt<3 * LPS>();
    else if (NS > 3 && ahead >= 2) wait_vmcnt<2 * LPS>();
    else if (NS > 2 && ahead >= 1) wait_vmcnt<LPS>();
    else wait_vmcnt<0>();
    __syncthreads();                                   // ... for every wave; the stage consumed at step kt-1 is free again
    if (issued < nkt) { advance_k(); issue_loads(st_i); ++issued; if (++st_i == NS) st_i = 0; }
    const uint4* ldsX = lds + st_c * STAGE;
    const uint4* ldsW = ldsX + BM * CPRW;
    if (++st_c == NS) st_c = 0;
#pragma unroll
    for (int h = 0; h < CPRW / 4; ++h) {
      uint4 xf[MI], wf[NJ];
#pragma unroll
      for (int i = 0; i < MI; ++i) { const int row = wm0 + i * 16 + fr; xf[i] = ldsX[row * CPRW + ((h * 4 + fg) ^ swz<CPRW>(row))]; }
#pragma unroll
      for (int j = 0; j < NJ; ++j) { const int row = wn0 + j * 16 + fr; wf[j] = ldsW[row * CPRW + ((h * 4 + fg) ^ swz<CPRW>(row))]; }
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int i = 0; i < MI; ++i) Mma<T>::run(wf[j], xf[i], acc[j][i]);
    }
  }

  // ---- epilogue ------------------------------------------------------------------------------------
  conv_epilogue<T, BM, BN, WM, MI, NJ, NT>(p, acc, n0, lds, [&](int row) { const int m = m0 + row; return m < M ? (long)m : -1L; });
}

// ---- 256 x 256 tile, four waves of 128 x 128: the schedule for wide, deep GEMM-shaped layers ---------------------------------
// Why a second main loop: in the 128x128 kernel every MFMA costs half a ds_read_b128 and every K step two barriers' worth of
// skew; the ablation puts its ceiling near 0.9 PF while the library reaches 1.0-1.2 PF with 256x256 macro tiles.
// Here one block per CU, one wave per SIMD; a wave owns 128 pixels x 128 channels (64 accumulator fragments = 256 registers,
// AGPRs), so a 32-wide k-substep is 64 MFMAs fed by 16 fragment reads (0.25 per MFMA).  Fragments are double-buffered in
// registers and the loop has ONE barrier per K step, placed in the middle of the second substep:
//     64 MFMA on F0(t)         | 16 ds_read -> F1(t)
//     32 MFMA on F1(t)
//     s_waitcnt vmcnt(0) ; barrier          stage t+1 has landed for everyone, nobody reads stage t any more
//     32 MFMA on F1(t)         | 16 DMA issues for stage t+2 (into the buffer of stage t) | 16 ds_read -> F0(t+1)
// so the LDS-DMA of a stage is issued a whole step before it is needed and the matrix pipe never waits for a fragment.
// sched_barrier pins the interleave (the scheduler otherwise hoists all reads to the top and spills).
// TS = tile side: 256 (one block per CU, accumulators in AGPRs) or 128 (64 x 64 per wave, two blocks per CU; same schedule).
template <class T, int TS>
__global__ __launch_bounds__(256, (TS == 256 ? 1 : 2)) void conv_big_kernel(const ConvP p, const ConvAux a) {
  constexpr int BM = TS, BN = TS, NT = 256, WM = 2, WN = 2, MI = BM / WM / 16, NJ = BN / WN / 16;
  constexpr int NF = MI + NJ, MF = MI * NJ, MPG = MF / NF;   // fragments and MFMAs per 32-wide k-substep, MFMAs per fragment read
  constexpr int E = 8, CPRW = 8, BK = 64, RPP = NT / CPRW, XR = BM / RPP, WR = BN / RPP;
  constexpr int STAGE = (BM + BN) * CPRW;              // uint4 per stage; two stages hold the epilogue tile
  static_assert(XR + WR == NF && MF % NF == 0 && NF % 2 == 0, "one DMA piece and MF/NF MFMAs per fragment read");
  static_assert(sizeof(T) == 2, "16-bit storage only");
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

  // ---- loader state (same scheme as conv_mfma_kernel's SIMPLE path) ------------------------------------------------------
  const int ppos = tid % CPRW, prow = tid / CPRW;
  const int chunk = ppos ^ swz<CPRW>(prow);
  const char* rowp[XR]; unsigned vmask[XR];
  const char* cur[XR]; unsigned inc[XR];
#pragma unroll
  for (int i = 0; i < XR; ++i) {
    const int m = m0 + prow + RPP * i;
    if (a.is1x1) {
      rowp[i] = reinterpret_cast<const char*>(p.s0.ptr) + ((size_t)m * p.s0.cstride + p.s0.coff) * sizeof(T);
      vmask[i] = m < M ? 1u : 0u;
    } else {
      const int mm = m < M ? m : 0;
      const int b = fdiv(mm, hw, a.inv_hw), rem = mm - b * hw, ho = fdiv(rem, p.Wo, a.inv_wo), wo = rem - ho * p.Wo;
      const int h0 = ho * p.stride - p.pad, w0 = wo * p.stride - p.pad;
      unsigned hm = 0, wmk = 0;
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        hm |= (unsigned)(r < p.ks && (unsigned)(h0 + r) < (unsigned)p.Hin) << r;
        wmk |= (unsigned)(r < p.ks && (unsigned)(w0 + r) < (unsigned)p.Win) << r;
      }
      const unsigned vm = ((hm & 1u) ? wmk : 0u) | ((hm & 2u) ? wmk << p.ks : 0u) | ((hm & 4u) ? wmk << (2 * p.ks) : 0u);
      vmask[i] = m < M ? vm : 0u;
      rowp[i] = reinterpret_cast<const char*>(p.s0.ptr) +
                ((((long)b * p.s0.H + h0) * p.s0.W + w0) * (long)p.s0.cstride + p.s0.coff) * (long)sizeof(T);
    }
  }
  const char* wcur[WR]; unsigned winc[WR];
#pragma unroll
  for (int i = 0; i < WR; ++i) {
    const int n = n0 + prow + RPP * i;
    const bool ok = n < p.Cout;
    wcur[i] = ok ? reinterpret_cast<const char*>(p.w) + ((size_t)n * p.Kw + chunk * E) * sizeof(T) : reinterpret_cast<const char*>(&g_zero16);
    winc[i] = ok ? (unsigned)(BK * sizeof(T)) : 0u;
  }
  int k0 = chunk * E, kc, kr, ks_, vt;
  if (k0 < p.Cin) { kc = k0; kr = 0; ks_ = 0; vt = 0; }
  else { vt = k0 / p.Cin; kc = k0 - vt * p.Cin; const int tap = vt >> p.split; kr = tap / p.ks; ks_ = tap - kr * p.ks; }
  auto retarget = [&]() {
    const bool kok = kr < p.ks;
    const int tbit = kr * p.ks + ks_;
    const long delta = ((long)(kr * p.s0.W + ks_) * p.s0.cstride + kc) * (long)sizeof(T);
#pragma unroll
    for (int i = 0; i < XR; ++i) {
      const bool ok = kok && ((vmask[i] >> tbit) & 1u);
      cur[i] = ok ? rowp[i] + delta : reinterpret_cast<const char*>(&g_zero16);
      inc[i] = ok ? (unsigned)(BK * sizeof(T)) : 0u;
    }
  };
  retarget();
  auto advance_k = [&]() {
    k0 += BK; kc += BK;
#pragma unroll
    for (int i = 0; i < WR; ++i) wcur[i] += winc[i];
    if (kc >= p.Cin) {
      do { kc -= p.Cin; ++vt; if (next_filter_tap(vt, p.split)) { if (++ks_ == p.ks) { ks_ = 0; ++kr; } } } while (kc >= p.Cin);
      retarget();
    } else {
#pragma unroll
      for (int i = 0; i < XR; ++i) cur[i] += inc[i];
    }
  };
  // DMA piece q of a stage: q < XR -> pixel rows prow + 32q, else weight rows prow + 32(q - XR)
  auto issue_piece = [&](int stage, int q) {
    const unsigned sbase = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(stage * STAGE + wave * 64) * 16u);
    if (q < XR) glds16_m0(cur[q], sbase + q * (NT * 16u));
    else glds16_m0(wcur[q - XR], sbase + (BM * CPRW + (q - XR) * NT) * 16u);
  };

  const int wm0 = (wave % WM) * (BM / WM), wn0 = (wave / WM) * (BN / WN);
  const int fr = lane & 15, fg = lane >> 4;
  const int nkt = (p.Ktot + BK - 1) / BK;
  f32x4 acc[NJ][MI];
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int i = 0; i < MI; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment index q < 8 -> pixel fragment q, else weight fragment q - 8, of k-substep h in `stage`
  uint4 f0[NF], f1[NF];
  auto frag = [&](int stage, int h, int q) -> uint4 {
    const int row = (q < MI ? wm0 + q * 16 : BM + wn0 + (q - MI) * 16) + fr;   // row in the stage: pixels first, then weights
    const int srow = q < MI ? row : row - BM;                                    // swizzle uses the row inside its own slab
    return lds[stage * STAGE + row * CPRW + ((h * 4 + fg) ^ swz<CPRW>(srow))];
  };

  // prologue: stages 0 and 1 in flight, stage 0 landed, F0(0) in registers
#pragma unroll
  for (int q = 0; q < XR + WR; ++q) issue_piece(0, q);
  if (nkt > 1) {
    advance_k();
#pragma unroll
    for (int q = 0; q < XR + WR; ++q) issue_piece(1, q);
    wait_vmcnt<XR + WR>();                               // only loads in flight here: they complete in order
  } else {
    wait_vmcnt<0>();
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < NF; ++q) f0[q] = frag(0, 0, q);

  // one K step; MORE = a next step exists (barrier + its first fragments), MORE2 = a step after that exists (its DMA).
  // The last two steps are peeled so that the steady-state body has no branches between MFMA groups.
  auto step = [&](int kt, auto more_tag, auto more2_tag) {
    constexpr bool MORE = decltype(more_tag)::value, MORE2 = decltype(more2_tag)::value;
    const int st = kt & 1;
    // substep 0: the MFMAs on F0, the reads of F1 spread underneath (one read per MPG MFMAs)
#pragma unroll
    for (int g = 0; g < NF; ++g) {
      f1[g] = frag(st, 1, g);
#pragma unroll
      for (int u = 0; u < MPG; ++u) { const int e = g * MPG + u, j = e / MI, i = e % MI; Mma<T>::run(f0[MI + j], f0[i], acc[j][i]); }
      __builtin_amdgcn_sched_barrier(0);
    }
    // substep 1, first half
#pragma unroll
    for (int g = 0; g < NF / 2; ++g) {
#pragma unroll
      for (int u = 0; u < MPG; ++u) { const int e = g * MPG + u, j = e / MI, i = e % MI; Mma<T>::run(f1[MI + j], f1[i], acc[j][i]); }
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (MORE) {
      wait_vmcnt<0>();
      __syncthreads();                                   // stage kt+1 landed for everyone; stage kt is dead
      if constexpr (MORE2) advance_k();
    }
    // substep 1, second half: DMA of stage kt+2 into the dead buffer and the reads of F0(kt+1) ride under the MFMAs
#pragma unroll
    for (int g = NF / 2; g < NF; ++g) {
      const int q = 2 * (g - NF / 2);
      if constexpr (MORE2) { issue_piece(st, q); issue_piece(st, q + 1); }
      if constexpr (MORE) { f0[q] = frag(st ^ 1, 0, q); f0[q + 1] = frag(st ^ 1, 0, q + 1); }
#pragma unroll
      for (int u = 0; u < MPG; ++u) { const int e = g * MPG + u, j = e / MI, i = e % MI; Mma<T>::run(f1[MI + j], f1[i], acc[j][i]); }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  int kt = 0;
  for (; kt + 2 < nkt; ++kt) step(kt, std::true_type{}, std::true_type{});
  if (kt + 1 < nkt) { step(kt, std::true_type{}, std::false_type{}); ++kt; }
  step(kt, std::false_type{}, std::false_type{});
  conv_epilogue<T, BM, BN, WM, MI, NJ, NT>(p, acc, n0, lds, [&](int row) { const int m = m0 + row; return m < M ? (long)m : -1L; });
}

// ---- 3x3 stride-1 convolution with the input halo tile resident in LDS ---------------------------------------------------
// The implicit GEMM above stages a fresh 128-pixel x 64-channel slab for each of the nine taps: nine times the input bytes
// through L2 -> LDS.  Bytes in flight per CU are capped by LDS capacity, so for the narrow layers (64..128 channels, where
// the weight slab is small and the input slab dominates) that inflated traffic, not the MFMAs, sets the pace.
// Here an output tile is 8 rows x 16 pixels and its 10 x 18 input patch (64 channels = one 128-byte row per pixel, same
// chunk swizzle) is loaded ONCE per 64-channel slab; the nine taps read shifted windows of it (16 consecutive patch rows
// at any offset still hit 16 distinct bank groups).  K order is (channel slab, tap) instead of (tap, channel slab).
// Per K step only the BN x 64 weight slab streams in; pieces of the next slab's patch ride along in the first six taps.
struct HaloAux { float inv_tiles, inv_tx; int tiles, tx, nt; };

template <class T, int BN>
__global__ __launch_bounds__(256) void conv3x3_halo_kernel(const ConvP p, const HaloAux a) {
  constexpr int BM = 128, WM = 2, WN = 2, MI = 4, NJ = BN / WN / 16;
  constexpr int PW = 18, PROWS = 184;                  // 10 x 18 = 180 patch pixels, padded to whole 8-row DMA groups
  constexpr int XP = PROWS * 8, WS = BN * 8;           // uint4 per patch buffer / weight stage
  constexpr int WR = BN / 32;                          // weight rows per thread per step
  static_assert(sizeof(T) == 2, "16-bit storage only");
  static_assert(2 * XP * 16 >= BM * BN * 2, "epilogue tile must fit in the patch buffers");
  extern __shared__ __attribute__((aligned(16))) uint4 lds[];   // [2][XP] patches, then [2][WS] weight stages
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned lds_base = lds_addr(lds);

  int wg;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int mt = wg / a.nt, n0 = (wg - mt * a.nt) * BN;
  const int b = fdiv(mt, a.tiles, a.inv_tiles), trem = mt - b * a.tiles, ty = fdiv(trem, a.tx, a.inv_tx), tx = trem - ty * a.tx;
  const int h0 = ty * 8, w0 = tx * 16;

  // ---- loaders: thread -> (row within an 8-row group, 16-byte position); source chunk = position ^ swizzle(row)
  const int ppos = tid & 7, prow = tid >> 3;           // prow 0..31; rows prow + 32*i
  const char* xcur[6];                                 // patch rows prow + 32*i, i < 6 (wave 3 has no i = 5: rows >= 184)
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int pr = prow + 32 * i;
    const int py = pr / PW, px = pr - py * PW;
    const int ih = h0 - 1 + py, iw = w0 - 1 + px;
    const bool ok = pr < 180 && (unsigned)ih < (unsigned)p.Hin && (unsigned)iw < (unsigned)p.Win;
    const int chunk = ppos ^ swz<8>(pr);
    xcur[i] = ok ? reinterpret_cast<const char*>(p.s0.ptr) +
                       ((((long)b * p.s0.H + ih) * p.s0.W + iw) * (long)p.s0.cstride + p.s0.coff + chunk * 8) * (long)sizeof(T)
                 : reinterpret_cast<const char*>(&g_zero16);
  }
  unsigned xinc[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) xinc[i] = xcur[i] == reinterpret_cast<const char*>(&g_zero16) ? 0u : 128u;
  const char* wbase[WR];
#pragma unroll
  for (int i = 0; i < WR; ++i) {
    const int row = prow + 32 * i, n = n0 + row;
    const int chunk = ppos ^ swz<8>(row);
    wbase[i] = n < p.Cout ? reinterpret_cast<const char*>(p.w) + ((size_t)n * p.Kw + chunk * 8) * sizeof(T) : nullptr;
  }
  auto issue_w = [&](int stage, int tap, int c0) {     // weight slab of (tap, channels c0..c0+63)
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(2 * XP + stage * WS + wave * 64) * 16u);
    const long off = ((long)tap * p.Cin + c0) * (long)sizeof(T);
#pragma unroll
    for (int i = 0; i < WR; ++i)
      glds16(wbase[i] ? static_cast<const void*>(wbase[i] + off) : static_cast<const void*>(&g_zero16), dst + i * (256 * 16u));
  };
  auto issue_x = [&](int buf, int i) {                 // 32 patch rows (one 8-row group per wave) of the current xcur slab
    if (wave * 8 + 32 * i < PROWS) {
      const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(buf * XP + (32 * i + wave * 8) * 8) * 16u);
      glds16(xcur[i], dst);
    }
  };

  const int wm = wave & 1, wn0 = (wave >> 1) * (BN / WN);
  const int fr = lane & 15, fg = lane >> 4;
  const int nch = p.Cin >> 6, nsteps = 9 * nch;

  f32x4 acc[NJ][MI];
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int i = 0; i < MI; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};

  // prologue: patch of slab 0 and the first weight slab
#pragma unroll
  for (int i = 0; i < 6; ++i) issue_x(0, i);
#pragma unroll
  for (int i = 0; i < 6; ++i) xcur[i] += xinc[i];
  issue_w(0, 0, 0);

  int ci = 0, tap = 0;
  for (int t = 0; t < nsteps; ++t) {
    wait_vmcnt<0>();
    __syncthreads();                                   // step t's data landed for everyone; step t-1's buffers are free
    {                                                  // stream the next weight slab and a piece of the next patch
      int ntap = tap + 1, nci = ci;
      if (ntap == 9) { ntap = 0; ++nci; }
      if (t + 1 < nsteps) issue_w((t + 1) & 1, ntap, nci << 6);
      if (tap < 6 && ci + 1 < nch) { issue_x((ci + 1) & 1, tap); xcur[tap] += xinc[tap]; }
    }
    const uint4* ldsX = lds + (ci & 1) * XP;
    const uint4* ldsW = lds + 2 * XP + (t & 1) * WS;
    const int r = tap / 3, sft = tap - r * 3;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      uint4 xf[MI], wf[NJ];
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int row = (wm * 4 + i + r) * PW + fr + sft;
        xf[i] = ldsX[row * 8 + ((h * 4 + fg) ^ swz<8>(row))];
      }
#pragma unroll
      for (int j = 0; j < NJ; ++j) { const int row = wn0 + j * 16 + fr; wf[j] = ldsW[row * 8 + ((h * 4 + fg) ^ swz<8>(row))]; }
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int i = 0; i < MI; ++i) Mma<T>::run(wf[j], xf[i], acc[j][i]);
    }
    if (++tap == 9) { tap = 0; ++ci; }
  }

  conv_epilogue<T, BM, BN, WM, MI, NJ>(p, acc, n0, lds, [&](int row) {
    const int ho = h0 + (row >> 4), wo = w0 + (row & 15);
    return (ho < p.Ho && wo < p.Wo) ? ((long)b * p.Ho + ho) * p.Wo + wo : -1L;
  });
}

// ---- 3x3 stride-1 convolution for narrow layers (Cin, Cout <= 64): weights stationary in LDS, persistent blocks ----------
// With K = 9*Cin <= 576 a tile is nine tiny K steps; in the kernels above each step waits for a DMA issued one step
// earlier, so a tile costs nine memory latencies for ~1 us of MFMA work (these layers ran at 2-3x their HBM time).
// Here a block loads ALL nine taps of the weights once (<= 72 KB), then walks its share of the 8x16-pixel output tiles:
// the 10x18 input patch of tile i+1 streams in (LDS-DMA) while tile i runs its 144 MFMAs per wave straight from LDS and
// leaves through the staged epilogue.  One wait + three barriers per tile; the layer becomes an HBM stream
// (patch in, tile out).  One block of four waves per CU (LDS: weights + two patches + epilogue tile).
struct WsAux { float inv_tiles, inv_tx; int tiles, tx, total; };

template <class T, int CIN, int COUT>
__global__ __launch_bounds__(256) void conv3x3_ws_kernel(const ConvP p, const WsAux a) {
  constexpr int BM = 128, WM = 2, WN = 2, MI = 4, NJ = COUT / WN / 16;
  constexpr int E = 8, CPRW = CIN / E;                 // 16-byte chunks per pixel row: 8 (64 ch) or 4 (32 ch)
  constexpr int PW = 18, RPP = 256 / CPRW;             // patch width; rows covered by one pass of the 256 threads
  constexpr int XPASS = (180 + RPP - 1) / RPP, PROWS = XPASS * RPP;
  constexpr int XP = PROWS * CPRW;                     // uint4 per patch buffer
  constexpr int WROWS = 9 * COUT, WPASS = (WROWS + RPP - 1) / RPP;
  constexpr int WL = WPASS * RPP * CPRW;               // uint4 of resident weights
  constexpr int ET = BM * COUT * 2 / 16;               // uint4 of the epilogue tile
  constexpr int HS = CPRW / 4;                         // MFMA k-steps (32 channels) per tap
  static_assert(sizeof(T) == 2 && (CIN == 32 || CIN == 64) && (COUT == 32 || COUT == 64), "narrow 16-bit layers only");
  extern __shared__ __attribute__((aligned(16))) uint4 lds[];   // [WL] weights, [2][XP] patches, [ET] epilogue tile
  uint4* ldsW = lds;
  uint4* ldsE = lds + WL + 2 * XP;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned lds_base = lds_addr(lds);
  const int ppos = tid % CPRW, prow = tid / CPRW;

  // ---- resident weights: LDS row R = tap*COUT + n holds w[n][tap*CIN .. +CIN), chunk-swizzled like every other tile
#pragma unroll
  for (int i = 0; i < WPASS; ++i) {
    const int R = prow + RPP * i;
    const int tap = R / COUT, n = R - tap * COUT;
    const int chunk = ppos ^ swz<CPRW>(R);
    const void* src = (R < WROWS && n < p.Cout) ? static_cast<const void*>(reinterpret_cast<const char*>(p.w) +
                          ((size_t)n * p.Kw + tap * CIN + chunk * E) * sizeof(T)) : static_cast<const void*>(&g_zero16);
    glds16(src, __builtin_amdgcn_readfirstlane(lds_base + (unsigned)((RPP * i) * CPRW + wave * 64) * 16u));
  }

  auto issue_patch = [&](int tile, int buf) {          // the 10x18 input patch of output tile `tile`
    const int b = fdiv(tile, a.tiles, a.inv_tiles), trem = tile - b * a.tiles, ty = fdiv(trem, a.tx, a.inv_tx), tx = trem - ty * a.tx;
    const int h0 = ty * 8 - 1, w0 = tx * 16 - 1;
#pragma unroll
    for (int i = 0; i < XPASS; ++i) {
      const int pr = prow + RPP * i;
      const int py = pr / PW, px = pr - py * PW;
      const int ih = h0 + py, iw = w0 + px;
      const bool ok = pr < 180 && (unsigned)ih < (unsigned)p.Hin && (unsigned)iw < (unsigned)p.Win;
      const int chunk = ppos ^ swz<CPRW>(pr);
      const void* src = ok ? static_cast<const void*>(reinterpret_cast<const char*>(p.s0.ptr) +
                                 ((((long)b * p.s0.H + ih) * p.s0.W + iw) * (long)p.s0.cstride + p.s0.coff + chunk * E) * (long)sizeof(T))
                           : static_cast<const void*>(&g_zero16);
      glds16(src, __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(WL + buf * XP + (RPP * i) * CPRW + wave * 64) * 16u));
    }
  };

  const int wm = wave & 1, wn0 = (wave >> 1) * (COUT / WN);
  const int fr = lane & 15, fg = lane >> 4;
  int tile = blockIdx.x, buf = 0;
  if (tile < a.total) issue_patch(tile, 0);
  for (; tile < a.total; tile += gridDim.x, buf ^= 1) {
    wait_vmcnt<0>();
    __syncthreads();                                   // weights + this tile's patch landed; the other patch buffer is free
    if (tile + (int)gridDim.x < a.total) issue_patch(tile + gridDim.x, buf ^ 1);
    const uint4* ldsX = lds + WL + buf * XP;
    f32x4 acc[NJ][MI];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int i = 0; i < MI; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int r = tap / 3, sft = tap % 3;
#pragma unroll
      for (int h = 0; h < HS; ++h) {
        uint4 xf[MI], wf[NJ];
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          const int row = (wm * 4 + i + r) * PW + fr + sft;
          xf[i] = ldsX[row * CPRW + ((h * 4 + fg) ^ swz<CPRW>(row))];
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const int row = tap * COUT + wn0 + j * 16 + fr;
          wf[j] = ldsW[row * CPRW + ((h * 4 + fg) ^ swz<CPRW>(row))];
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int i = 0; i < MI; ++i) Mma<T>::run(wf[j], xf[i], acc[j][i]);
      }
    }
    const int b = fdiv(tile, a.tiles, a.inv_tiles), trem = tile - b * a.tiles, ty = fdiv(trem, a.tx, a.inv_tx), tx = trem - ty * a.tx;
    conv_epilogue<T, BM, COUT, WM, MI, NJ>(p, acc, 0, ldsE, [&](int row) {
      const int ho = ty * 8 + (row >> 4), wo = tx * 16 + (row & 15);
      return (ho < p.Ho && wo < p.Wo) ? ((long)b * p.Ho + ho) * p.Wo + wo : -1L;
    });
  }
}

template <class T> static bool supported_t(const ConvP& p) {
  const int E = 16 / (int)sizeof(T);
  auto src_ok = [&](const Src& s) { return s.C == 0 || (s.C % E == 0 && s.coff % E == 0 && s.cstride % E == 0 && ((uintptr_t)s.ptr & 15) == 0); };
  if (!src_ok(p.s0) || !src_ok(p.s1)) return false;
  if (p.Cin % E || p.Cout % 4 || p.out_coff % 4 || p.out_cstride % 4) return false;
  if (p.res && (p.res_coff % 4 || p.res_cstride % 4)) return false;
  if (p.s0.C + p.s1.C != p.Cin || p.Ktot != p.ks * p.ks * p.Cin * (p.split ? 2 : 1)) return false;
  if (p.split && (sizeof(T) != 2 || p.s0.shift < 0)) return false;     // split weights: 16-bit storage, no averaging loader
  if (p.Kw < (p.Ktot + 8 * E - 1) / (8 * E) * (8 * E)) return false;      // weight rows must cover whole K steps
  if ((long)p.B * p.Ho * p.Wo >= (1L << 31) || (long)p.Ho * p.Wo >= (1L << 22)) return false;
  return true;
}

bool conv_mfma_supported(int dt, const ConvP& p) {
  return dt == F32 ? supported_t<float>(p) : supported_t<f16_t>(p);
}

// Tile configuration per channel-tile width, tunable with CLEARCAM_CONV_CFG="bm128,ns128,bm64,ns64"
// (bm: 128|256 pixels per tile, ns: 2|3 stages).
static int g_cfg[4] = {-1, 0, 0, 0};

template <class T, int BM, int BN, int WM, bool SIMPLE, int CPRW, int NS, int NTH = 2 * BM>
static void launch_k(const ConvP& p, const ConvAux& a, int mtiles, hipStream_t stream) {
  constexpr size_t lds = (size_t)NS * (BM + BN) * CPRW * 16;
  static PerDevice once;                               // the attribute is per device (common.h)
  if (once.first(once.index())) {
    CC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_mfma_kernel<T, BM, BN, WM, SIMPLE, CPRW, NS, NTH>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  }
  note_launch(BM == 256 ? "conv_mfma_256" : (BM == 64 ? "conv_mfma_64" : (BM == 32 ? "conv_mfma_32" : "conv_mfma_128")), conv_mfma_kernel<T, BM, BN, WM, SIMPLE, CPRW, NS, NTH>, (long)mtiles * a.nt, NTH, lds);
  hipLaunchKernelGGL((conv_mfma_kernel<T, BM, BN, WM, SIMPLE, CPRW, NS, NTH>), dim3(mtiles * a.nt), dim3(NTH), lds, stream, p, a);
}

static int g_thin_k = -1;     // K (elements) at or below which the 64-byte-row / 5-blocks-per-CU variant is used

template <class T, int BN, bool SIMPLE>
static void launch_cfg(const ConvP& p, const ConvAux& a, int M, int bm, int ns, hipStream_t stream) {
  if (g_thin_k < 0) { const char* e = getenv("CLEARCAM_THIN_K"); g_thin_k = e ? atoi(e) : 256; }
  if (p.Ktot * (int)sizeof(T) <= g_thin_k * 2) {        // few K steps: latency-bound -> favour occupancy over step size
    launch_k<T, 128, BN, 2, SIMPLE, 4, 2>(p, a, (M + 127) / 128, stream);
    return;
  }
  if (bm == 256) {
    const int mt = (M + 255) / 256;
    if (ns == 3) launch_k<T, 256, BN, 4, SIMPLE, 8, 3>(p, a, mt, stream);
    else launch_k<T, 256, BN, 4, SIMPLE, 8, 2>(p, a, mt, stream);
  } else {
    const int mt = (M + 127) / 128;
    if (ns == 3) launch_k<T, 128, BN, 2, SIMPLE, 8, 3>(p, a, mt, stream);
    else launch_k<T, 128, BN, 2, SIMPLE, 8, 2>(p, a, mt, stream);
  }
}

template <class T, int TS> static void launch_big(const ConvP& p, const ConvAux& a, int M, hipStream_t stream) {
  constexpr size_t lds = (size_t)2 * (2 * TS) * 8 * 16;
  static PerDevice once;                               // the attribute is per device (common.h)
  if (once.first(once.index())) {
    CC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_big_kernel<T, TS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  }
  note_launch(TS == 256 ? "conv_big_256" : "conv_big_128", conv_big_kernel<T, TS>, (long)((M + TS - 1) / TS) * a.nt, 256, lds);
  hipLaunchKernelGGL((conv_big_kernel<T, TS>), dim3(((M + TS - 1) / TS) * a.nt), dim3(256), lds, stream, p, a);
}

void launch_conv_phase(int dt, const ConvP& p, const ConvAux& a, int M, hipStream_t stream);   // conv_phase.hip
bool conv_tile64_legal(const ConvP& p);                                                          // conv_tile64.hip
void launch_conv_tile64(int dt, const ConvP& p, hipStream_t stream);
bool conv_wave_legal(const ConvP& p);                                                            // conv_wave.hip
void launch_conv_wave(int dt, const ConvP& p, hipStream_t stream);
bool conv_stream_legal(const ConvP& p);                                                          // conv_stream.hip
void launch_conv_stream(int dt, const ConvP& p, hipStream_t stream);

template <class T, bool SIMPLE> static void launch_ts(const ConvP& p, const ConvAux& a, int bn, int M, hipStream_t stream) {
  if (g_cfg[0] < 0) {
    g_cfg[0] = 128; g_cfg[1] = 2; g_cfg[2] = 128; g_cfg[3] = 2;
    if (const char* e = getenv("CLEARCAM_CONV_CFG")) sscanf(e, "%d,%d,%d,%d", &g_cfg[0], &g_cfg[1], &g_cfg[2], &g_cfg[3]);
  }
  if (bn == 256) {
    if constexpr (SIMPLE && sizeof(T) == 2) {
      static int sched = -1;
      if (sched < 0) { const char* e = getenv("CLEARCAM_BIG_SCHED"); sched = e ? atoi(e) : 1; }
      if (p.variant == 7 || (sched == 2 && p.Cin % 64 == 0)) {
        CC_CHECK(p.Cin % 64 == 0 && p.Cout % 256 == 0, "eight-wave 256x256 kernel: Cin must be a multiple of 64 and Cout of 256");
        launch_conv_phase(TypeTag<T>::dt, p, a, M, stream); return;
      }
      if (sched) { launch_big<T, 256>(p, a, M, stream); return; }
    }
    launch_k<T, 256, 256, 4, SIMPLE, 8, 2>(p, a, (M + 255) / 256, stream);
  }
  else if (bn == 128) {
    if constexpr (SIMPLE && sizeof(T) == 2) {
      // the single-barrier, register-double-buffered schedule with 128x128 tiles: 4-5 % faster than the two-barrier loop
      // for K >= 1024 (3x3 with >= 128 channels, deep 1x1), 3-8 % slower for short K (per-layer A/B, YOLOv9-C B=64)
      static int mid = -1;
      if (mid < 0) { const char* e = getenv("CLEARCAM_MID_SCHED"); mid = e ? atoi(e) : 1024; }
      if ((mid > 0 && p.Ktot >= mid) || p.variant == 6) { launch_big<T, 128>(p, a, M, stream); return; }
    }
    launch_cfg<T, 128, SIMPLE>(p, a, M, g_cfg[0], g_cfg[1], stream);
  }
  else if (bn == 64) {
    // deep K on 64-wide channel tiles (the DDetect box branch's 3x3 256 -> 64 / 512 -> 64 entry convs): three LDS stages, i.e. the DMA
    // two K steps ahead - 206 us against 257 with two stages at 80x80, B = 64 (399 against 485 with split weights; r04c_head_split.txt)
    const int ns64 = (sizeof(T) == 2 && g_cfg[3] == 2 && p.Ktot >= 2048) ? 3 : g_cfg[3];
    launch_cfg<T, 64, SIMPLE>(p, a, M, g_cfg[2], ns64, stream);
  }
  else launch_k<T, 128, 32, 4, SIMPLE, 8, 2>(p, a, (M + 127) / 128, stream);
}

template <class T, int BN> static void launch_halo(const ConvP& p, hipStream_t stream) {
  constexpr size_t lds = (size_t)(2 * 184 * 8 + 2 * BN * 8) * 16;
  static PerDevice once;                               // the attribute is per device (common.h)
  if (once.first(once.index())) {
    CC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_halo_kernel<T, BN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  }
  HaloAux a{};
  a.tx = (p.Wo + 15) / 16; a.tiles = ((p.Ho + 7) / 8) * a.tx; a.nt = (p.Cout + BN - 1) / BN;
  a.inv_tiles = 1.0f / (float)a.tiles; a.inv_tx = 1.0f / (float)a.tx;
  note_launch("conv3x3_halo", conv3x3_halo_kernel<T, BN>, (long)p.B * a.tiles * a.nt, 256, lds);
  hipLaunchKernelGGL((conv3x3_halo_kernel<T, BN>), dim3(p.B * a.tiles * a.nt), dim3(256), lds, stream, p, a);
}

// 3x3 s1 p1, one source, whole 64-channel slabs.  Measured per layer (YOLOv9-C, B=64): 6-7 % faster than the generic
// kernel for 128/256-channel layers whose image is covered by whole 8x16 tiles and whose Cout is a multiple of 128;
// slower for Cout = 320 (64-wide channel tiles reload the patch five times), for ragged coverage (40x40) and for the
// 64-channel layers (those are latency-chain bound, see conv3x3_ws_kernel).  Tests force it with variant 3.
static bool halo_legal(const ConvP& p) {
  return !p.split && p.ks == 3 && p.stride == 1 && p.pad == 1 && p.s1.C == 0 && p.s0.shift == 0 && p.Cin % 64 == 0 && p.Hin == p.Ho && p.Win == p.Wo;
}
static bool halo_applicable(const ConvP& p) {
  static int on = -1;
  // Off by default since round 2: with the single-barrier 128x128 schedule and the eight-wave kernel in place it no longer wins
  // (whole plan, batch 64: 11.67 ms without it against 11.70 with it; single frame 1.30 against 1.34 ms, where it also kept its
  // layers off the few-tile configuration because of its (channel slab, tap) K order).  CLEARCAM_HALO=1 brings it back.
  if (on < 0) { const char* e = getenv("CLEARCAM_HALO"); on = e ? atoi(e) : 0; }
  if (!on || !halo_legal(p)) return false;
  const long covered = (long)((p.Ho + 7) / 8 * 8) * ((p.Wo + 15) / 16 * 16);
  return p.Cin >= 128 && p.Cout % 128 == 0 && (long)p.Ho * p.Wo == covered;
}

template <class T, int CIN, int COUT> static void launch_ws(const ConvP& p, hipStream_t stream) {
  constexpr int CPRW = CIN / 8, RPP = 256 / CPRW;
  constexpr int XP = (180 + RPP - 1) / RPP * RPP * CPRW, WL = (9 * COUT + RPP - 1) / RPP * RPP * CPRW;
  constexpr size_t lds = (size_t)(WL + 2 * XP + 128 * COUT * 2 / 16) * 16;
  static PerDevice pd;                                 // attribute and CU count per device ordinal (common.h)
  const int pdi = pd.index();
  if (pd.first(pdi))
    CC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_ws_kernel<T, CIN, COUT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int cus = pd.cu_count(pdi);
  WsAux a{};
  a.tx = (p.Wo + 15) / 16; a.tiles = ((p.Ho + 7) / 8) * a.tx; a.total = p.B * a.tiles;
  a.inv_tiles = 1.0f / (float)a.tiles; a.inv_tx = 1.0f / (float)a.tx;
  const int blocks_per_cu = (int)(160 * 1024 / lds);           // persistent: as many blocks as fit, each walks total/grid tiles
  note_launch("conv3x3_ws", conv3x3_ws_kernel<T, CIN, COUT>, (long)a.total, 256, lds, std::min(a.total, cus * std::max(1, blocks_per_cu)));
  hipLaunchKernelGGL((conv3x3_ws_kernel<T, CIN, COUT>), dim3(std::min(a.total, cus * std::max(1, blocks_per_cu))), dim3(256), lds, stream, p, a);
}

// narrow 3x3 s1 p1 layers: Cin in {32, 64}, Cout <= 64 (a multiple of 8), one source
static bool ws_legal(const ConvP& p) {
  return !p.split && p.ks == 3 && p.stride == 1 && p.pad == 1 && p.s1.C == 0 && p.s0.shift == 0 && (p.Cin == 32 || p.Cin == 64) &&
         p.Cout <= 64 && p.Cout % 8 == 0 && p.Hin == p.Ho && p.Win == p.Wo;
}
static bool ws_applicable(const ConvP& p) {
  static int on = -1;
  if (on < 0) { const char* e = getenv("CLEARCAM_WS"); on = e ? atoi(e) : 1; }
  if (!on || !ws_legal(p)) return false;
  const long covered = (long)((p.Ho + 7) / 8 * 8) * ((p.Wo + 15) / 16 * 16);
  if ((long)p.Ho * p.Wo * 5 < covered * 4) return false; // 8x16 tiles must not waste more than a fifth of their pixels
  // Measured (YOLOv9-C, B=64): 32-channel layers (52 KB of LDS, three blocks and three patches in flight per CU) run 27 %
  // faster than the generic kernel.  With 64 input channels the 74 KB of weights leave room for one block per CU - one
  // patch in flight, ~3.7 us per tile plus ~40 us to fill 256 weight copies - which only pays off once a block walks
  // many tiles: -7 % at 160x160 (50 tiles per block), +20 % at 80x80 (12 tiles per block).  (Holding the weights in
  // registers instead - 144 VGPRs per wave, two blocks per CU - was tried and was slower still: the register pressure
  // pushes address temporaries to scratch, and every scratch reload waits on vmcnt, i.e. on the patch prefetch.)
  if (p.Cin == 32 && p.Cout <= 32) return true;
  const long tiles = (long)p.B * (covered / 128);
  return tiles >= 256L * 32;
}
template <class T> static void launch_ws_t(const ConvP& p, hipStream_t stream) {
  if (p.Cin == 64) { if (p.Cout > 32) launch_ws<T, 64, 64>(p, stream); else launch_ws<T, 64, 32>(p, stream); }
  else { if (p.Cout > 32) launch_ws<T, 32, 64>(p, stream); else launch_ws<T, 32, 32>(p, stream); }
}

static bool halo_applicable(const ConvP& p);
static bool ws_legal(const ConvP& p);
template <class T> static void launch_ws_t(const ConvP& p, hipStream_t stream);

// Few-tile layers (a single frame, or the 20x20 maps of a batch): the grid does not fill the chip, one block per CU at best, and a
// K step is then paced by the round trip of the DMA issued one step earlier (~0.6 us per step measured at batch 1: a 36-step 3x3
// 256 -> 256 layer took 31 us on 8 blocks).  Here the generic kernel runs with narrow channel tiles (more blocks -> more CUs) and
// three to six LDS stages (prefetch distance two to five steps: the queue is never drained, counted vmcnt only), which is
// what the otherwise idle LDS is for.  CLEARCAM_SMALL_TILES=0 disables; tests force it with variant 9.
template <class T> static bool launch_small(const ConvP& p, int M, hipStream_t stream, bool force) {
  static int on = -1;
  if (on < 0) { const char* e = getenv("CLEARCAM_SMALL_TILES"); on = e ? atoi(e) : 1; }
  if (!force && !on) return false;
  const long mt = (M + 127) / 128, b32 = mt * ((p.Cout + 31) / 32), b64 = mt * ((p.Cout + 63) / 64);
  int bn = 0;
  if (b32 <= 512) bn = 32; else if (b64 <= 512) bn = 64;
  if (force && !bn) bn = 64;
  if (!bn) return false;
  bool deep = bn == 32 && b32 <= 256;                  // at most one block per CU anyway: six stages (120 KB), prefetch distance five
  if (p.variant >= 91 && p.variant <= 93) { bn = p.variant == 93 ? 64 : 32; deep = p.variant == 92; }   // tuning sweeps: 91 32/4, 92 32/6, 93 64/3
  {   // Round 6: smaller pixel tiles on 256 threads for the layers of a single frame.  A K step of the 128 x 32 configuration costs ~0.43 us however
      // deep the prefetch: five LDS-DMA issues per wave (60-100 cycles each), two dependent fragment-read batches, one barrier - and a 40 x 40
      // layer is 104 blocks on 256 CUs.  32 x 32 tiles: one pixel + one weight DMA per wave and step, four times the blocks; 64 x 32: three
      // DMAs, twice the blocks.  Measured per shape at batch 1 (profiles/r06r_small_tiles_b1.txt, cc_conv_bench): 3x3 256 -> 256 at 40 x 40
      // 18.2 -> 13.3 us, 3x3 512 -> 64 at 20 x 20 29.2 -> 17.8, 1x1 256 -> 256 at 40 x 40 5.0 -> 3.5; the sum over the 103 launches 1087 -> ~800 us.
      // 32 x 32 up to 800 blocks, 64 x 32 up to 512; six stages for K >= 4096 on 32 x 32, four otherwise.  CLEARCAM_SMALL_PIX=0 disables;
      // tests force them with variants 94-97.  Same K order and MFMA as every tile kernel: same bits, so the batch a frame arrives in does not matter.
    static const int small_pix = [] { const char* e = getenv("CLEARCAM_SMALL_PIX"); return e ? atoi(e) : 1; }();
    const long n32 = (p.Cout + 31) / 32, t32 = (long)((M + 31) / 32) * n32, t64 = (long)((M + 63) / 64) * n32;
    int pix = 0, st = 4;
    if (p.variant == 0 && small_pix) { if (t32 <= 800) { pix = 32; st = p.Ktot >= 4096 ? 6 : 4; } else if (t64 <= 512) pix = 64; }
    if (p.variant == 94) { pix = 64; st = 6; } else if (p.variant == 95) pix = 64; else if (p.variant == 96) { pix = 32; st = 6; } else if (p.variant == 97) pix = 32;
    if (pix) {
      ConvAux a{};
      a.nt = (int)n32;
      a.inv_hw = 1.0f / (float)(p.Ho * p.Wo); a.inv_wo = 1.0f / (float)p.Wo;
      const bool simple = p.s1.C == 0 && p.s0.shift == 0 && p.ks <= 3;
      a.is1x1 = simple && p.ks == 1 && p.stride == 1 && p.pad == 0 && p.Hin == p.Ho && p.Win == p.Wo;
      const int mt64 = (M + 63) / 64, mt32 = (M + 31) / 32;
      if (pix == 32 && st == 6) { if (simple) launch_k<T, 32, 32, 2, true, 8, 6, 256>(p, a, mt32, stream); else launch_k<T, 32, 32, 2, false, 8, 6, 256>(p, a, mt32, stream); }
      else if (pix == 32) { if (simple) launch_k<T, 32, 32, 2, true, 8, 4, 256>(p, a, mt32, stream); else launch_k<T, 32, 32, 2, false, 8, 4, 256>(p, a, mt32, stream); }
      else if (st == 6) { if (simple) launch_k<T, 64, 32, 4, true, 8, 6, 256>(p, a, mt64, stream); else launch_k<T, 64, 32, 4, false, 8, 6, 256>(p, a, mt64, stream); }
      else { if (simple) launch_k<T, 64, 32, 4, true, 8, 4, 256>(p, a, mt64, stream); else launch_k<T, 64, 32, 4, false, 8, 4, 256>(p, a, mt64, stream); }
      CC_HIP(hipGetLastError());
      return true;
    }
  }
  ConvAux a{};
  a.nt = (p.Cout + bn - 1) / bn;
  a.inv_hw = 1.0f / (float)(p.Ho * p.Wo); a.inv_wo = 1.0f / (float)p.Wo;
  const bool simple = p.s1.C == 0 && p.s0.shift == 0 && p.ks <= 3;
  a.is1x1 = simple && p.ks == 1 && p.stride == 1 && p.pad == 0 && p.Hin == p.Ho && p.Win == p.Wo;
  if (deep) { if (simple) launch_k<T, 128, 32, 4, true, 8, 6>(p, a, (int)mt, stream); else launch_k<T, 128, 32, 4, false, 8, 6>(p, a, (int)mt, stream); }
  else if (bn == 32) { if (simple) launch_k<T, 128, 32, 4, true, 8, 4>(p, a, (int)mt, stream); else launch_k<T, 128, 32, 4, false, 8, 4>(p, a, (int)mt, stream); }
  else { if (simple) launch_k<T, 128, 64, 2, true, 8, 3>(p, a, (int)mt, stream); else launch_k<T, 128, 64, 2, false, 8, 3>(p, a, (int)mt, stream); }
  CC_HIP(hipGetLastError());
  return true;
}

int g_stream_override = -1;                            // cc_dev_set("stream", v): A/B inside one process (tests, tools/dev)

template <class T> static void launch_t(const ConvP& p, hipStream_t stream) {
  const int M = p.B * p.Ho * p.Wo;
  if constexpr (sizeof(T) == 2) {
    // (layers the halo-resident kernel takes at any batch size keep it: its K order is (channel slab, tap), every other kernel's
    //  (tap, channel), and a frame's result must not depend on the batch it arrives in - test_batch_invariance_and_determinism)
    //  narrow 3x3 layers on a few tiles go to the weights-stationary kernel instead: 6.5-7.3 us against 8.5-10.5 at batch 1 (same K order))
    {   // thin 1x1 layers whose roof is HBM: all weights resident in registers, pixels streamed through an LDS ring (conv_stream.hip).
        // CLEARCAM_STREAM=0 disables; CLEARCAM_STREAM_MIN_PIX = fewest output pixels it is taken for; tests force it with variant 10.
      // both values come from ONE thread-safe static (two plain statics let a second thread see stream_on set and stream_min still 0)
      struct StreamEnv { int on, min_pix; };
      static const StreamEnv env = [] {
        const char* e = getenv("CLEARCAM_STREAM"); const char* m = getenv("CLEARCAM_STREAM_MIN_PIX");
        return StreamEnv{e ? atoi(e) : 1, m ? atoi(m) : 200000};   // measured (B = 64, profiles/r05f_stream_ab.txt): 1.13-1.5x at 160x160 and 80x80, a tie at 40x40 (102 400 pixels), 0.75x at 20x20
      }();
      const int stream_min = env.min_pix;
      const int on = g_stream_override >= 0 ? g_stream_override : env.on;
      if (p.variant == 10 || (p.variant == 0 && on && M >= stream_min && conv_stream_legal(p))) {
        launch_conv_stream(TypeTag<T>::dt, p, stream);
        return;
      }
    }
    const bool few_narrow = p.variant == 0 && ws_legal(p) && (long)((M + 127) / 128) * ((p.Cout + 31) / 32) <= 512;
    if (few_narrow) { launch_ws_t<T>(p, stream); CC_HIP(hipGetLastError()); return; }
    if (((p.variant == 0 && !halo_applicable(p)) || p.variant == 9 || (p.variant >= 91 && p.variant <= 97)) && launch_small<T>(p, M, stream, p.variant >= 9)) return;
    {   // narrow 3x3 layers: one autonomous wave per 2x16-pixel sub-tile over LDS-resident weights (conv_wave.hip).
        // CLEARCAM_WAVE=0 falls back to the cooperative kernels below; tests force it with variant 8.
      static int wave_on = -1;
      if (wave_on < 0) { const char* e = getenv("CLEARCAM_WAVE"); wave_on = e ? atoi(e) : 1; }
      // measured (B=64): 64 -> 64 at 80x80 51 us vs 57 (weights-stationary) / 59 (generic), at 160x160 204 vs 223 / 262; 32 -> 32
      // ties the weights-stationary kernel (61 vs 58 us) and a handful of sub-tiles (batch 1) is better served by the tile kernels
      // 64 -> 64 from one round of 8 x 32-pixel tiles on: the persistent tile kernel with resident fragment-order weights (conv_tile64.hip,
      // round 6; CLEARCAM_TILE64=0 disables, tests force it with variant 12).  Same K order as every kernel here: same bits.
      if constexpr (sizeof(T) == 2) {
        static const int tile64_on = [] { const char* e = getenv("CLEARCAM_TILE64"); return e ? atoi(e) : 1; }();
        // ... and only where its 256-pixel tiles cover the map without much waste (the better of 16 x 16 and 8 x 32 tiles within 1.25x of the map:
        // 80 x 80 and 160 x 160 exactly; a 40 x 40 map is 9 tiles of 16 x 16 = 1.44x and runs 21 us against 18 on the wave-autonomous kernel)
        const long per_frame = std::min((long)((p.Ho + 7) / 8) * ((p.Wo + 31) / 32), (long)((p.Ho + 15) / 16) * ((p.Wo + 15) / 16));
        const long tiles64 = (long)p.B * per_frame;
        const bool covers = per_frame * 256 * 4 <= (long)p.Ho * p.Wo * 5;
        if (p.variant == 12 || (p.variant == 0 && tile64_on && conv_tile64_legal(p) && tiles64 >= 256 && covers)) {
          CC_CHECK(conv_tile64_legal(p), "3x3 64 -> 64 tile kernel: shape not eligible");
          launch_conv_tile64(TypeTag<T>::dt, p, stream);
          return;
        }
      }
      const long subtiles = (long)p.B * ((p.Ho + 1) / 2) * ((p.Wo + 15) / 16);
      if (p.variant == 8 || (p.variant == 0 && wave_on && conv_wave_legal(p) && p.Cin == 64 && subtiles >= 4096)) {
        CC_CHECK(conv_wave_legal(p), "wave-autonomous 3x3: shape not eligible");
        launch_conv_wave(TypeTag<T>::dt, p, stream);
        return;
      }
    }
    if (p.variant == 4 || (p.variant == 0 && ws_applicable(p))) {
      CC_CHECK(ws_legal(p), "weights-stationary 3x3: shape not eligible");
      launch_ws_t<T>(p, stream);
      CC_HIP(hipGetLastError());
      return;
    }
    if (p.variant == 3 || (p.variant == 0 && halo_applicable(p))) {
      CC_CHECK(halo_legal(p), "halo-resident 3x3: shape not eligible");
      auto padded = [&](int bn) { return (p.Cout + bn - 1) / bn * bn; };
      if (padded(64) < padded(128)) launch_halo<T, 64>(p, stream); else launch_halo<T, 128>(p, stream);
      CC_HIP(hipGetLastError());
      return;
    }
  }
  auto padded = [&](int bn) { return (p.Cout + bn - 1) / bn * bn; };
  int bn = 128;
  bool use_phase = false, use_two = false;
  if (padded(64) < padded(bn)) bn = 64;
  if (padded(32) < padded(bn)) bn = 32;
  {   // 256x256 tiles / four-wave single-barrier schedule (conv_big_kernel) for wide, deep, GEMM-shaped layers.
      // One block per CU, so whole rounds of 256 tiles matter: taken when K >= 1024 and the last round wastes <= 12 %
      // (measured: -7..-18 % on such layers, +3..+16 % when K = 512 or when 400-800 tiles leave a half-empty round).
      // CLEARCAM_BIG_TILE=0 disables, -1 forces it for every Cout % 256 == 0 layer (tests use variant 5).
    static int big = -2, phase = -2;
    if (big == -2) { const char* e = getenv("CLEARCAM_BIG_TILE"); big = e ? atoi(e) : 1; }
    if (phase == -2) { const char* e = getenv("CLEARCAM_PHASE"); phase = e ? atoi(e) : 2; }
    if (sizeof(T) == 2 && p.Cout % 256 == 0) {
      const long tiles = (long)((M + 255) / 256) * (p.Cout / 256), rounds = (tiles + 255) / 256;
      const bool fits = p.Ktot >= 1024 && tiles >= 256 && rounds * 256 * 100 <= tiles * 112;
      // layers with a residual keep the 128x128 kernel: their f32 read-modify-write epilogue needs another block to hide behind
      if ((big > 0 && fits && !p.res && p.variant == 0) || big < 0 || p.variant == 5 || p.variant == 7) bn = 256;
      // The eight-wave two-group kernel (conv_phase.hip).  Per-layer A/B on YOLOv9-C B=64 and on the CLIP GEMMs (profiles/r02c):
      // 5-18 % faster than the kernels above for layers with >= 400 tiles of 256x256 and for every CLIP GEMM (with or without
      // the f32 residual epilogue), slower below ~400 tiles (one block per CU: a 100-tile layer leaves 156 CUs idle).
      // CLEARCAM_PHASE=0 disables, 1 forces it for every eligible layer.
      const bool simple1 = p.s1.C == 0 && p.s0.shift == 0 && p.ks <= 3;
      // one block of eight waves per CU vs two blocks of four with 128x128 tiles: whichever fills its last round of CUs better wins,
      // the eight-wave kernel being 12-20 % faster at equal fill (per-layer A/B, profiles/r02c_*)
      const long t128 = (long)((M + 127) / 128) * ((p.Cout + 127) / 128);
      const double fill256 = (double)tiles / (double)(rounds * 256), fill128 = (double)t128 / (double)(((t128 + 511) / 512) * 512);
      const bool phase_rule = tiles >= 128 && fill256 * 1.2 >= fill128;
      // ... and the 1x1 convs that read a Concat (two sources, one possibly upsampled: the first conv of the neck's ELAN blocks)
      const bool concat1 = p.s1.C > 0 && p.ks == 1 && p.stride == 1 && p.pad == 0 && p.s0.C % 64 == 0 && p.s1.C % 64 == 0;
      use_two = concat1;
      if ((p.variant == 0 || (p.variant == 7 && concat1)) && (simple1 || concat1) && p.Cin % 64 == 0 && (phase == 1 || p.variant == 7 || (phase != 0 && phase_rule))) { bn = 256; use_phase = true; }
    }
  }
  if (p.variant == 6) bn = 128;
  ConvAux a{};
  a.nt = (p.Cout + bn - 1) / bn;
  a.inv_hw = 1.0f / (float)(p.Ho * p.Wo); a.inv_wo = 1.0f / (float)p.Wo;

  const bool simple = p.s1.C == 0 && p.s0.shift == 0 && p.ks <= 3;
  a.is1x1 = simple && p.ks == 1 && p.stride == 1 && p.pad == 0 && p.Hin == p.Ho && p.Win == p.Wo;
  a.two = use_phase && use_two;
  if (use_phase) { if constexpr (sizeof(T) == 2) launch_conv_phase(TypeTag<T>::dt, p, a, M, stream); }
  else if (simple) launch_ts<T, true>(p, a, bn, M, stream); else launch_ts<T, false>(p, a, bn, M, stream);
  CC_HIP(hipGetLastError());
}

void launch_conv_mfma(int dt, const ConvP& p, hipStream_t stream) {
  CC_CHECK(conv_mfma_supported(dt, p), "conv_mfma: unsupported shape/alignment");
  if (dt == F32) launch_t<float>(p, stream);
  else if (dt == F16) launch_t<f16_t>(p, stream);
  else launch_t<bf16_t>(p, stream);
}

void launch_conv(int dt, const ConvP& p, hipStream_t stream) {
  if (p.s0.shift < 0) { launch_conv_adown(dt, p, stream); return; }
  if (conv_mfma_supported(dt, p)) launch_conv_mfma(dt, p, stream);
  else launch_conv_direct(dt, p, stream);
}

}  // namespace cc

// The eight-wave, two-group ("ping-pong") 256 x 256 implicit-GEMM kernel: its own translation unit (compile time).
#include "conv_tile.h"

namespace cc {

typedef int srd_t __attribute__((ext_vector_type(4)));
// LDS-DMA through a buffer descriptor: 32-bit per-lane byte offset, rows past `bytes` (and the 0xffffffff "no such row"
// offset of halo taps) read as zero in hardware - no zero page, no 64-bit address VGPRs.
__device__ __forceinline__ srd_t make_srd(const void* base, unsigned bytes) {
  const unsigned long long b = (unsigned long long)base;
  srd_t r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(b & 0xffffffffu));
  r[1] = __builtin_amdgcn_readfirstlane((int)((b >> 32) & 0xffffu));
  r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
  r[3] = 0x00020000;
  return r;
}
__device__ __forceinline__ void bdma16(unsigned voff, srd_t srd, unsigned lds_wave_byte_addr) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" : : "v"(voff), "s"(srd), "s"(lds_wave_byte_addr) : "memory", "m0");
}

// ---- 256 x 256 tile, EIGHT waves in two groups that alternate an LDS/DMA segment and an MFMA segment -----------------------
// The four-wave kernel above leaves one wave per SIMD: whenever that wave reads LDS, issues DMA or waits at its barrier the
// SIMD's matrix pipe idles.  Here two waves share each SIMD (waves w and w+4) and run half a phase apart, so that while one
// reads the operands of its next quadrant and issues DMA, the other issues 16 MFMAs with raised priority (the 8-phase schedule
// of cdna_hip_programming.md §5, re-derived for this loader).  A wave owns 128 pixels x 64 channels (32 accumulator fragments);
// a K tile (64 halfs) is four phases, one accumulator quadrant (64 pixels x 32 channels x K 64 = 16 MFMAs) each:
//     phase 1  read W-subtile 0 + P-subtile 0 | DMA | barrier | MFMA (P0,W0) | barrier
//     phase 2  read W-subtile 1               | DMA | barrier | MFMA (P0,W1) | barrier
//     phase 3  read P-subtile 1               | DMA | barrier | MFMA (P1,W1) | barrier
//     phase 4  read W-subtile 0 again          | DMA | vmcnt | barrier | MFMA (P1,W0) | barrier   (one W buffer: registers)
// LDS holds two K tiles (2 x 64 KB); each is four half-tiles of 128 rows (pixel rows 0-127 / 128-255, weight rows likewise),
// 2 DMA instructions per thread each, ONE half-tile issued per phase into a slot whose last reader finished at least one full
// phase earlier: with E / O the even / odd tile in LDS and E' / O' the tiles two steps ahead
//     phase of tile t:   1           2           3           4
//     issue:             P1(t+1)     W0(t+1)     W1(t+1)     P0(t+2)
//     wait:                                                  vmcnt(2): tile t+1 landed
// pixel pieces (HBM / far L2) have four to five phases to land, weight pieces (L2 hits) one to two, and the queue is never
// drained in the main loop.  Every wave's ds_reads are retired
// (lgkmcnt(0)) BEFORE the barrier that ends its read segment, which is what allows a slot to be re-used one phase later.
template <class T, int SCHED>
__global__ __launch_bounds__(512) void conv_phase_kernel(const ConvP p, const ConvAux a) {
  constexpr int BM = 256, BN = 256, NT = 512, MI = 8, NJ = 4;
  constexpr int E = 8, CPRW = 8, BK = 64, RPP = NT / CPRW, XR = BM / RPP, WR = BN / RPP;   // 64 rows per pass, 4 passes per operand
  constexpr int STAGE = (BM + BN) * CPRW;              // uint4 per K tile
  static_assert(sizeof(T) == 2 && XR == 4 && WR == 4, "16-bit storage, four DMA passes per operand");
  extern __shared__ __attribute__((aligned(16))) uint4 lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int grp = wave >> 2, wq = wave & 3;            // pixel half (= ping-pong group) / channel quarter
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

  // ---- loader state: pixel rows prow + 64 i with per-row pointers (halo taps -> zero page), weight rows from one pointer ------
  const int ppos = tid % CPRW, prow = tid / CPRW;
  const int chunk = ppos ^ swz<CPRW>(prow);            // swz(prow + 64 i) is the same for every i
  // Per pixel row: the byte address of (its window's top-left pixel, channel coff) and the 9-bit tap-validity mask.  They are
  // only needed when the K walk crosses into the next filter tap (every Cin/64 tiles), so they live in the 24 KB of LDS behind
  // the two K-tile stages instead of in 12 VGPRs that the accumulators need.
  constexpr bool BUF = SCHED == 2;                     // DMA through buffer descriptors (32-bit offsets, hardware zero fill)
  struct RowInfo { const char* ptr; unsigned long long aux; };   // aux: tap-validity mask, or (two sources) the row's address in source 1
  RowInfo* rinfo = reinterpret_cast<RowInfo*>(lds + 2 * STAGE) + tid * XR;
  const char* cur[XR]; unsigned inc[XR];
  unsigned cur32[XR];                                  // BUF: byte offsets from p.s0.ptr
  const srd_t srd_x = make_srd(p.s0.ptr, (unsigned)a.x_bytes), srd_w = make_srd(p.w, (unsigned)a.w_bytes);
#pragma unroll
  for (int i = 0; i < XR; ++i) {
    const int m = m0 + prow + RPP * i;
    RowInfo ri;
    if (a.two) {                                       // Concat folded into the loader (detection/yolov9.py:151-155; Upsample :285-292 as index >> shift)
      const int mm = m < M ? m : 0;
      const int b = fdiv(mm, hw, a.inv_hw), rem = mm - b * hw, ho = fdiv(rem, p.Wo, a.inv_wo), wo = rem - ho * p.Wo;
      const long i0 = ((long)b * p.s0.H + (ho >> p.s0.shift)) * p.s0.W + (wo >> p.s0.shift);
      const long i1 = ((long)b * p.s1.H + (ho >> p.s1.shift)) * p.s1.W + (wo >> p.s1.shift);
      ri.ptr = m < M ? reinterpret_cast<const char*>(p.s0.ptr) + (i0 * p.s0.cstride + p.s0.coff) * (long)sizeof(T) : nullptr;
      ri.aux = (unsigned long long)(reinterpret_cast<const char*>(p.s1.ptr) + (i1 * p.s1.cstride + p.s1.coff) * (long)sizeof(T));
    } else if (a.is1x1) {
      ri.ptr = reinterpret_cast<const char*>(p.s0.ptr) + ((size_t)m * p.s0.cstride + p.s0.coff) * sizeof(T);
      ri.aux = m < M ? 1u : 0u;
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
      ri.aux = m < M ? vm : 0u;
      ri.ptr = reinterpret_cast<const char*>(p.s0.ptr) +
               ((((long)b * p.s0.H + h0) * p.s0.W + w0) * (long)p.s0.cstride + p.s0.coff) * (long)sizeof(T);
    }
    rinfo[i] = ri;
  }
  // weights: Cout is a multiple of 256 here, every row exists; row n0 + prow + 64 i, advancing BK halfs per K tile
  const char* wptr = reinterpret_cast<const char*>(p.w) + ((size_t)(n0 + prow) * p.Kw + chunk * E) * sizeof(T);
  const size_t wpass = (size_t)RPP * p.Kw * sizeof(T);
  // K walk: Cin is a multiple of 64 (checked on the host), so a K tile never straddles two filter taps and every thread
  // changes tap at the same tile
  int kc = 0, tap = 0;                                 // channel offset of the current tile inside its tap; tap index r * ks + s
  auto retarget = [&]() {
    const int kr = a.two ? 0 : tap / p.ks, ks_ = a.two ? 0 : tap - kr * p.ks;
    const long delta = ((long)(kr * p.s0.W + ks_) * p.s0.cstride + kc + chunk * E) * (long)sizeof(T);
#pragma unroll
    for (int i = 0; i < XR; ++i) {
      const RowInfo ri = rinfo[i];
      const bool ok = a.two ? ri.ptr != nullptr : (bool)((ri.aux >> tap) & 1u);
      const char* src = (a.two && tap) ? reinterpret_cast<const char*>(ri.aux) : ri.ptr;   // two sources: `tap` counts the source
      if constexpr (BUF) cur32[i] = ok ? (unsigned)((src + delta) - reinterpret_cast<const char*>(p.s0.ptr)) : 0xffffffffu;
      else cur[i] = ok ? src + delta : reinterpret_cast<const char*>(&g_zero16);
      inc[i] = ok ? (unsigned)(BK * sizeof(T)) : 0u;
    }
  };
  retarget();
  auto advance_p = [&]() {
    kc += BK;
    if (kc == (a.two ? (tap ? p.s1.C : p.s0.C) : p.Cin)) { kc = 0; ++tap; retarget(); }
    else {
#pragma unroll
      for (int i = 0; i < XR; ++i) { if constexpr (BUF) cur32[i] += inc[i]; else cur[i] += inc[i]; }
    }
  };
  const unsigned wave_lds = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(wave * 64) * 16u);
  // half-tile h of the pixels / weights of a K tile -> passes 2h, 2h+1 of that operand's slab in `stage`
  auto issue_p = [&](unsigned stage_bytes, int h) {
    const unsigned sb = wave_lds + stage_bytes;
    if constexpr (BUF) { bdma16(cur32[2 * h], srd_x, sb + (2 * h) * (NT * 16u)); bdma16(cur32[2 * h + 1], srd_x, sb + (2 * h + 1) * (NT * 16u)); }
    else { glds16_m0(cur[2 * h], sb + (2 * h) * (NT * 16u)); glds16_m0(cur[2 * h + 1], sb + (2 * h + 1) * (NT * 16u)); }
  };
  auto issue_w = [&](unsigned stage_bytes, int h) {
    const unsigned sb = wave_lds + stage_bytes + (unsigned)(BM * CPRW) * 16u;
    if constexpr (BUF) {
      const unsigned w32 = (unsigned)(wptr - reinterpret_cast<const char*>(p.w));
      bdma16(w32 + (2 * h) * (unsigned)wpass, srd_w, sb + (2 * h) * (NT * 16u)); bdma16(w32 + (2 * h + 1) * (unsigned)wpass, srd_w, sb + (2 * h + 1) * (NT * 16u));
    } else { glds16_m0(wptr + (2 * h) * wpass, sb + (2 * h) * (NT * 16u)); glds16_m0(wptr + (2 * h + 1) * wpass, sb + (2 * h + 1) * (NT * 16u)); }
  };

  const int fr = lane & 15, fg = lane >> 4;
  const int nkt = (p.Ktot + BK - 1) / BK;
  f32x4 acc[NJ][MI];
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int i = 0; i < MI; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  uint4 pa[8], wb[4];                                  // pixel subtile (4 fragments x 2 k-halves), one weight subtile (2 x 2): 48 VGPRs
  uint4 wb1[4];                                        // SCHED 1 keeps both weight subtiles in registers
  // Fragment addresses: row = 16-aligned base + fr, so the chunk swizzle (row >> 1) & 7 depends on the lane only and every
  // fragment of a stage is ONE of two per-lane byte addresses (k-half 0 / 1) plus a compile-time offset (ds_read offset field).
  const int sw = (fr >> 1) & 7;
  const char* ldsb = reinterpret_cast<const char*>(lds);
  const char* pb[2] = {ldsb + (grp * 128 + fr) * 128 + ((fg ^ sw) * 16), ldsb + (grp * 128 + fr) * 128 + (((4 + fg) ^ sw) * 16)};
  const char* wbp[2] = {ldsb + (BM + wq * 64 + fr) * 128 + ((fg ^ sw) * 16), ldsb + (BM + wq * 64 + fr) * 128 + (((4 + fg) ^ sw) * 16)};
  auto read_p = [&](unsigned so, int ps) {             // so: byte offset of the stage (0 or 64 KB), one v_add per base per tile
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        pa[kh * 4 + i] = *reinterpret_cast<const uint4*>(pb[kh] + so + (ps * 64 + i * 16) * 128);
  };
  auto read_w = [&](unsigned so, int ws) {
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        wb[kh * 2 + j] = *reinterpret_cast<const uint4*>(wbp[kh] + so + (ws * 32 + j * 16) * 128);
  };
  // `mid` runs after the first four MFMAs: with CLEARCAM_PHASE_FLAGS bit 2 the phase's DMA pieces are issued there, under the
  // wave's own matrix work, instead of lengthening the read segment the other group's MFMAs are waiting behind
  auto mma_w = [&](int ps, int ws, const uint4 (&wv)[4], auto&& mid) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
#pragma unroll
        for (int i = 0; i < 4; ++i) Mma<T>::run(wv[kh * 2 + j], pa[kh * 4 + i], acc[ws * 2 + j][ps * 4 + i]);
        if (kh == 0 && j == 0) { __builtin_amdgcn_sched_barrier(0); mid(); __builtin_amdgcn_sched_barrier(0); }
      }
    __builtin_amdgcn_s_setprio(0);
  };
  auto mma = [&](int ps, int ws, auto&& mid) { mma_w(ps, ws, wb, mid); };
  auto read_w1 = [&](unsigned so) {
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        wb1[kh * 2 + j] = *reinterpret_cast<const uint4*>(wbp[kh] + so + (32 + j * 16) * 128);
  };
  auto end_read = [&]() {                              // retire this wave's LDS reads, then meet the other group
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  auto end_mma = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  int pt = 0, wt = 0;                                  // K tile the pixel / weight walk points at
  auto step_w = [&]() { if (wt + 1 < nkt) { wptr += BK * sizeof(T); ++wt; } };
  auto step_p = [&]() { if (pt + 1 < nkt) { advance_p(); ++pt; } };
  constexpr unsigned SB = STAGE * 16u;                 // bytes per stage
  auto nothing = [] {};
  // DMA is issued unconditionally: past the last K tile the walks stop advancing, so the surplus pieces re-read the last tile
  // into stage slots nobody reads any more (straight-line phases; the queue is drained before the epilogue re-uses the LDS).
  if constexpr (SCHED >= 1) {
    // ---- schedule 1: LDS reads are NOT retired before the barrier (they complete while the wave waits there and under the first
    // MFMAs: the compiler's own lgkmcnt ladder), so a slot is re-used two phases after its last read at the earliest.  Both
    // weight subtiles stay in registers (no re-read in phase 4), which frees a stage's weight slots after phase 2:
    //     phase of tile t:   1          2          3        4
    //     read (stage so):   W0, P0     W1         P1       -
    //     issue:             P0(t+1)    P1(t+1)    -        W0(t+2), W1(t+2) -> so ; vmcnt(4): tile t+1 complete
    // weights travel five phases ahead of their use, pixels three to four.
    auto sync = [&]() { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); };
    issue_p(0, 0); issue_p(0, 1); step_p();
    issue_w(0, 0); issue_w(0, 1); step_w();
    issue_w(SB, 0); issue_w(SB, 1); step_w();
    wait_vmcnt<4>();
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();        // group 1 runs one barrier behind group 0 from here on
    __builtin_amdgcn_sched_barrier(0);
    for (int t = 0; t < nkt; ++t) {
      const unsigned so = (t & 1) ? SB : 0u, no = SB - so;
      read_w(so, 0); read_p(so, 0);
      issue_p(no, 0);
      sync(); mma_w(0, 0, wb, nothing); sync();
      read_w1(so);
      issue_p(no, 1); step_p();
      sync(); mma_w(0, 1, wb1, nothing); sync();
      read_p(so, 1);
      sync(); mma_w(1, 1, wb1, nothing); sync();
      issue_w(so, 0); issue_w(so, 1); step_w();
      wait_vmcnt<4>();                                                       // tile t+1 has landed
      sync(); mma_w(1, 0, wb, nothing); sync();
    }
  } else {
  // ---- schedule 0: reads retired before the barrier, one weight buffer (W subtile 0 is read again in phase 4) ------------------
  issue_p(0, 0); issue_p(0, 1); step_p();
  issue_w(0, 0); issue_w(0, 1); step_w();
  issue_p(SB, 0);
  wait_vmcnt<2>();
  __builtin_amdgcn_s_barrier();
  if (grp == 1) __builtin_amdgcn_s_barrier();          // group 1 runs one barrier behind group 0 from here on
  __builtin_amdgcn_sched_barrier(0);
  if (a.flags & 24) {
    // ABLATIONS (timing only, wrong results): bit 3 = MFMA segments only (operands of tile 0 stay in registers, no LDS reads, no
    // DMA in the loop); bit 4 = LDS reads + MFMA, no DMA in the loop.  Barriers as in the real loop.
    read_w(0, 0); read_p(0, 0);
    for (int t = 0; t < nkt; ++t) {
      const unsigned so = (t & 1) ? SB : 0u;
      if (a.flags & 16) { read_w(so, 0); read_p(so, 0); }
      end_read(); mma(0, 0, nothing); end_mma();
      if (a.flags & 16) read_w(so, 1);
      end_read(); mma(0, 1, nothing); end_mma();
      if (a.flags & 16) read_p(so, 1);
      end_read(); mma(1, 1, nothing); end_mma();
      if (a.flags & 16) read_w(so, 0);
      end_read(); mma(1, 0, nothing); end_mma();
    }
  } else
  for (int t = 0; t < nkt; ++t) {
    const unsigned so = (t & 1) ? SB : 0u, no = SB - so;
    read_w(so, 0); read_p(so, 0);
    issue_p(no, 1); step_p();                                                // pixels of tile t+1, rows 128..255
    end_read(); mma(0, 0, nothing); end_mma();
    read_w(so, 1);
    issue_w(no, 0);                                                          // weights of tile t+1, rows 0..127
    end_read(); mma(0, 1, nothing); end_mma();
    read_p(so, 1);
    issue_w(no, 1); step_w();                                                // weights of tile t+1, rows 128..255
    end_read(); mma(1, 1, nothing); end_mma();
    read_w(so, 0);                                                           // W subtile 0 again (one weight buffer in registers)
    issue_p(so, 0);                                                          // pixels of tile t+2, rows 0..127 (this stage's were last read in phase 3)
    wait_vmcnt<2>();                                                         // tile t+1 has landed
    end_read(); mma(1, 0, nothing); end_mma();
  }
  }
  wait_vmcnt<0>();
  if (grp == 0) __builtin_amdgcn_s_barrier();          // group 0 waits for group 1's last segment
  conv_epilogue<T, BM, BN, 2, MI, NJ, NT, true>(p, acc, n0, lds, [&](int row) { const int m = m0 + row; return m < M ? (long)m : -1L; });
}

// ---- the same kernel as a PERSISTENT loop over tiles (one block per CU) ------------------------------------------------------
// A 256x256 tile costs ~11 us outside its K loop (block dispatch, loader set-up + the latency of the first DMA pieces, the
// activation epilogue, the staged store): a third of the time of a K = 1024 GEMM tile and most of a K = 256 one.  Here a block
// walks tiles b, b + grid, ...; when a tile's K loop ends it sets up the NEXT tile's loader and issues its first twelve DMA
// pieces, and only then runs the epilogue - straight from registers (8-byte / 16-byte stores, no LDS, no barrier), so the LDS
// stages already belong to the next tile.  Dispatch gaps and the first-DMA latency disappear behind the epilogue VALU work.
// Schedule 1 of conv_phase_kernel inside; same results bit for bit.
// MEASURED (MI355X, bf16, r02): SLOWER than one tile per block on every shape tried - CLIP fc GEMM 736 vs 632 us, head 3x3 554 vs
// 510 us, 1x1 256->256 @160x160 666 vs 512 us.  The register epilogue stores 32-byte row segments (a 16x16 MFMA tile is 16
// channels wide) where the LDS-staged one writes whole 512-byte rows, and that costs more than dispatch + first-DMA latency
// saved.  Kept behind CLEARCAM_PHASE_FLAGS=256 for the record and as the base of a staged-epilogue variant; not the default.
template <class T>
__global__ __launch_bounds__(512) void conv_phase_persist_kernel(const ConvP p, const ConvAux a) {
  constexpr int BM = 256, BN = 256, NT = 512, MI = 8, NJ = 4;
  constexpr int E = 8, CPRW = 8, BK = 64, RPP = NT / CPRW, XR = BM / RPP;
  constexpr int STAGE = (BM + BN) * CPRW;
  constexpr unsigned SB = STAGE * 16u;
  extern __shared__ __attribute__((aligned(16))) uint4 lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int grp = wave >> 2, wq = wave & 3;
  const int M = p.B * p.Ho * p.Wo, hw = p.Ho * p.Wo;
  const unsigned lds_base = lds_addr(lds);
  const int ntiles = a.ntiles;
  const int ppos = tid % CPRW, prow = tid / CPRW;
  const int chunk = ppos ^ swz<CPRW>(prow);
  struct RowInfo { const char* ptr; unsigned long long aux; };
  RowInfo* rinfo = reinterpret_cast<RowInfo*>(lds + 2 * STAGE) + tid * XR;
  const char* cur[XR]; unsigned inc[XR];
  const char* wptr = nullptr;
  const size_t wpass = (size_t)RPP * p.Kw * sizeof(T);
  int kc = 0, tap = 0, pt = 0, wt = 0, m0 = 0, n0 = 0;
  const int nkt = (p.Ktot + BK - 1) / BK;

  auto retarget = [&]() {
    const int kr = a.two ? 0 : tap / p.ks, ks_ = a.two ? 0 : tap - kr * p.ks;
    const long delta = ((long)(kr * p.s0.W + ks_) * p.s0.cstride + kc + chunk * E) * (long)sizeof(T);
#pragma unroll
    for (int i = 0; i < XR; ++i) {
      const RowInfo ri = rinfo[i];
      const bool ok = a.two ? ri.ptr != nullptr : (bool)((ri.aux >> tap) & 1u);
      const char* src = (a.two && tap) ? reinterpret_cast<const char*>(ri.aux) : ri.ptr;
      cur[i] = ok ? src + delta : reinterpret_cast<const char*>(&g_zero16);
      inc[i] = ok ? (unsigned)(BK * sizeof(T)) : 0u;
    }
  };
  auto setup_tile = [&](int vb) {                      // loader state of tile `vb` (XCD-aware order as in the one-tile kernel)
    const int q = ntiles >> 3, r = ntiles & 7, xcd = vb & 7, idx = vb >> 3;
    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int mt_ = wg / a.nt;
    m0 = mt_ * BM; n0 = (wg - mt_ * a.nt) * BN;
#pragma unroll
    for (int i = 0; i < XR; ++i) {
      const int m = m0 + prow + RPP * i;
      RowInfo ri;
      if (a.two) {
        const int mm = m < M ? m : 0;
        const int b = fdiv(mm, hw, a.inv_hw), rem = mm - b * hw, ho = fdiv(rem, p.Wo, a.inv_wo), wo = rem - ho * p.Wo;
        const long i0 = ((long)b * p.s0.H + (ho >> p.s0.shift)) * p.s0.W + (wo >> p.s0.shift);
        const long i1 = ((long)b * p.s1.H + (ho >> p.s1.shift)) * p.s1.W + (wo >> p.s1.shift);
        ri.ptr = m < M ? reinterpret_cast<const char*>(p.s0.ptr) + (i0 * p.s0.cstride + p.s0.coff) * (long)sizeof(T) : nullptr;
        ri.aux = (unsigned long long)(reinterpret_cast<const char*>(p.s1.ptr) + (i1 * p.s1.cstride + p.s1.coff) * (long)sizeof(T));
      } else if (a.is1x1) {
        ri.ptr = reinterpret_cast<const char*>(p.s0.ptr) + ((size_t)m * p.s0.cstride + p.s0.coff) * sizeof(T);
        ri.aux = m < M ? 1u : 0u;
      } else {
        const int mm = m < M ? m : 0;
        const int b = fdiv(mm, hw, a.inv_hw), rem = mm - b * hw, ho = fdiv(rem, p.Wo, a.inv_wo), wo = rem - ho * p.Wo;
        const int h0 = ho * p.stride - p.pad, w0 = wo * p.stride - p.pad;
        unsigned hm = 0, wmk = 0;
#pragma unroll
        for (int rr = 0; rr < 3; ++rr) {
          hm |= (unsigned)(rr < p.ks && (unsigned)(h0 + rr) < (unsigned)p.Hin) << rr;
          wmk |= (unsigned)(rr < p.ks && (unsigned)(w0 + rr) < (unsigned)p.Win) << rr;
        }
        const unsigned vm = ((hm & 1u) ? wmk : 0u) | ((hm & 2u) ? wmk << p.ks : 0u) | ((hm & 4u) ? wmk << (2 * p.ks) : 0u);
        ri.aux = m < M ? vm : 0u;
        ri.ptr = reinterpret_cast<const char*>(p.s0.ptr) +
                 ((((long)b * p.s0.H + h0) * p.s0.W + w0) * (long)p.s0.cstride + p.s0.coff) * (long)sizeof(T);
      }
      rinfo[i] = ri;
    }
    wptr = reinterpret_cast<const char*>(p.w) + ((size_t)(n0 + prow) * p.Kw + chunk * E) * sizeof(T);
    kc = 0; tap = 0; pt = 0; wt = 0;
    retarget();
  };
  auto advance_p = [&]() {
    kc += BK;
    if (kc == (a.two ? (tap ? p.s1.C : p.s0.C) : p.Cin)) { kc = 0; ++tap; retarget(); }
    else {
#pragma unroll
      for (int i = 0; i < XR; ++i) cur[i] += inc[i];
    }
  };
  const unsigned wave_lds = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(wave * 64) * 16u);
  auto issue_p = [&](unsigned stage_bytes, int h) {
    const unsigned sb = wave_lds + stage_bytes;
    glds16_m0(cur[2 * h], sb + (2 * h) * (NT * 16u)); glds16_m0(cur[2 * h + 1], sb + (2 * h + 1) * (NT * 16u));
  };
  auto issue_w = [&](unsigned stage_bytes, int h) {
    const unsigned sb = wave_lds + stage_bytes + (unsigned)(BM * CPRW) * 16u;
    glds16_m0(wptr + (2 * h) * wpass, sb + (2 * h) * (NT * 16u)); glds16_m0(wptr + (2 * h + 1) * wpass, sb + (2 * h + 1) * (NT * 16u));
  };
  auto step_w = [&]() { if (wt + 1 < nkt) { wptr += BK * sizeof(T); ++wt; } };
  auto step_p = [&]() { if (pt + 1 < nkt) { advance_p(); ++pt; } };
  auto first_pieces = [&]() {                          // tile 0 of the K walk complete, the weights of tile 1 on their way
    issue_p(0, 0); issue_p(0, 1); step_p();
    issue_w(0, 0); issue_w(0, 1); step_w();
    issue_w(SB, 0); issue_w(SB, 1); step_w();
  };

  const int fr = lane & 15, fg = lane >> 4;
  const int sw = (fr >> 1) & 7;
  const char* ldsb = reinterpret_cast<const char*>(lds);
  const char* pb[2] = {ldsb + (grp * 128 + fr) * 128 + ((fg ^ sw) * 16), ldsb + (grp * 128 + fr) * 128 + (((4 + fg) ^ sw) * 16)};
  const char* wbp[2] = {ldsb + (BM + wq * 64 + fr) * 128 + ((fg ^ sw) * 16), ldsb + (BM + wq * 64 + fr) * 128 + (((4 + fg) ^ sw) * 16)};
  uint4 pa[8], wb[4], wb1[4];
  f32x4 acc[NJ][MI];
  auto read_p = [&](unsigned so, int ps) {
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
      for (int i = 0; i < 4; ++i) pa[kh * 4 + i] = *reinterpret_cast<const uint4*>(pb[kh] + so + (ps * 64 + i * 16) * 128);
  };
  auto read_w = [&](unsigned so, int ws, uint4 (&dst)[4]) {
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
      for (int j = 0; j < 2; ++j) dst[kh * 2 + j] = *reinterpret_cast<const uint4*>(wbp[kh] + so + (ws * 32 + j * 16) * 128);
  };
  auto mma_w = [&](int ps, int ws, const uint4 (&wv)[4]) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) Mma<T>::run(wv[kh * 2 + j], pa[kh * 4 + i], acc[ws * 2 + j][ps * 4 + i]);
    __builtin_amdgcn_s_setprio(0);
  };
  auto sync = [&]() { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); };

  int vb = blockIdx.x;
  setup_tile(vb);
  first_pieces();
  for (; vb < ntiles; vb += gridDim.x) {
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int i = 0; i < MI; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    wait_vmcnt<0>();                                   // the first pieces of this tile (and the previous tile's stores) are done
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();        // group 1 runs one barrier behind group 0 through the K loop
    __builtin_amdgcn_sched_barrier(0);
    for (int t = 0; t < nkt; ++t) {
      const unsigned so = (t & 1) ? SB : 0u, no = SB - so;
      read_w(so, 0, wb); read_p(so, 0);
      issue_p(no, 0);
      sync(); mma_w(0, 0, wb); sync();
      read_w(so, 1, wb1);
      issue_p(no, 1); step_p();
      sync(); mma_w(0, 1, wb1); sync();
      read_p(so, 1);
      sync(); mma_w(1, 1, wb1); sync();
      issue_w(so, 0); issue_w(so, 1); step_w();
      wait_vmcnt<4>();
      sync(); mma_w(1, 0, wb); sync();
    }
    wait_vmcnt<0>();                                   // this wave's surplus pieces have landed: its stage slices are free
    if (grp == 0) __builtin_amdgcn_s_barrier();        // everybody is out of the K loop
    __builtin_amdgcn_sched_barrier(0);
    const int em0 = m0, en0 = n0;
    if (vb + (int)gridDim.x < ntiles) { setup_tile(vb + gridDim.x); first_pieces(); }   // next tile's DMA runs under the epilogue
    __builtin_amdgcn_sched_barrier(0);
    long mrow[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) { const int m = em0 + grp * 128 + i * 16 + fr; mrow[i] = m < M ? (long)m : -1L; }
    const int nb = en0 + wq * 64 + fg * 4;
    if (p.act == 1) direct_tile<T, 1, MI, NJ>(p, acc, mrow, nb);
    else if (p.act == 2) direct_tile<T, 2, MI, NJ>(p, acc, mrow, nb);
    else if (p.act == 3) direct_tile<T, 3, MI, NJ>(p, acc, mrow, nb);
    else if (p.act == 4) direct_tile<T, 4, MI, NJ>(p, acc, mrow, nb);
    else direct_tile<T, 0, MI, NJ>(p, acc, mrow, nb);
  }
}

template <class T> static void launch_phase(const ConvP& p, const ConvAux& a, int M, hipStream_t stream) {
  constexpr size_t lds = (size_t)2 * 512 * 8 * 16 + 512 * 4 * 16;     // two K tiles of (256 + 256) rows x 128 bytes + the per-row loader table
  static bool configured = false;
  if (!configured) {
    CC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_phase_kernel<T, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_phase_kernel<T, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_phase_kernel<T, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    configured = true;
  }
  static int flags = -1;
  if (flags < 0) { const char* e = getenv("CLEARCAM_PHASE_FLAGS"); flags = e ? atoi(e) : 32; }   // 32: schedule 1 (default); 0: schedule 0; 8 / 16: timing ablations of schedule 0
  ConvAux b = a; b.flags = flags;
  const size_t xb = (size_t)p.B * p.s0.H * p.s0.W * p.s0.cstride * sizeof(T), wbytes = (size_t)p.Cout * p.Kw * sizeof(T);
  const bool buf_ok = xb < ((size_t)1 << 32) - 256 && wbytes < ((size_t)1 << 32) - 256;
  b.x_bytes = (unsigned)xb; b.w_bytes = (unsigned)wbytes;
  if (flags & 256) {
    static int cus = 0;
    if (!cus) {
      CC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_phase_persist_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      int dev = 0; hipDeviceProp_t pr;
      CC_HIP(hipGetDevice(&dev)); CC_HIP(hipGetDeviceProperties(&pr, dev));
      cus = pr.multiProcessorCount;
    }
    b.ntiles = ((M + 255) / 256) * a.nt;
    hipLaunchKernelGGL((conv_phase_persist_kernel<T>), dim3(std::min(b.ntiles, cus)), dim3(512), lds, stream, p, b);
    return;
  }
  if ((flags & 64) && buf_ok && !a.two) { hipLaunchKernelGGL((conv_phase_kernel<T, 2>), dim3(((M + 255) / 256) * a.nt), dim3(512), lds, stream, p, b); return; }
  if (flags & 32) hipLaunchKernelGGL((conv_phase_kernel<T, 1>), dim3(((M + 255) / 256) * a.nt), dim3(512), lds, stream, p, b);
  else hipLaunchKernelGGL((conv_phase_kernel<T, 0>), dim3(((M + 255) / 256) * a.nt), dim3(512), lds, stream, p, b);
}

void launch_conv_phase(int dt, const ConvP& p, const ConvAux& a, int M, hipStream_t stream) {
  if (dt == F16) launch_phase<f16_t>(p, a, M, stream);
  else launch_phase<bf16_t>(p, a, M, stream);
}

}  // namespace cc

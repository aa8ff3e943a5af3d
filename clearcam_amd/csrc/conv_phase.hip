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
  int kc = 0, vt = 0;                                  // channel offset of the current tile inside its tap; virtual tap (tap index r * ks + s, doubled for split weights; the source index of a Concat)
  auto retarget = [&]() {
    const int tap = vt >> p.split;                       // split weights: virtual taps 2t, 2t+1 read filter tap t's channels (hi / lo plane)
    const int kr = a.two ? 0 : tap / p.ks, ks_ = a.two ? 0 : tap - kr * p.ks;
    const long delta = ((long)(kr * p.s0.W + ks_) * p.s0.cstride + kc + chunk * E) * (long)sizeof(T);
#pragma unroll
    for (int i = 0; i < XR; ++i) {
      const RowInfo ri = rinfo[i];
      const bool ok = a.two ? ri.ptr != nullptr : (bool)((ri.aux >> tap) & 1u);
      const char* src = (a.two && (vt & 1)) ? reinterpret_cast<const char*>(ri.aux) : ri.ptr;   // two sources: the low bit of `vt` counts the source
      if constexpr (BUF) cur32[i] = ok ? (unsigned)((src + delta) - reinterpret_cast<const char*>(p.s0.ptr)) : 0xffffffffu;
      else cur[i] = ok ? src + delta : reinterpret_cast<const char*>(&g_zero16);
      inc[i] = ok ? (unsigned)(BK * sizeof(T)) : 0u;
    }
  };
  retarget();
  auto advance_p = [&]() {
    kc += BK;
    if (kc == (a.two ? ((vt & 1) ? p.s1.C : p.s0.C) : p.Cin)) { kc = 0; ++vt; retarget(); }
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

// The persistent-loop form of this kernel (next tile's first DMA pieces under a wave-private LDS-staged epilogue) is in
// conv_persist.hip.  The r02 persistent form with a REGISTER epilogue (32-byte store segments) measured slower than one tile per
// block on every shape (CLIP fc GEMM 736 vs 632 us) and was removed.
bool conv_persist_ok(const ConvP& p);
void launch_conv_persist(int dt, int mm, int abl, const ConvP& p, const ConvAux& a, int M, hipStream_t stream);
int g_phase_flags_override = -1;                       // cc_dev_set("phase_flags", v): A/B inside one process (tools/dev)

template <class T> static void launch_phase(const ConvP& p, const ConvAux& a, int M, hipStream_t stream) {
  constexpr size_t lds = (size_t)2 * 512 * 8 * 16 + 512 * 4 * 16;     // two K tiles of (256 + 256) rows x 128 bytes + the per-row loader table
  static PerDevice once;                               // the attribute is per device (common.h)
  if (once.first(once.index())) {
    CC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_phase_kernel<T, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_phase_kernel<T, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_phase_kernel<T, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  }
  static int env_flags = -2;
  // CLEARCAM_PHASE_FLAGS: unset = the default rule below; 32: schedule 1 one tile per block; 0: schedule 0; 8 / 16: timing ablations of
  // schedule 0; 64: buffer-descriptor DMA; 512: persistent loop with the wave-staged epilogue (conv_persist.hip) for every
  // eligible layer, + 1024: on v_mfma_f32_32x32x16, + 2048: drain the previous tile's stores before the next K loop, bits 12-15:
  // its timing ablations (development)
  if (env_flags == -2) { const char* e = getenv("CLEARCAM_PHASE_FLAGS"); env_flags = e ? atoi(e) : -1; }
  int flags = g_phase_flags_override >= 0 ? g_phase_flags_override : env_flags;
  if (flags < 0) {
    // Default: the persistent loop where a block walks at least ~2.5 tiles (same results bit for bit; r03 A/B on MI355X: CLIP GEMMs
    // +3-6 %, 1x1 convs at 80x80 / 160x160 +4-10 %, 3x3 256->256 at 80x80 +-1 %), one tile per block below that (400-tile layers at
    // 40x40: the persistent form measured 2-4 % slower - its second round is as ragged, and the first has no dispatch skew)
    // 1x1 layers take it from one full round on (their tiles are short - K <= 512 at 40x40 - so the next tile's first DMA under the
    // store epilogue is a larger share: 256 -> 256 at 40x40, 400 tiles: 34.0 -> 30.2 us, 512 -> 256: 44.2 -> 41.7; r03p)
    const long tiles = (long)((M + 255) / 256) * a.nt;
    flags = ((tiles >= 640 || (a.is1x1 && tiles > 256)) && conv_persist_ok(p)) ? 512 : 32;
  }
  ConvAux b = a; b.flags = flags;
  const size_t xb = (size_t)p.B * p.s0.H * p.s0.W * p.s0.cstride * sizeof(T), wbytes = (size_t)p.Cout * p.Kw * sizeof(T);
  const bool buf_ok = xb < ((size_t)1 << 32) - 256 && wbytes < ((size_t)1 << 32) - 256;
  b.x_bytes = (unsigned)xb; b.w_bytes = (unsigned)wbytes;
  if ((flags & 512) && conv_persist_ok(p)) { launch_conv_persist(TypeTag<T>::dt, (flags & 1024) ? 1 : 0, (flags >> 12) & 15, p, b, M, stream); return; }
  const long tiles = (long)((M + 255) / 256) * a.nt;                   // the note names the variant that is actually launched
  if ((flags & 64) && buf_ok && !a.two) {
    note_launch("conv_phase", conv_phase_kernel<T, 2>, tiles, 512, lds);
    hipLaunchKernelGGL((conv_phase_kernel<T, 2>), dim3(tiles), dim3(512), lds, stream, p, b);
    return;
  }
  if (flags & 32) { note_launch("conv_phase", conv_phase_kernel<T, 1>, tiles, 512, lds); hipLaunchKernelGGL((conv_phase_kernel<T, 1>), dim3(tiles), dim3(512), lds, stream, p, b); }
  else { note_launch("conv_phase", conv_phase_kernel<T, 0>, tiles, 512, lds); hipLaunchKernelGGL((conv_phase_kernel<T, 0>), dim3(tiles), dim3(512), lds, stream, p, b); }
}

void launch_conv_phase(int dt, const ConvP& p, const ConvAux& a, int M, hipStream_t stream) {
  if (dt == F16) launch_phase<f16_t>(p, a, M, stream);
  else launch_phase<bf16_t>(p, a, M, stream);
}

}  // namespace cc

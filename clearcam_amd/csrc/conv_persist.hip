// The eight-wave 256 x 256 implicit-GEMM kernel of conv_phase.hip as a PERSISTENT loop over tiles with a wave-private,
// LDS-staged epilogue (round 3).
//
// One tile per block costs ~11 us outside its K loop: block dispatch, loader set-up, the latency of the first DMA pieces, the
// activation epilogue and the block-wide staged store (two __syncthreads, 128 KB through LDS) - a third of a K = 1024 GEMM tile.
// The r02 persistent form hid dispatch + first-DMA latency behind a REGISTER epilogue and lost, because a 16x16 MFMA tile is 16
// channels wide and its stores were 32-byte row segments.  Here the epilogue goes through LDS again, but wave by wave:
//   * a wave owns 128 pixels x 64 channels of the tile; 32 pixels x 64 channels (4 KB of 16-bit values) at a time are written
//     to the wave's own 4 KB of LDS (chunk-swizzled ds_write_b64) and read back as 16 bytes per lane so that each pixel's 64
//     channels leave as ONE 128-byte line (8 lanes x 16 B): no block barrier, no other wave involved;
//   * those 8 x 4 KB live in the PIXEL half of K stage 1, the one region of the stages the next tile's first twelve DMA pieces
//     (all of K tile 0, the weights of K tile 1) do not write - so the next tile's loader is set up and its first pieces are in
//     flight BEFORE the epilogue starts and land under it; the first pixel piece into that region is issued after the barrier
//     that opens the next K loop, which every wave reaches only with its epilogue behind it;
//   * f32 outputs / residual adds (the CLIP out-proj and mlp-proj GEMMs: x = x + W h, models/objects.py:120,127) take the same
//     route with 16 pixels x 64 channels of f32 per round: residual read and output written as whole 256-byte rows.
// MM selects the matrix instruction: 0 = v_mfma_f32_16x16x32 (the accumulation grouping of every other kernel in this library),
// 1 = v_mfma_f32_32x32x16 (half as many matrix instructions per K tile, 32 cycles each instead of ~17 for half the work).  Same
// LDS image and chunk swizzle for both; only the fragment addresses and the accumulator -> (pixel, channel) map differ.
#include "conv_tile.h"

namespace cc {

typedef __attribute__((ext_vector_type(16))) float f32x16;

template <class T> struct Mma32;
template <> struct Mma32<bf16_t> {
  static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x16& acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
  }
};
template <> struct Mma32<f16_t> {
  static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x16& acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0);
  }
};

template <class T, int ACT> __device__ __forceinline__ float act_bias(float osc, float x, float b, float slope) {
  float v = activate<T, ACT>(__builtin_fmaf(x, osc, b));       // osc = 1 unless the weights are split (conv_tile.h out_scale)
  if constexpr (ACT == 3) v = v > 0.f ? v : slope * v;
  return v;
}

// ABL (development, timing only - results are WRONG with any bit set): 1 = no DMA in the K loop, 2 = no LDS fragment reads in the
// K loop, 4 = no MFMA, 8 = no barriers in the K loop, 16 = no epilogue, 32 = no global stores in the 16-bit epilogue, 64 = the next
// tile's first DMA pieces are not issued (no DMA latency at the loop top).
template <class T, int MM, int ABL = 0>
__global__ __launch_bounds__(512) void conv_persist_kernel(const ConvP p, const ConvAux a) {
  constexpr int BM = 256, BN = 256, NT = 512;
  constexpr int E = 8, CPRW = 8, BK = 64, RPP = NT / CPRW, XR = BM / RPP;
  constexpr int STAGE = (BM + BN) * CPRW;
  constexpr unsigned SB = STAGE * 16u;
  static_assert(sizeof(T) == 2, "16-bit storage");
  extern __shared__ __attribute__((aligned(16))) uint4 lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform: row bases, staging offsets, the group test live in SGPRs
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
  int kc = 0, vt = 0, pt = 0, wt = 0, m0 = 0, n0 = 0;
  const int nkt = (p.Ktot + BK - 1) / BK;
  const float osc = out_scale(p);

  auto retarget = [&]() {
    const int tap = vt >> p.split;                       // split weights: virtual taps 2t, 2t+1 read filter tap t's channels (hi / lo plane)
    const int kr = a.two ? 0 : tap / p.ks, ks_ = a.two ? 0 : tap - kr * p.ks;
    const long delta = ((long)(kr * p.s0.W + ks_) * p.s0.cstride + kc + chunk * E) * (long)sizeof(T);
#pragma unroll
    for (int i = 0; i < XR; ++i) {
      const RowInfo ri = rinfo[i];
      const bool ok = a.two ? ri.ptr != nullptr : (bool)((ri.aux >> tap) & 1u);
      const char* src = (a.two && (vt & 1)) ? reinterpret_cast<const char*>(ri.aux) : ri.ptr;
      cur[i] = ok ? src + delta : reinterpret_cast<const char*>(&g_zero16);
      inc[i] = ok ? (unsigned)(BK * sizeof(T)) : 0u;
    }
  };
  auto setup_tile = [&](int vb) {                      // loader state of tile `vb` (XCD-aware order as in the one-tile kernel)
    int tid_o = tid;                                   // opaque copy: keeps what is derived from it out of the K loop's live set
    asm volatile("" : "+v"(tid_o));
    const int prow = tid_o / CPRW;
    const int q = ntiles >> 3, r = ntiles & 7, xcd = vb & 7, idx = vb >> 3;
    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int mt_ = wg / a.nt;
    m0 = mt_ * BM; n0 = (wg - mt_ * a.nt) * BN;
#pragma unroll
    for (int i = 0; i < XR; ++i) {
      const int m = m0 + prow + RPP * i;
      RowInfo ri;
      if (a.two) {                                     // Concat folded into the loader (detection/yolov9.py:151-155; Upsample :285-292 as index >> shift)
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
    kc = 0; vt = 0; pt = 0; wt = 0;
    retarget();
  };
  auto advance_p = [&]() {
    kc += BK;
    if (kc == (a.two ? ((vt & 1) ? p.s1.C : p.s0.C) : p.Cin)) { kc = 0; ++vt; retarget(); }
    else {
#pragma unroll
      for (int i = 0; i < XR; ++i) cur[i] += inc[i];
    }
  };
  const unsigned wave_lds = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(wave * 64) * 16u);
  bool in_loop = false;                                // ABL 1 drops the DMA pieces issued inside the K loop only
  auto issue_p = [&](unsigned stage_bytes, int h) {
    if constexpr (ABL & 1) { if (in_loop) return; }
    const unsigned sb = wave_lds + stage_bytes;
    glds16_m0(cur[2 * h], sb + (2 * h) * (NT * 16u)); glds16_m0(cur[2 * h + 1], sb + (2 * h + 1) * (NT * 16u));
  };
  auto issue_w = [&](unsigned stage_bytes, int h) {
    if constexpr (ABL & 1) { if (in_loop) return; }
    const unsigned sb = wave_lds + stage_bytes + (unsigned)(BM * CPRW) * 16u;
    glds16_m0(wptr + (2 * h) * wpass, sb + (2 * h) * (NT * 16u)); glds16_m0(wptr + (2 * h + 1) * wpass, sb + (2 * h + 1) * (NT * 16u));
  };
  auto step_w = [&]() { if (wt + 1 < nkt) { wptr += BK * sizeof(T); ++wt; } };
  auto step_p = [&]() { if (pt + 1 < nkt) { advance_p(); ++pt; } };
  auto first_pieces = [&]() {                          // K tile 0 complete + the weights of K tile 1: nothing lands in stage 1's pixel half
    issue_p(0, 0); issue_p(0, 1); step_p();
    issue_w(0, 0); issue_w(0, 1); step_w();
    issue_w(SB, 0); issue_w(SB, 1); step_w();
  };

  // ---- fragment addresses -------------------------------------------------------------------------------------------------
  // MM 0: row = 16-aligned base + (lane & 15), k chunk (of 8 halfs) = 4 kh + (lane >> 4): two per-lane addresses per operand.
  // MM 1: row = 32-aligned base + (lane & 31), k chunk = 2 s + (lane >> 5) for k step s: four per-lane addresses per operand.
  // The chunk swizzle (row >> 1) & 7 depends on the lane only in both.
  constexpr int NA = MM ? 4 : 2;
  const char* ldsb = reinterpret_cast<const char*>(lds);
  const char* pb[NA]; const char* wbp[NA];
  {
    const int fr = MM ? (lane & 31) : (lane & 15), fh = MM ? (lane >> 5) : (lane >> 4), sw = (fr >> 1) & 7;
#pragma unroll
    for (int s = 0; s < NA; ++s) {
      const int ch = MM ? (2 * s + fh) : (4 * s + fh);
      pb[s] = ldsb + (grp * 128 + fr) * 128 + ((ch ^ sw) * 16);
      wbp[s] = ldsb + (BM + wq * 64 + fr) * 128 + ((ch ^ sw) * 16);
    }
  }
  uint4 pa[8], wb[4], wb1[4];
  // accumulators: MM 0 acc[j][i] = 16 channels (block j of 4) x 16 pixels (block i of 8); MM 1 acc32[ws][pbk] = 32 channels x 32 pixels
  f32x4 acc[MM ? 1 : 4][MM ? 1 : 8];
  f32x16 acc32[MM ? 2 : 1][MM ? 4 : 1];
  auto zero_acc = [&]() {
    if constexpr (MM == 0) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    } else {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc32[j][i][e] = 0.f;
    }
  };
  auto read_p = [&](unsigned so, int ps) {             // pixel subtile ps: 64 pixels x K 64
    if constexpr (ABL & 2) { if (in_loop) return; }
    if constexpr (MM == 0) {
#pragma unroll
      for (int kh = 0; kh < 2; ++kh)
#pragma unroll
        for (int i = 0; i < 4; ++i) pa[kh * 4 + i] = *reinterpret_cast<const uint4*>(pb[kh] + so + (ps * 64 + i * 16) * 128);
    } else {
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < 2; ++i) pa[s * 2 + i] = *reinterpret_cast<const uint4*>(pb[s] + so + (ps * 64 + i * 32) * 128);
    }
  };
  auto read_w = [&](unsigned so, int ws, uint4 (&dst)[4]) {   // weight subtile ws: 32 channels x K 64
    if constexpr (ABL & 2) { if (in_loop) return; }
    if constexpr (MM == 0) {
#pragma unroll
      for (int kh = 0; kh < 2; ++kh)
#pragma unroll
        for (int j = 0; j < 2; ++j) dst[kh * 2 + j] = *reinterpret_cast<const uint4*>(wbp[kh] + so + (ws * 32 + j * 16) * 128);
    } else {
#pragma unroll
      for (int s = 0; s < 4; ++s) dst[s] = *reinterpret_cast<const uint4*>(wbp[s] + so + (ws * 32) * 128);
    }
  };
  auto mma_w = [&](int ps, int ws, const uint4 (&wv)[4]) {
    if constexpr (ABL & 4) {                           // keep the fragments alive (the reads must not be dead code), issue nothing
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("" :: "v"(pa[i].x), "v"(pa[i].y), "v"(pa[i].z), "v"(pa[i].w));
#pragma unroll
      for (int i = 0; i < 4; ++i) asm volatile("" :: "v"(wv[i].x), "v"(wv[i].y), "v"(wv[i].z), "v"(wv[i].w));
      return;
    }
    __builtin_amdgcn_s_setprio(1);
    if constexpr (MM == 0) {
#pragma unroll
      for (int kh = 0; kh < 2; ++kh)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int i = 0; i < 4; ++i) Mma<T>::run(wv[kh * 2 + j], pa[kh * 4 + i], acc[ws * 2 + j][ps * 4 + i]);
    } else {
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < 2; ++i) Mma32<T>::run(wv[s], pa[s * 2 + i], acc32[ws][ps * 2 + i]);
    }
    __builtin_amdgcn_s_setprio(0);
  };
  auto sync = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (!(ABL & 8)) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---- epilogue: the wave's 128 pixels x 64 channels through its own 4 KB of stage 1's pixel half ---------------------------
  char* stg = reinterpret_cast<char*>(lds) + SB + wave * 4096;
  // the value this lane holds for (round, slot): 4 consecutive channels of one pixel
  //   MM 0, 16-bit round r (pixel blocks 2r, 2r+1):  slot (ib, j) -> acc[j][2r + ib], pixel ib*16 + (lane & 15), channels j*16 + (lane >> 4)*4
  //   MM 1, 16-bit round r (pixel block r):          slot (ws, q) -> acc32[ws][r][4q..4q+3], pixel lane & 31, channels ws*32 + 8q + (lane >> 5)*4
  // `lane` is passed through an empty asm first: everything derived from it below would otherwise be hoisted out of the tile loop,
  // stay live through the K loop (which has no register to spare) and be spilled to scratch - and a scratch reload in the epilogue
  // waits, in order, for the next tile's DMA pieces issued just before it.
  auto epilogue16 = [&](auto act_tag, int em0, int en0, const float4 (&bq)[8], bool full) {
    constexpr int ACT = decltype(act_tag)::value;
    int lane = tid & 63;
    asm volatile("" : "+v"(lane));
    const int nb = en0 + wq * 64;
    T* outp = reinterpret_cast<T*>(p.out) + p.out_coff + nb;
    // store side (same for both MM): lane -> pixel t*8 + (lane >> 3) of the round, 16-byte chunk lane & 7 of its 128-byte row
    const int spix = lane >> 3, schunk = lane & 7;
    const char* rdp = stg + spix * 128 + ((schunk ^ (spix & 7)) * 16);
    if constexpr (MM == 0) {
      const int fr = lane & 15, fg = lane >> 4;
      float4 b4[4], s4[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        b4[j] = bq[j];
        if constexpr (ACT == 3) s4[j] = *reinterpret_cast<const float4*>(p.slope + nb + j * 16 + fg * 4); else s4[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int ib = 0; ib < 2; ++ib)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const f32x4 av = acc[j][2 * r + ib];
            float v0 = act_bias<T, ACT>(osc, av[0], b4[j].x, s4[j].x), v1 = act_bias<T, ACT>(osc, av[1], b4[j].y, s4[j].y);
            float v2 = act_bias<T, ACT>(osc, av[2], b4[j].z, s4[j].z), v3 = act_bias<T, ACT>(osc, av[3], b4[j].w, s4[j].w);
            if constexpr (ACT == 4) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
            const int row = ib * 16 + fr, cl = j * 2 + (fg >> 1);
            *reinterpret_cast<uint2*>(stg + row * 128 + ((cl ^ (row & 7)) * 16) + (fg & 1) * 8) = make_uint2(pack2<T>(v0, v1), pack2<T>(v2, v3));
          }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const uint4 v = *reinterpret_cast<const uint4*>(rdp + t * 1024);
          const int m = em0 + grp * 128 + r * 32 + t * 8 + spix;
          if constexpr (ABL & 32) asm volatile("" :: "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
          else if (full || m < M) *reinterpret_cast<uint4*>(outp + (size_t)m * p.out_cstride + schunk * 8) = v;
        }
        asm volatile("" ::: "memory");
      }
    } else {
      const int fr = lane & 31, fh = lane >> 5;
      float4 b4[2][4], s4[2][4];
#pragma unroll
      for (int ws = 0; ws < 2; ++ws)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = nb + ws * 32 + q * 8 + fh * 4;
          b4[ws][q] = bq[ws * 4 + q];
          if constexpr (ACT == 3) s4[ws][q] = *reinterpret_cast<const float4*>(p.slope + n); else s4[ws][q] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int ws = 0; ws < 2; ++ws)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x16 av = acc32[ws][r];
            float v0 = act_bias<T, ACT>(osc, av[4 * q + 0], b4[ws][q].x, s4[ws][q].x), v1 = act_bias<T, ACT>(osc, av[4 * q + 1], b4[ws][q].y, s4[ws][q].y);
            float v2 = act_bias<T, ACT>(osc, av[4 * q + 2], b4[ws][q].z, s4[ws][q].z), v3 = act_bias<T, ACT>(osc, av[4 * q + 3], b4[ws][q].w, s4[ws][q].w);
            if constexpr (ACT == 4) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
            const int cl = ws * 4 + q;
            *reinterpret_cast<uint2*>(stg + fr * 128 + ((cl ^ (fr & 7)) * 16) + fh * 8) = make_uint2(pack2<T>(v0, v1), pack2<T>(v2, v3));
          }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const uint4 v = *reinterpret_cast<const uint4*>(rdp + t * 1024);
          const int m = em0 + grp * 128 + r * 32 + t * 8 + spix;
          if constexpr (ABL & 32) asm volatile("" :: "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
          else if (full || m < M) *reinterpret_cast<uint4*>(outp + (size_t)m * p.out_cstride + schunk * 8) = v;
        }
        asm volatile("" ::: "memory");
      }
    }
  };
  // f32 values through the same 4 KB, 16 pixels x 64 channels (256-byte rows, 16 chunks of 16 bytes) per round: residual add and
  // store (f32 or storage type) as whole rows.  MM 0: round r = pixel block r; MM 1: round 2r + half = pixels half*16.. of block r,
  // written by the lanes that hold them (the other half of the wave is masked off for the LDS writes).  The residual rows of
  // round k+1 are requested BEFORE the stores of round k are issued: vmcnt retires in issue order, so a load queued behind the
  // previous round's stores would wait for their acknowledgement every round.
  auto epilogue32 = [&](auto act_tag, int em0, int en0, const float4 (&bq)[8], bool full) {
    constexpr int ACT = decltype(act_tag)::value;
    int lane = tid & 63;
    asm volatile("" : "+v"(lane));
    const int nb = en0 + wq * 64;
    const int spix = lane >> 4, schunk = lane & 15;    // store side: 4 pixels per pass, 16 lanes x 16 bytes = one 256-byte row
    const int n = nb + schunk * 4;
    const int mbase = em0 + grp * 128 + spix;          // + 16 * round + 4 * t
    auto load_res = [&](int k, float (&rv)[4][4]) {
      if (!p.res) return;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int m = mbase + k * 16 + t * 4;
        if (full || m < M) {
          const size_t ri = (size_t)m * p.res_cstride + p.res_coff + n;
          if (p.res_f32) load4<float>(p.res, ri, rv[t]); else load4<T>(p.res, ri, rv[t]);
        }
      }
    };
    auto finish = [&](int k, const float (&rv)[4][4]) {
      asm volatile("" ::: "memory");
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int pr = t * 4 + spix;
        const float4 v = *reinterpret_cast<const float4*>(stg + pr * 256 + ((schunk ^ (pr & 15)) * 16));
        const int m = mbase + k * 16 + t * 4;
        if (full || m < M) {
          float o[4] = {v.x, v.y, v.z, v.w};
          if (p.res) {
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = rv[t][e] + o[e];
          }
          if constexpr (ACT == 4) {
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = fmaxf(o[e], 0.f);
          }
          const size_t oi = (size_t)m * p.out_cstride + p.out_coff + n;
          if (p.out_f32) store4<float>(p.out, oi, o); else store4<T>(p.out, oi, o);
        }
      }
      asm volatile("" ::: "memory");
    };
    float4 b4[8], s4[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      b4[k] = bq[k];
      s4[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      if constexpr (ACT == 3) {
        if (MM == 0) { if (k < 4) s4[k] = *reinterpret_cast<const float4*>(p.slope + nb + k * 16 + (lane >> 4) * 4); }
        else s4[k] = *reinterpret_cast<const float4*>(p.slope + nb + (k >> 2) * 32 + (k & 3) * 8 + (lane >> 5) * 4);
      }
    }
    auto stage = [&](int k) {                          // round k of 8: this lane's activated values -> the wave's 4 KB
      if constexpr (MM == 0) {
        const int fr = lane & 15, fg = lane >> 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const f32x4 av = acc[j][k];
          const int cl = j * 4 + fg;
          *reinterpret_cast<float4*>(stg + fr * 256 + ((cl ^ (fr & 15)) * 16)) =
              make_float4(act_bias<T, ACT>(osc, av[0], b4[j].x, s4[j].x), act_bias<T, ACT>(osc, av[1], b4[j].y, s4[j].y),
                          act_bias<T, ACT>(osc, av[2], b4[j].z, s4[j].z), act_bias<T, ACT>(osc, av[3], b4[j].w, s4[j].w));
        }
      } else {
        const int fr = lane & 31, fh = lane >> 5, r = k >> 1, half = k & 1;
        if ((fr >> 4) == half) {
          const int row = fr & 15;
#pragma unroll
          for (int ws = 0; ws < 2; ++ws)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const f32x16 av = acc32[ws][r];
              const float4 b = b4[ws * 4 + q], sl = s4[ws * 4 + q];
              const int cl = ws * 8 + q * 2 + fh;
              *reinterpret_cast<float4*>(stg + row * 256 + ((cl ^ (row & 15)) * 16)) =
                  make_float4(act_bias<T, ACT>(osc, av[4 * q + 0], b.x, sl.x), act_bias<T, ACT>(osc, av[4 * q + 1], b.y, sl.y),
                              act_bias<T, ACT>(osc, av[4 * q + 2], b.z, sl.z), act_bias<T, ACT>(osc, av[4 * q + 3], b.w, sl.w));
            }
        }
      }
    };
    float ra[4][4] = {}, rb[4][4] = {};
    load_res(0, ra);
#pragma unroll
    for (int k = 0; k < 8; k += 2) {                   // two rounds per trip: the residual buffers alternate by name, not by index
      stage(k);
      load_res(k + 1, rb);
      finish(k, ra);
      stage(k + 1);
      if (k + 2 < 8) load_res(k + 2, ra);
      finish(k + 1, rb);
    }
  };
  const bool staged16 = !p.res && !p.out_f32 && (p.out_coff % 8 == 0) && (p.out_cstride % 8 == 0);
  const bool rows32 = (p.out_f32 ? (p.out_coff % 4 == 0 && p.out_cstride % 4 == 0) : (p.out_coff % 4 == 0 && p.out_cstride % 4 == 0)) &&
                      (!p.res || (p.res_coff % 4 == 0 && p.res_cstride % 4 == 0));
  (void)rows32;                                        // launch_persist only takes layers whose views are 4-element aligned
  // the lane's bias values for the tile (4 consecutive channels per slot): loaded BEFORE the next tile's DMA pieces are issued,
  // because vmcnt retires in order - a bias load issued behind the pieces would be waited for together with them
  auto load_bias = [&](int en0, float4 (&bq)[8]) {
    int lane = tid & 63;
    asm volatile("" : "+v"(lane));
    const int nb = en0 + wq * 64;
#pragma unroll
    for (int k = 0; k < 8; ++k) bq[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias) {
      if constexpr (MM == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) bq[j] = *reinterpret_cast<const float4*>(p.bias + nb + j * 16 + (lane >> 4) * 4);
      } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) bq[k] = *reinterpret_cast<const float4*>(p.bias + nb + (k >> 2) * 32 + (k & 3) * 8 + (lane >> 5) * 4);
      }
    }
  };
  auto epilogue = [&](int em0, int en0, const float4 (&bq)[8], bool full) {
    if (staged16) {
      if (p.act == 1) epilogue16(std::integral_constant<int, 1>{}, em0, en0, bq, full);
      else if (p.act == 2) epilogue16(std::integral_constant<int, 2>{}, em0, en0, bq, full);
      else if (p.act == 3) epilogue16(std::integral_constant<int, 3>{}, em0, en0, bq, full);
      else if (p.act == 4) epilogue16(std::integral_constant<int, 4>{}, em0, en0, bq, full);
      else epilogue16(std::integral_constant<int, 0>{}, em0, en0, bq, full);
    } else {
      if (p.act == 1) epilogue32(std::integral_constant<int, 1>{}, em0, en0, bq, full);
      else if (p.act == 2) epilogue32(std::integral_constant<int, 2>{}, em0, en0, bq, full);
      else if (p.act == 3) epilogue32(std::integral_constant<int, 3>{}, em0, en0, bq, full);
      else if (p.act == 4) epilogue32(std::integral_constant<int, 4>{}, em0, en0, bq, full);
      else epilogue32(std::integral_constant<int, 0>{}, em0, en0, bq, full);
    }
  };

  // The loop top waits for EVERYTHING this wave has in flight (vmcnt(0)): the next tile's first pieces and the previous tile's
  // stores.  Waiting for the pieces only (a counted vmcnt(16) behind exactly 16 store instructions, resting on vector-memory
  // loads and stores retiring in issue order) was built and measured: no difference (profiles/r03d_persist_ab.txt) - the stall
  // is at store ISSUE inside the epilogue, not at their acknowledgement - so the plain, obviously safe wait stays.
  int vb = blockIdx.x;
  setup_tile(vb);
  first_pieces();
  for (; vb < ntiles; vb += gridDim.x) {
    zero_acc();
    wait_vmcnt<0>();                                   // this tile's first pieces have landed, the previous tile's stores are out
    __builtin_amdgcn_s_barrier();                      // ... for every wave; every wave's epilogue (its LDS rounds) is behind it
    if (grp == 1) __builtin_amdgcn_s_barrier();        // group 1 runs one barrier behind group 0 through the K loop
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (ABL & 2) { read_w(0, 0, wb); read_w(0, 1, wb1); read_p(0, 0); }
    in_loop = true;
    for (int t = 0; t < nkt; ++t) {                    // schedule 1 of conv_phase_kernel
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
    in_loop = false;
    wait_vmcnt<0>();                                   // this wave's surplus pieces have landed: nothing of it is in flight towards the stages
    if (grp == 0) __builtin_amdgcn_s_barrier();        // everybody is out of the K loop: the stages are free
    __builtin_amdgcn_sched_barrier(0);
    const int em0 = m0, en0 = n0;
    float4 bq[8];
    load_bias(en0, bq);
    __builtin_amdgcn_sched_barrier(0);
    if (vb + (int)gridDim.x < ntiles) { setup_tile(vb + gridDim.x); if constexpr (!(ABL & 64)) first_pieces(); }   // next tile's DMA runs under the epilogue
    __builtin_amdgcn_sched_barrier(0);
    const bool full = em0 + BM <= M;
    if constexpr (!(ABL & 16)) epilogue(em0, en0, bq, full);
    else {                                             // every accumulator stays live: the MFMAs must not become dead code
      if constexpr (MM == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int i = 0; i < 8; ++i) asm volatile("" :: "v"(acc[j][i]));
      } else {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int i = 0; i < 4; ++i) asm volatile("" :: "v"(acc32[j][i]));
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// Eligibility beyond conv_phase_kernel's: Cout a multiple of 256 (whole channel tiles: the epilogue has no channel guard) and
// 4-element aligned views for the row-wise f32 / residual path.
bool conv_persist_ok(const ConvP& p) {
  if (p.Cout % 256) return false;
  if (p.res || p.out_f32) {
    if (p.out_coff % 4 || p.out_cstride % 4) return false;
    if (p.res && (p.res_coff % 4 || p.res_cstride % 4)) return false;
  } else if (p.out_coff % 8 || p.out_cstride % 8) return false;
  return true;
}

template <class T, int MM, int ABL = 0> static void launch_persist_t(const ConvP& p, const ConvAux& a, int M, hipStream_t stream) {
  constexpr size_t lds = (size_t)2 * 512 * 8 * 16 + 512 * 4 * 16;     // two K tiles of (256 + 256) rows x 128 bytes + the per-row loader table
  static PerDevice pd;                                 // attribute and CU count per device ordinal
  const int d = pd.index();
  if (pd.first(d))
    CC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_persist_kernel<T, MM, ABL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int cus = pd.cu_count(d);
  ConvAux b = a;
  b.ntiles = ((M + 255) / 256) * a.nt;
  note_launch("conv_persist", conv_persist_kernel<T, MM, ABL>, (long)b.ntiles, 512, lds, std::min(b.ntiles, cus));
  hipLaunchKernelGGL((conv_persist_kernel<T, MM, ABL>), dim3(std::min(b.ntiles, cus)), dim3(512), lds, stream, p, b);
}

// abl: timing ablations (development build only: -DCC_PERSIST_ABLATIONS; bf16, 16x16x32; results are wrong with any of them)
void launch_conv_persist(int dt, int mm, int abl, const ConvP& p, const ConvAux& a, int M, hipStream_t stream) {
#ifdef CC_PERSIST_ABLATIONS
  if (abl && dt == BF16 && !mm) {
    switch (abl) {
      case 1: launch_persist_t<bf16_t, 0, 1>(p, a, M, stream); return;    // no DMA in the loop
      case 2: launch_persist_t<bf16_t, 0, 3>(p, a, M, stream); return;    // no DMA, no fragment reads
      case 3: launch_persist_t<bf16_t, 0, 4>(p, a, M, stream); return;    // no MFMA
      case 4: launch_persist_t<bf16_t, 0, 16>(p, a, M, stream); return;   // no epilogue
      case 5: launch_persist_t<bf16_t, 0, 11>(p, a, M, stream); return;   // MFMA only: no DMA, no reads, no barriers
      case 6: launch_persist_t<bf16_t, 0, 27>(p, a, M, stream); return;   // ... and no epilogue
      case 7: launch_persist_t<bf16_t, 0, 32>(p, a, M, stream); return;   // no global stores in the epilogue
      case 8: launch_persist_t<bf16_t, 0, 64>(p, a, M, stream); return;   // next tile's first pieces not issued
      case 9: launch_persist_t<bf16_t, 0, 96>(p, a, M, stream); return;   // ... and no global stores
      case 10: launch_persist_t<bf16_t, 0, 75>(p, a, M, stream); return;  // MFMA only + epilogue without first pieces
      default: break;
    }
  }
  // v_mfma_f32_32x32x16 (MM = 1): correct and bit-identical end to end, 10-15 % slower on every shape (profiles/r03b): development build only
  if (mm) { if (dt == F16) launch_persist_t<f16_t, 1>(p, a, M, stream); else launch_persist_t<bf16_t, 1>(p, a, M, stream); return; }
#endif
  (void)abl; (void)mm;
  if (dt == F16) launch_persist_t<f16_t, 0>(p, a, M, stream); else launch_persist_t<bf16_t, 0>(p, a, M, stream);
}

}  // namespace cc

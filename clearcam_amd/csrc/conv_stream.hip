// Weights-resident streaming kernel for the thin 1x1 convolutions (round 5): the layers whose roof is HBM, not the matrix pipes
// (detection/yolov9.py:65-125 - cv1 / cv4 of every RepNCSPELAN4, cv1 | cv2 and cv3 of every RepNCSP, ADown's cv2).
//
// The tile kernels (conv_mfma / conv_phase / conv_persist) fetch a weight tile per pixel tile, set a loader up per tile and pay ~11 us of
// fixed cost per 256 x 256 tile against 4-8 K steps: on K <= 512 they run at 2.2-3.0 TB/s, issue- and latency-bound.  Here
//   * ALL weights of the layer live in REGISTERS for the life of the block: wave (wn, wm) holds the rows of its 32 output channels for
//     every K chunk (2 x KT x 16 bytes per lane: 128 registers at K = 512), loaded once from L2;
//   * pixel rows stream HBM -> LDS through a four-slot ring of 32 KB tiles by LDS-DMA (global_load_lds_dwordx4), three tiles ahead of the
//     one being multiplied; the loads are never drained: counted vmcnt over the DMA pieces of later tiles (by default the stores issued since are
//     assumed acknowledged - they are half a tile old - so nothing rests on stores and loads retiring in one order; StreamAux::flags bit 8);
//   * two barriers per tile (two wave groups half a tile apart, below); no weight traffic, no loader tables, no per-tile set-up beyond
//     four pointer increments per lane;
//   * two weight planes (ConvP::split, dtype f16s / f16h): the K walk visits the tile's channels twice against the second half of the weight
//     row - the same LDS image, no second HBM read;
//   * the epilogue stores from registers: the (MFMA row -> output channel) permutation below gives every lane 8 consecutive channels of one
//     pixel (16 bytes) and every store instruction 64 contiguous bytes per pixel.
// Same MFMA instruction, same chunk -> k mapping, same (plane, channel) walk and the same epilogue arithmetic as the tile kernels: the same
// bits (tests/test_gpu_yolo.py::test_stream_1x1_equals_generic).
#include "conv_tile.h"

namespace cc {

struct StreamAux { int ntiles, M, flags; };   // flags (cc_dev_set("stream_flags") / CLEARCAM_STREAM_FLAGS, default 10): 1 = raised priority for the MFMA phase, 2 = for the memory phase, 4 = activation arithmetic in the memory phase (LATE), 8 = waits count loads only
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;   // a register quadruple an asm operand can name (HIP's uint4 is a struct)

// runtime-valued counted wait: N = pieces and stores issued after the DMA pieces that must have landed (multiples of 2 up to 24)
__device__ __forceinline__ void wait_vmcnt_dyn(int n) {
  switch (n) {
    case 0: wait_vmcnt<0>(); break;   case 2: wait_vmcnt<2>(); break;   case 4: wait_vmcnt<4>(); break;   case 6: wait_vmcnt<6>(); break;
    case 8: wait_vmcnt<8>(); break;   case 10: wait_vmcnt<10>(); break; case 12: wait_vmcnt<12>(); break; case 14: wait_vmcnt<14>(); break;
    case 16: wait_vmcnt<16>(); break; case 18: wait_vmcnt<18>(); break; case 20: wait_vmcnt<20>(); break; case 22: wait_vmcnt<22>(); break;
    case 24: wait_vmcnt<24>(); break; default: wait_vmcnt<0>(); break;
  }
}

// WN waves along the output channels (32 each: Cout = 32 WN), 8 / WN along the pixels; KT = K chunks of 32 (Ktot / 32, both planes);
// CINC = input-channel chunks of 32 (Cin / 32; KT = CINC or 2 CINC); NP = 16-pixel MFMA tiles per wave and ring slot.
//
// Two wave groups half a tile period apart (waves w and w + 4 share a SIMD): every wave passes two barriers per tile, P before its compute
// phase (MFMAs, then bias + SiLU + pack in registers) and Q before its memory phase (this tile's stores, the next tile's DMA pieces), and
// group 1 passes one extra barrier before the loop - so global barrier 2t .. 2t+1 has group 0 computing tile t beside group 1 storing
// tile t - 1, and 2t+1 .. 2t+2 the reverse: on every SIMD one wave computes while the other sits in vector-memory issue (both in
// lockstep measured 3.9 TB/s at 256 -> 256, this 4.1; docs/rounds/r05.md has the ablations and what did not help).
//   pieces of tile u land before global barrier 2u: group 0 waits for its own before P_u, group 1 before Q_{u-1};
//   slot of tile u is free after global barrier 2u + 2: group 0 refills it (tile u + S) after Q_{u+1}, group 1 after Q_u.
// ABL (development, timing only - results are WRONG with any bit set): 1 no MFMA, 2 no fragment reads, 4 no activation arithmetic,
// 8 no DMA inside the loop, 16 no stores.
// LATE (A/B only, flags bit 4): the activation arithmetic behind barrier Q, in the memory phase beside the OTHER group's MFMAs.  On paper
// 2 x max(MFMA, VALU + memory issue) per tile instead of 2 x (MFMA + VALU); measured 10-15 % slower in the plan (profiles/r05r_*): a wave
// parked at store issue holds its arithmetic up with it.
template <class T, int WN, int KT, int CINC, int NP, int ABL = 0, int LATE = 0>
__global__ __launch_bounds__(512) void conv_stream_kernel(const ConvP p, const StreamAux a) {
  constexpr int NT = 2, WM = 8 / WN;
  constexpr int PT = WM * NP * 16;                     // pixels per tile
  constexpr int KS = CINC / 2;                         // 64-channel slabs: LDS rows of 128 bytes, chunk-swizzled by (row >> 1) & 7
  constexpr int SLOT = PT * KS * 128;                  // bytes per ring slot
  constexpr int S = 4;                                 // ring slots
  constexpr int RB = PT / 8;                           // 1 KB pieces (8 rows of one slab) per slab
  constexpr int NPIECE = SLOT / 1024 / 8;              // pieces per wave per tile
  constexpr int Q = (ABL & 16) ? 0 : NP;               // output store instructions per wave per tile
  static_assert(sizeof(T) == 2 && (CINC % 2) == 0 && (KT == CINC || KT == 2 * CINC), "16-bit storage, whole 64-channel slabs");
  static_assert(NPIECE * 8 * 1024 == SLOT && NPIECE >= 1, "a tile is a whole number of pieces per wave");
  static_assert((S - 1) * Q + (S - 1) * NPIECE <= 24, "counted wait within wait_vmcnt_dyn's range");
  extern __shared__ __attribute__((aligned(16))) uint4 lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave % WN, wm = wave / WN, grp = wave >> 2;
  const int r = lane & 15, g = lane >> 4;
  const int M = a.M, G = gridDim.x, bx = blockIdx.x;
  const int nmine = (a.ntiles - bx + G - 1) / G;
  const unsigned lds_base = lds_addr(lds);
  const char* X = reinterpret_cast<const char*>(p.s0.ptr) + (size_t)p.s0.coff * sizeof(T);
  const size_t xrow = (size_t)p.s0.cstride * sizeof(T);

  // ---- pixel stream: tile `tl` of this block -> ring slot tl % S ---------------------------------------------------------------
  // piece q = wave + 8 i: slab q / RB, rows (q % RB) * 8 .. + 8; lane -> row + (lane >> 3), LDS position lane & 7 holds source chunk
  // position ^ swizzle(row) (the DMA writes lane-linear, so the swizzle sits on the source side).  Tiles are issued in order: the
  // per-piece source pointers advance by one block stride per tile; only a ragged last tile recomputes them (rows past the end re-read
  // the last pixel and are never stored).
  const char* nptr[NPIECE];
#pragma unroll
  for (int i = 0; i < NPIECE; ++i) {
    const int q = wave + 8 * i, slab = q / RB, rb = q - slab * RB;
    const int row = rb * 8 + (lane >> 3), chunk = (lane & 7) ^ ((row >> 1) & 7);
    nptr[i] = X + (size_t)(bx * PT + row) * xrow + (size_t)(slab * 64 + chunk * 8) * sizeof(T);
  }
  const size_t tstride = (size_t)G * PT * xrow;
  auto issue = [&](int tl) {
    const int m0 = (bx + tl * G) * PT;
    const unsigned sb = lds_base + (unsigned)(tl % S) * SLOT;
    if (m0 + PT <= M) {
#pragma unroll
      for (int i = 0; i < NPIECE; ++i) glds16_m0(nptr[i], __builtin_amdgcn_readfirstlane(sb + (unsigned)(wave + 8 * i) * 1024u));
    } else {
#pragma unroll
      for (int i = 0; i < NPIECE; ++i) {
        const int q = wave + 8 * i, slab = q / RB, rb = q - slab * RB;
        const int row = rb * 8 + (lane >> 3), chunk = (lane & 7) ^ ((row >> 1) & 7);
        int m = m0 + row;
        m = m < M ? m : M - 1;
        glds16_m0(X + (size_t)m * xrow + (size_t)(slab * 64 + chunk * 8) * sizeof(T), __builtin_amdgcn_readfirstlane(sb + (unsigned)q * 1024u));
      }
    }
#pragma unroll
    for (int i = 0; i < NPIECE; ++i) nptr[i] += tstride;
  };
  const int npro = nmine < S ? nmine : S, npro0 = npro < 2 ? npro : 2;
  for (int tl = 0; tl < npro0; ++tl) issue(tl);        // the stream starts before the weights are fetched

  // ---- weights: MFMA row i of channel tile ct <-> output channel cb + (i >> 2) * 8 + ct * 4 + (i & 3) -----------------------------
  // (the A operand's row is lane & 15; the D rows a lane holds are 4 g .. 4 g + 3 of both tiles = channels cb + 8 g .. + 8)
  const int cb = wn * 32;
  u32x4 wreg[NT][KT];
  {
    const T* W = reinterpret_cast<const T*>(p.w);
#pragma unroll
    for (int ct = 0; ct < NT; ++ct) {
      const int co = cb + (r >> 2) * 8 + ct * 4 + (r & 3);
#pragma unroll
      for (int kc = 0; kc < KT; ++kc) wreg[ct][kc] = *reinterpret_cast<const u32x4*>(W + (size_t)co * p.Kw + kc * 32 + g * 8);
    }
  }
  float bv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (p.bias) {
    const float4 b0 = *reinterpret_cast<const float4*>(p.bias + cb + g * 8), b1 = *reinterpret_cast<const float4*>(p.bias + cb + g * 8 + 4);
    bv[0] = b0.x; bv[1] = b0.y; bv[2] = b0.z; bv[3] = b0.w; bv[4] = b1.x; bv[5] = b1.y; bv[6] = b1.z; bv[7] = b1.w;
  }
  // The compiler waits for its own loads with a count that knows nothing of the DMA pieces: whatever is issued before that wait has to
  // land with them.  So the weights are made "consumed" here (the wait lands here, with two tiles in flight), and only then does the rest
  // of the ring fill.
#pragma unroll
  for (int ct = 0; ct < NT; ++ct)
#pragma unroll
    for (int kc = 0; kc < KT; ++kc) asm volatile("" : "+v"(wreg[ct][kc]));
#pragma unroll
  for (int e = 0; e < 8; ++e) asm volatile("" : "+v"(bv[e]));
  for (int tl = npro0; tl < npro; ++tl) issue(tl);

  const float osc = out_scale(p);
  // fragment addresses inside a slab: pixel row = wave's first row + 16 pt + r, k chunk (of 8 halfs) 4 h + g of the 128-byte row
  const int sw = (r >> 1) & 7;
  const unsigned off0 = (unsigned)(wm * NP * 16 + r) * 128u + (unsigned)((g ^ sw) * 16), off1 = (unsigned)(wm * NP * 16 + r) * 128u + (unsigned)(((4 + g) ^ sw) * 16);
  T* optr = reinterpret_cast<T*>(p.out) + p.out_coff + cb + g * 8 + (size_t)(bx * PT + wm * NP * 16 + r) * p.out_cstride;
  const size_t ostride = (size_t)G * PT * p.out_cstride, opt = (size_t)16 * p.out_cstride;

  // ops this wave issued after the pieces of the tile it now needs: d later tiles' pieces, the stores of st tiles (both retire in issue order).
  // Per iteration a wave issues stores(t) and then the pieces of ONE later tile (group 0: t - 1 + S, group 1: t + S), after the S tiles of the prologue.
  // flags bit 8 (DEFAULT ON): the wait does not rest on stores retiring in issue order with the loads - it waits as if every store issued
  // since had already been acknowledged, so only later tiles' pieces may stay outstanding (loads do return in order).  Counting the stores
  // as well (bit 8 off: allowed on gfx9-class counters, where LLVM itself treats loads and stores as one in-order event class) measured the
  // same: 12.17 vs 12.18 ms per step on the same box - the stores of half a tile ago have long been acknowledged.
  const bool loads_only = (a.flags & 8) != 0;
  auto wait_tile = [&](int d, int st) { wait_vmcnt_dyn(d * NPIECE + (loads_only ? 0 : st * Q)); };
  auto barrier = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  if (grp == 1) {                                      // tile 0's pieces, then the extra barrier that puts the group half a period behind
    wait_tile((nmine - 1) < (S - 1) ? (nmine - 1) : (S - 1), 0);
    barrier();
  }

  for (int t = 0; t < nmine; ++t) {
    if (grp == 0) {
      const int dmax = t == 0 ? S - 1 : S - 2, left = nmine - 1 - t;
      wait_tile(left < dmax ? left : dmax, t < S ? t : S - 2);
    }
    barrier();                                         // P: every wave's pieces of tile t have landed

    const char* sbp = reinterpret_cast<const char*>(lds) + (t % S) * SLOT;
    f32x4 acc[NT][NP];
#pragma unroll
    for (int ct = 0; ct < NT; ++ct)
#pragma unroll
      for (int pt = 0; pt < NP; ++pt) acc[ct][pt] = f32x4{0.f, 0.f, 0.f, 0.f};
    // fragments of K chunk kc + 1 are requested before the MFMAs of chunk kc; the scheduling fence per chunk keeps hipcc from hoisting
    // every ds_read of the unrolled walk to the top (it spilled 116 registers doing so at KT = 16)
    auto frag = [&](int kc, uint4 (&bf)[NP]) {
      if constexpr (ABL & 2) { if (kc > 0) { for (int pt = 0; pt < NP; ++pt) asm volatile("" : "+v"(bf[pt].x)); return; } }
      const int pc = kc % CINC, slab = pc >> 1;
      const unsigned off = (pc & 1) ? off1 : off0;
#pragma unroll
      for (int pt = 0; pt < NP; ++pt) bf[pt] = *reinterpret_cast<const uint4*>(sbp + slab * (PT * 128) + pt * 2048 + off);
    };
    uint4 bfa[NP], bfb[NP];
    frag(0, bfa);
    if (a.flags & 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kc = 0; kc < KT; kc += 2) {
      frag(kc + 1, bfb);
#pragma unroll
      for (int ct = 0; ct < NT; ++ct)
#pragma unroll
        for (int pt = 0; pt < NP; ++pt) { if constexpr (ABL & 1) asm volatile("" : "+v"(acc[ct][pt]) : "v"(wreg[ct][kc]), "v"(bfa[pt].x)); else Mma<T>::run(__builtin_bit_cast(uint4, wreg[ct][kc]), bfa[pt], acc[ct][pt]); }
      __builtin_amdgcn_sched_group_barrier(0x100, NP, 0);        // the next chunk's ds_reads first, then this chunk's MFMAs
      __builtin_amdgcn_sched_group_barrier(0x008, NT * NP, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (kc + 2 < KT) frag(kc + 2, bfa);
#pragma unroll
      for (int ct = 0; ct < NT; ++ct)
#pragma unroll
        for (int pt = 0; pt < NP; ++pt) { if constexpr (ABL & 1) asm volatile("" : "+v"(acc[ct][pt]) : "v"(wreg[ct][kc + 1]), "v"(bfb[pt].x)); else Mma<T>::run(__builtin_bit_cast(uint4, wreg[ct][kc + 1]), bfb[pt], acc[ct][pt]); }
      __builtin_amdgcn_sched_group_barrier(0x100, NP, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, NT * NP, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (a.flags & 1) __builtin_amdgcn_s_setprio(0);
    // bias + activation (compile-time branch per tile) in the SAME phase as the MFMAs: 8 consecutive channels of one pixel per lane, packed.
    // (With the arithmetic behind barrier Q the kernel ran at memory time + 2 x epilogue: a wave's stores queue behind the ring's loads,
    //  the wave is parked at store issue for as long as the memory takes, and whatever else it has to do waits with it - profiles/r05c.)
    uint4 ov[NP];
    auto finish = [&](auto act_tag) {
      constexpr int ACT = (ABL & 4) ? 0 : decltype(act_tag)::value;
#pragma unroll
      for (int pt = 0; pt < NP; ++pt) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] = activate<T, ACT>(__builtin_fmaf(acc[0][pt][e], osc, bv[e]));
          v[4 + e] = activate<T, ACT>(__builtin_fmaf(acc[1][pt][e], osc, bv[4 + e]));
        }
        ov[pt] = make_uint4(pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3]), pack2<T>(v[4], v[5]), pack2<T>(v[6], v[7]));
      }
    };
    auto finish_rt = [&]() {
      if (p.act == 1) finish(std::integral_constant<int, 1>{});
      else if (p.act == 2) finish(std::integral_constant<int, 2>{});
      else finish(std::integral_constant<int, 0>{});
    };
    if constexpr (!LATE) finish_rt();
    if (grp == 1 && t + 1 < nmine) {                   // tile t + 1's pieces: due before global barrier 2 (t + 1), this group's Q_t
      const int left = nmine - 2 - t;
      wait_tile(left < S - 2 ? left : S - 2, t < S - 2 ? t : S - 2);
    }
    barrier();                                         // Q: the other group is done reading tile t - 1 (group 0) / tile t (group 1)

    // memory phase: this tile's rows - one 16-byte store per pixel tile - and then the next tile of the stream into the slot that has just
    // been freed.  Stores FIRST: the vector-memory queue is in order, and behind a freshly issued tile of loads a store waits (and parks its
    // wave, and with it the barrier) until the loads ahead of it have been sent off.
    if (a.flags & 2) __builtin_amdgcn_s_setprio(2);
    if constexpr (LATE) finish_rt();
    const int mb = (bx + t * G) * PT + wm * NP * 16 + r;
#pragma unroll
    for (int pt = 0; pt < NP; ++pt) {
      if constexpr (ABL & 16) asm volatile("" :: "v"(ov[pt].x), "v"(ov[pt].y), "v"(ov[pt].z), "v"(ov[pt].w));
      else if (mb + pt * 16 < M) *reinterpret_cast<uint4*>(optr + pt * opt) = ov[pt];
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    {
      const int u = grp ? t + S : t - 1 + S;
      if constexpr (!(ABL & 8)) { if (u >= S && u < nmine) issue(u); }
    }
    if (a.flags & 2) __builtin_amdgcn_s_setprio(0);
    optr += ostride;
    asm volatile("" ::: "memory");
  }
  if (grp == 0) barrier();                             // pairs with group 1's last Q
}

// ---- host side -------------------------------------------------------------------------------------------------------------------
struct StreamCfg { int wn, kt, cinc, np; };
static bool stream_cfg(const ConvP& p, StreamCfg& c) {
  if (p.ks != 1 || p.stride != 1 || p.pad != 0 || p.s1.C != 0 || p.s0.shift != 0 || p.Hin != p.Ho || p.Win != p.Wo) return false;
  if (p.res || p.out_f32 || p.act > 2 || p.slope) return false;
  if (p.Cin % 64 || p.s0.C != p.Cin || p.s0.cstride % 8 || p.s0.coff % 8 || p.out_cstride % 8 || p.out_coff % 8) return false;
  if (p.Ktot != p.Cin * (1 + (p.split ? 1 : 0)) || p.Kw < p.Ktot) return false;
  if ((long)p.B * p.Ho * p.Wo >= (1L << 30)) return false;
  // the kernel moves 16 bytes per lane on every path (LDS-DMA pieces, weight rows, float4 bias, uint4 stores)
  if ((((uintptr_t)p.s0.ptr | (uintptr_t)p.w | (uintptr_t)p.out) & 15) || (p.bias && ((uintptr_t)p.bias & 15)) || p.Kw % 8) return false;
  c.cinc = p.Cin / 32; c.kt = p.Ktot / 32;
  if (p.Cout == 256) c.wn = 8; else if (p.Cout == 128) c.wn = 4; else if (p.Cout == 64) c.wn = 2; else return false;
  // the instantiated shapes: (Cout, Cin, planes)
  const int key = p.Cout * 10000 + p.Cin * 10 + (p.split ? 2 : 1);
  switch (key) {
    case 256 * 10000 + 256 * 10 + 2: case 256 * 10000 + 256 * 10 + 1: c.np = 4; return true;
    case 256 * 10000 + 512 * 10 + 1: c.np = 2; return true;
    case 128 * 10000 + 128 * 10 + 2: case 128 * 10000 + 128 * 10 + 1: c.np = 4; return true;
    case 64 * 10000 + 64 * 10 + 2: case 64 * 10000 + 64 * 10 + 1: c.np = 4; return true;
    default: return false;
  }
}
bool conv_stream_legal(const ConvP& p) { StreamCfg c; return stream_cfg(p, c); }

// cc_dev_set("stream_flags", bits): StreamAux::flags.  CLEARCAM_STREAM_FLAGS is read HERE, once, when the library is loaded, so that a later
// cc_dev_set wins (a lazy read inside the launcher overwrote an earlier cc_dev_set on the first launch)
int g_stream_flags = [] { const char* e = getenv("CLEARCAM_STREAM_FLAGS"); return e ? atoi(e) : 10; }();
int g_stream_abl = 0;                                  // cc_dev_set("stream_abl", bits): timing ablations of the f16 256 -> 256 shapes (development)
template <class T, int WN, int KT, int CINC, int NP, int ABL = 0, int LATE = 0> static void launch_stream_k(const ConvP& p, hipStream_t stream) {
  constexpr int PT = (8 / WN) * NP * 16;
  constexpr size_t lds = (size_t)4 * PT * (CINC / 2) * 128;
  static PerDevice pd;
  const int d = pd.index();
  if (pd.first(d))
    CC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_stream_kernel<T, WN, KT, CINC, NP, ABL, LATE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int cus = pd.cu_count(d);
  StreamAux a{};
  a.M = p.B * p.Ho * p.Wo;
  a.ntiles = (a.M + PT - 1) / PT;
  a.flags = g_stream_flags;
  note_launch("conv_stream", conv_stream_kernel<T, WN, KT, CINC, NP, ABL, LATE>, (long)a.ntiles, 512, lds, std::min(a.ntiles, cus));
  hipLaunchKernelGGL((conv_stream_kernel<T, WN, KT, CINC, NP, ABL, LATE>), dim3(std::min(a.ntiles, cus)), dim3(512), lds, stream, p, a);
}

template <class T> static void launch_stream_t(const ConvP& p, const StreamCfg& c, hipStream_t stream) {
  const int pl = p.split ? 2 : 1;
#ifdef CC_STREAM_ABLATIONS                              // development build only (HIPCC_EXTRA=-DCC_STREAM_ABLATIONS python clearcam_amd/build.py --force; tools/dev/stream_ablate.py)
  if constexpr (std::is_same<T, f16_t>::value) {
    if (g_stream_abl && p.Cout == 256 && p.Cin == 256) {
#define CC_ABL_CASE(b) case b: if (pl == 2) launch_stream_k<T, 8, 16, 8, 4, b>(p, stream); else launch_stream_k<T, 8, 8, 8, 4, b>(p, stream); return;
      switch (g_stream_abl) { CC_ABL_CASE(1) CC_ABL_CASE(2) CC_ABL_CASE(3) CC_ABL_CASE(4) CC_ABL_CASE(8) CC_ABL_CASE(16) CC_ABL_CASE(20) CC_ABL_CASE(23) CC_ABL_CASE(24) CC_ABL_CASE(7) default: break; }
#undef CC_ABL_CASE
    }
  }
#else
  CC_CHECK(!g_stream_abl, "timing ablations of the streaming kernel need a development build (-DCC_STREAM_ABLATIONS)");
#endif
  // flags bit 4: the activation arithmetic in the memory phase (LATE) - A/B per shape (cc_dev_set("stream_flags"))
  const bool late = (g_stream_flags & 4) != 0;
  if (p.Cout == 256 && p.Cin == 256) {
    if (pl == 2) { if (late) launch_stream_k<T, 8, 16, 8, 4, 0, 1>(p, stream); else launch_stream_k<T, 8, 16, 8, 4>(p, stream); }
    else { if (late) launch_stream_k<T, 8, 8, 8, 4, 0, 1>(p, stream); else launch_stream_k<T, 8, 8, 8, 4>(p, stream); }
  }
  else if (p.Cout == 256 && p.Cin == 512) { if (late) launch_stream_k<T, 8, 16, 16, 2, 0, 1>(p, stream); else launch_stream_k<T, 8, 16, 16, 2>(p, stream); }
  else if (p.Cout == 128) {
    if (pl == 2) { if (late) launch_stream_k<T, 4, 8, 4, 4, 0, 1>(p, stream); else launch_stream_k<T, 4, 8, 4, 4>(p, stream); }
    else { if (late) launch_stream_k<T, 4, 4, 4, 4, 0, 1>(p, stream); else launch_stream_k<T, 4, 4, 4, 4>(p, stream); }
  }
  else {
    if (pl == 2) { if (late) launch_stream_k<T, 2, 4, 2, 4, 0, 1>(p, stream); else launch_stream_k<T, 2, 4, 2, 4>(p, stream); }
    else { if (late) launch_stream_k<T, 2, 2, 2, 4, 0, 1>(p, stream); else launch_stream_k<T, 2, 2, 2, 4>(p, stream); }
  }
  (void)c;
}

void launch_conv_stream(int dt, const ConvP& p, hipStream_t stream) {
  StreamCfg c;
  CC_CHECK(dt != F32 && stream_cfg(p, c), "streaming 1x1: shape not eligible");
  if (dt == F16) launch_stream_t<f16_t>(p, c, stream); else launch_stream_t<bf16_t>(p, c, stream);
  CC_HIP(hipGetLastError());
}

}  // namespace cc

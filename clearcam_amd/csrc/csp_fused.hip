// RepNCSP (detection/yolov9.py:92-105) as ONE kernel for the narrow levels of the detector (hidden width 32 or 64):
//
//     a = SiLU(cv1 x)            1x1, 2*HID -> HID          b = SiLU(cv2 x)      1x1, 2*HID -> HID
//     t = SiLU(rep3x3 a)         RepConvN folded to one 3x3 (yolov9.py:82-90), HID -> HID
//     u = a + SiLU(conv3x3 t)    RepNBottleneck's second conv + shortcut
//     out = SiLU(cv3 [u | b])    1x1, 2*HID -> 2*HID
//
// Layer at a time these are four launches that move x, [a|b], t, u and out through HBM (544 channel-pixels per pixel at
// HID = 32 against 128 for x in / out out) and run at 150-550 TFLOP/s because every one of them is a few tiny K steps
// (profiles/r02c_yolo_per_launch.csv, ops 3-6 and 19-22).  Here a block of eight waves owns an 8 x 16-pixel output tile and
// keeps every intermediate in LDS:
//     X   the 12 x 20 input patch (halo 2), DMA'd once, 64-channel slabs of 128-byte rows (chunk-swizzled like every tile)
//     A   cv1 output on the whole patch (zero outside the image: it is the zero padding of the 3x3 that reads it)
//     B   cv2 output on the 8 x 16 inner pixels
//     T   first 3x3 on the 10 x 18 ring (halo 1; zero outside the image), aliasing X
//     U   second 3x3 + a (inner pixels), aliasing X behind T
// The recompute on the halo is 14 % of the block's MACs (cv1 on 240 instead of 128 pixels, the first 3x3 on 180).
// Every intermediate is rounded to the storage type exactly where the layer-at-a-time path stores it and the K order of every
// accumulation is the same, so the result is bit-identical to the four launches it replaces
// (tests/test_gpu_yolo.py::test_fused_csp_equals_unfused).
//
// Work split: stage 1 gives wave w patch-pixel fragments w and w+8 (all 2*HID channels); stage 2 gives (pixel group, channel half);
// from stage 3 on a wave OWNS one 16-pixel output row with all its channels, so stage 4 and the store epilogue (staged through the
// wave's own u / b rows) need no barrier: three barriers per tile (top, 1->2, 2->3) plus, when weights stream, one per chunk.
//
// Two variants of the outer loop:
//   RES = true  (HID 32: 56 KB of weights): PERSISTENT blocks, one per CU; the four weight matrices and the biases are loaded once
//               and stay in LDS, the input patch is double-buffered (the next tile's patch streams in under the current tile and is
//               waited for at the opening of stage 4, so output stores are never waited for).  0.296 ms against 0.394 ms for the
//               four launches at 160x160, batch 64.
//   RES = false (HID 64: 208 KB of weights cannot stay): one tile per block; weights stream from L2 in chunks of one or two
//               64-wide K slabs through a two-slot ring, chunk i+1 issued at the barrier that opens chunk i.  0.225 ms against
//               0.239 ms at 80x80, batch 64; 19 us against ~50 us for a single frame.
// DESIGN.md section 4 has the timing ablations (the block is bound by its lock-step skeleton and its epilogue VALU, not by MFMA).
#include <utility>
#include "conv_tile.h"

namespace cc {

template <int V> using IC = std::integral_constant<int, V>;
template <int... I, class F> __device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) { (f(IC<I>{}), ...); }
template <int N, class F> __device__ __forceinline__ void static_for(F&& f) { static_for_impl(std::make_integer_sequence<int, N>{}, f); }

// SPLIT 1: every weight matrix is two f16 planes (ConvP::split; rows [tap: hi(Cin) | lo(Cin)]): twice the weight slabs, the same x / a / t /
// u / b fragments walked once per plane, in the conv kernels' order (tap, plane, channel) - still bit-identical to the launches it replaces.
// SPLIT 2 (round 5): two planes in the 1x1 convs only (cv1 | cv2 and cv3), one in the two 3x3 convs - what dtype "f16h" gives the backbone's
// RepNCSP blocks, which had to run as four launches while the kernel took one flag for all four convs.
template <int HID, int SPLIT = 0> struct CspGeom {
  static constexpr bool SPLIT1 = SPLIT != 0, SPLIT3 = SPLIT == 1;            // two planes in the 1x1 convs / in the 3x3 convs
  static constexpr int C2 = 2 * HID, NSLAB = C2 / 64;                        // X / cv1|cv2 / cv3 K slabs of 64 channels
  static constexpr int NSLABW = NSLAB * (SPLIT1 ? 2 : 1);                     // weight slabs of cv1|cv2 / cv3: [hi slabs | lo slabs]
  static constexpr int TH = 8, TW = 16, PW = TW + 4, PH = TH + 4, PR = PH * PW;   // 12 x 20 = 240 patch pixels
  static constexpr int QW = TW + 2, QH = TH + 2, QR = QH * QW;               // 10 x 18 = 180 ring pixels
  static constexpr int CPA = HID / 8;                                        // 16-byte chunks per row of A, B, T, U
  // Row pitch of the intermediates: HID channels + 32 bytes of padding.  They are written by ds_write (not by the lane-linear DMA),
  // so their layout is free, and with padded rows every tap of a 3x3 is base + compile-time offset (no XOR swizzle to recompute).
  // The pad is chosen for the READS: a ds_read_b128 is served in lane groups that mix two neighbouring k-groups ({0-3, 12-15, 20-27},
  // ...), and a pitch of 2 (mod 4) sixteen-byte units - 6 at HID 32, 10 at HID 64 - puts sixteen consecutive rows of one k-group on
  // even units and the other's on odd ones.  (HID*2 + 16, an odd pitch, left three two-way conflicts per group: SQ_LDS_BANK_CONFLICT
  // was 6-8 % of the kernel's wave cycles.)
  static constexpr int PA = HID * 2 + 32;
  static_assert((PA / 16) % 4 == 2, "pitch of the intermediates must be 2 (mod 4) sixteen-byte units");
  static constexpr int X_BYTES = NSLAB * PR * 128, A_BYTES = 256 * PA, B_BYTES = 128 * PA, T_BYTES = 192 * PA, U_BYTES = 128 * PA;
  static constexpr int BIAS_BYTES = (2 * C2 + 2 * HID) * 4;                  // b12 | br | bb | b3 as f32
  static constexpr int SPC = HID == 64 ? 2 : 3;                              // streaming: 3x3 weight slabs per chunk
  // slabs / chunks per 3x3 stage: 9 / 5 (HID 64) and 5 / 2 (HID 32); split: a tap is [hi | lo] = two slabs at HID 64 (18 / 9), one at HID 32 (9 / 3)
  static constexpr int NS3 = SPLIT3 ? (HID == 64 ? 18 : 9) : (9 * HID + 63) / 64, NC3 = (NS3 + SPC - 1) / SPC;
  static constexpr int SLAB_BIG = C2 * 128, SLAB_SMALL = HID * 128;          // bytes of a 64-wide K slab of cv1|cv2 / cv3 and of a 3x3
  static constexpr int RING = SLAB_BIG > SPC * SLAB_SMALL ? SLAB_BIG : SPC * SLAB_SMALL;
  static constexpr int NCHUNK = 2 * NSLABW + 2 * NC3;
  // streaming layout: patch | A | B | two ring slots
  static constexpr int S_OFF_A = X_BYTES, S_OFF_B = S_OFF_A + A_BYTES, S_OFF_R = S_OFF_B + B_BYTES, S_OFF_BIAS = S_OFF_R + 2 * RING,
                       S_LDS_BYTES = (S_OFF_BIAS + BIAS_BYTES + 2047) & ~2047;
  // (sizes are rounded up to 2 KB: the tail of a dynamic-LDS request that is not a whole number of allocation granules is not
  //  addressable - reads of the last 256 bytes of a 150,272-byte request faulted on MI355X)
  // resident layout: two patch buffers | A | B | cv1|cv2 | rep 3x3 | 3x3 | cv3, slab by slab
  static constexpr int R_OFF_A = 2 * X_BYTES, R_OFF_B = R_OFF_A + A_BYTES, R_OFF_W12 = R_OFF_B + B_BYTES, R_OFF_WR = R_OFF_W12 + NSLABW * SLAB_BIG,
                       R_OFF_WB = R_OFF_WR + NS3 * SLAB_SMALL, R_OFF_W3 = R_OFF_WB + NS3 * SLAB_SMALL, R_OFF_BIAS = R_OFF_W3 + NSLABW * SLAB_BIG,
                       R_LDS_BYTES = (R_OFF_BIAS + BIAS_BYTES + 2047) & ~2047;
  static_assert(T_BYTES + U_BYTES <= X_BYTES, "T and U alias the input patch");
};

template <class T, int HID, bool RES, int SPLIT = 0>
__global__ __launch_bounds__(512, 2) void csp_fused_kernel(const CspP p) {
  using G = CspGeom<HID, SPLIT>;
  constexpr bool SPLIT1 = G::SPLIT1, SPLIT3 = G::SPLIT3;
  static_assert(!(RES && SPLIT), "split weights stream (two planes do not fit beside two patches)");
  constexpr int C2 = G::C2, NSLAB = G::NSLAB, NSLABW = G::NSLABW, PW = G::PW, PR = G::PR, QW = G::QW, QR = G::QR, CPA = G::CPA, PA = G::PA;
  const float os12 = SPLIT1 ? p.os12 : 1.0f, osr = SPLIT3 ? p.osr : 1.0f, osb = SPLIT3 ? p.osb : 1.0f, os3 = SPLIT1 ? p.os3 : 1.0f;   // exact 2^-e output scales
  constexpr int NJ1 = C2 / 16, NJ2 = HID / 32, SPC = G::SPC, NS3 = G::NS3, NC3 = G::NC3;
  constexpr int OFF_A = RES ? G::R_OFF_A : G::S_OFF_A, OFF_B = RES ? G::R_OFF_B : G::S_OFF_B, OFF_BIAS = RES ? G::R_OFF_BIAS : G::S_OFF_BIAS;
  constexpr int BI_12 = OFF_BIAS, BI_R = BI_12 + C2 * 4, BI_B = BI_R + HID * 4, BI_3 = BI_B + HID * 4;
  static_assert(sizeof(T) == 2 && (HID == 32 || HID == 64), "16-bit storage, hidden width 32 or 64");
  static_assert(!RES || G::R_LDS_BYTES <= 160 * 1024, "resident weights do not fit");
  extern __shared__ __attribute__((aligned(16))) uint4 lds[];
  char* ldsb = reinterpret_cast<char*>(lds);
  const unsigned lds_base = lds_addr(lds);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fg = lane >> 4;
  const int pg = wave & 3, chh = wave >> 2;                        // stages 2-4: pixel group / channel half of this wave
  const int total = p.B * p.tiles;

  // ---- tile walk: consecutive tiles (neighbours share halo pixels) stay on one XCD's L2 ---------------------------------------
  // one tile per block (streaming): the bijective blockIdx remap of the conv kernels; persistent: XCD x owns a contiguous range
  // of tiles and its blocks (blockIdx % 8 == x) walk it with stride gridDim / 8
  int tile, tile_end, tile_step;
  {
    const int nwg = gridDim.x, bid = blockIdx.x, xcd = bid & 7, idx = bid >> 3;
    if constexpr (RES) {
      const int q = total >> 3, r = total & 7;
      const int start = xcd * q + (xcd < r ? xcd : r);
      tile = start + idx; tile_end = start + q + (xcd < r ? 1 : 0); tile_step = nwg >> 3;
    } else {
      const int q = nwg >> 3, r = nwg & 7;
      tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx; tile_end = tile + 1; tile_step = 1;
    }
  }
  int b = 0, h0 = 0, w0 = 0;                                       // current tile: image, top-left output pixel
  auto in_image = [&](int ih, int iw) { return (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W; };

  // ---- DMA ---------------------------------------------------------------------------------------------------------------------
  auto issue_x = [&](int t, int x_off) {                           // the 12 x 20 input patch of tile t
    const int tb = fdiv(t, p.tiles, p.inv_tiles), trem = t - tb * p.tiles, ty = fdiv(trem, p.tx, p.inv_tx), tx = trem - ty * p.tx;
    const int th0 = ty * G::TH, tw0 = tx * G::TW;
    constexpr int XPIECES = NSLAB * (PR / 8);                      // 1 KiB pieces: 8 patch rows x 128 bytes
#pragma unroll
    for (int i = 0; i < (XPIECES + 7) / 8; ++i) {
      const int pi = wave + 8 * i;
      if (pi < XPIECES) {
        const int sl = pi / (PR / 8), rg = pi - sl * (PR / 8);
        const int row = rg * 8 + (lane >> 3), pos = lane & 7;
        const int py = row / PW, px = row - py * PW;
        const int ih = th0 - 2 + py, iw = tw0 - 2 + px;
        const int chunk = pos ^ swz<8>(row);
        const void* src = in_image(ih, iw)
            ? static_cast<const void*>(reinterpret_cast<const char*>(p.x) + ((((long)tb * p.H + ih) * p.W + iw) * (long)p.x_cstride + p.x_coff + sl * 64 + chunk * 8) * 2L)
            : static_cast<const void*>(&g_zero16);
        glds16(src, __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(x_off + sl * (PR * 128) + rg * 1024)));
      }
    }
  };
  // rows [0, NR) x 64 k of slabs slab0 .. slab0+NSL-1 of a [rows][kw] weight matrix -> LDS at dst_off, [slab][row][128 B] swizzled by row
  auto issue_w = [&](const void* wbase, int kw, auto nr_c, int slab0, auto nsl_c, unsigned dst_off) {
    constexpr int NR = decltype(nr_c)::value, NSL = decltype(nsl_c)::value, PPS = NR / 8, PIECES = NSL * PPS;
#pragma unroll
    for (int i = 0; i < (PIECES + 7) / 8; ++i) {
      const int pi = wave + 8 * i;
      if (pi < PIECES) {
        const int sl = pi / PPS, rg = pi - sl * PPS;
        const int n = rg * 8 + (lane >> 3), pos = lane & 7;
        const int chunk = pos ^ swz<8>(n);
        const char* src = reinterpret_cast<const char*>(wbase) + ((size_t)n * kw + (size_t)(slab0 + sl) * 64 + chunk * 8) * 2;
        glds16(src, __builtin_amdgcn_readfirstlane(lds_base + dst_off + (unsigned)(sl * (NR * 128) + rg * 1024)));
      }
    }
  };
  // streaming: chunk K of the tile's weight stream goes to ring slot K & 1
  auto issue_chunk = [&](auto k_c) {
    constexpr int K = decltype(k_c)::value;
    constexpr unsigned dst = G::S_OFF_R + (K & 1) * G::RING;
    if constexpr (K < NSLABW) issue_w(p.w12, p.kw12, IC<C2>{}, K, IC<1>{}, dst);
    else if constexpr (K < NSLABW + NC3) { constexpr int s0 = (K - NSLABW) * SPC; issue_w(p.wr, p.kwr, IC<HID>{}, s0, IC<(NS3 - s0 < SPC ? NS3 - s0 : SPC)>{}, dst); }
    else if constexpr (K < NSLABW + 2 * NC3) { constexpr int s0 = (K - NSLABW - NC3) * SPC; issue_w(p.wb, p.kwb, IC<HID>{}, s0, IC<(NS3 - s0 < SPC ? NS3 - s0 : SPC)>{}, dst); }
    else if constexpr (K < G::NCHUNK) issue_w(p.w3, p.kw3, IC<C2>{}, K - NSLABW - 2 * NC3, IC<1>{}, dst);
  };
  // Opens K slab SL of stage ST (0 cv1|cv2, 1 rep 3x3, 2 3x3, 3 cv3) and returns its LDS byte offset.
  // Streaming: at the first slab of a chunk wait for it, barrier (everything written to LDS before is visible, the other ring slot is
  // free again) and issue the next chunk.  Resident: a barrier at the first slab of a stage (the previous stage's LDS writes).
  // Stage 0 slab 0 is opened by the tile loop itself.
  auto open_slab = [&](auto st_c, auto sl_c) -> int {
    constexpr int ST = decltype(st_c)::value, SL = decltype(sl_c)::value;
    if constexpr (RES) {
      if constexpr (SL == 0 && ST == 3) wait_vmcnt<0>();           // the next tile's patch (issued a whole tile ago) has landed
      if constexpr (SL == 0 && (ST == 1 || ST == 2)) __syncthreads();   // stage 4 reads what its own wave wrote (u) and b from stage 1
      return ST == 0 ? G::R_OFF_W12 + SL * G::SLAB_BIG : ST == 1 ? G::R_OFF_WR + SL * G::SLAB_SMALL : ST == 2 ? G::R_OFF_WB + SL * G::SLAB_SMALL : G::R_OFF_W3 + SL * G::SLAB_BIG;
    } else {
      constexpr bool big = ST == 0 || ST == 3;
      constexpr int K = ST == 0 ? SL : ST == 1 ? NSLABW + SL / SPC : ST == 2 ? NSLABW + NC3 + SL / SPC : NSLABW + 2 * NC3 + SL;
      constexpr int within = big ? 0 : (SL % SPC) * G::SLAB_SMALL;
      if constexpr ((big || SL % SPC == 0) && K > 0) { wait_vmcnt<0>(); __syncthreads(); issue_chunk(IC<K + 1>{}); }
      return G::S_OFF_R + (K & 1) * G::RING + within;
    }
  };
  // bias + SiLU of one accumulator fragment -> four storage-type values (8 bytes); `keep` false -> zeros (outside the image)
  auto act4 = [&](const f32x4& av, float osc, int bias_off, bool keep) {
    const float4 b4 = *reinterpret_cast<const float4*>(ldsb + bias_off);
    uint2 pk = make_uint2(pack2<T>(activate<T, 1>(__builtin_fmaf(av[0], osc, b4.x)), activate<T, 1>(__builtin_fmaf(av[1], osc, b4.y))),
                          pack2<T>(activate<T, 1>(__builtin_fmaf(av[2], osc, b4.z)), activate<T, 1>(__builtin_fmaf(av[3], osc, b4.w))));
    if (!keep) pk = make_uint2(0u, 0u);
    return pk;
  };
  // row-major [rows][PA] intermediates: byte offset of channel n of row `row`
  auto elem_off = [&](int region, int row, int n) { return region + row * PA + n * 2; };
  auto out_pix = [&](int q) {
    const int ho = h0 + (q >> 4), wo = w0 + (q & 15);
    return (ho < p.H && wo < p.W) ? ((long)b * p.H + ho) * p.W + wo : -1L;
  };
  // development: copy an intermediate (inner 8 x 16 pixels) to the output view instead of finishing the block
  auto dump = [&](int region, auto rowfn, int ch_off) {
    for (int idx = tid; idx < 128 * CPA; idx += 512) {
      const int q = idx / CPA, ch = idx - q * CPA, row = rowfn(q);
      const uint4 v = *reinterpret_cast<const uint4*>(ldsb + region + row * PA + ch * 16);
      const long m = out_pix(q);
      if (m >= 0) *reinterpret_cast<uint4*>(reinterpret_cast<T*>(p.out) + m * p.out_cstride + p.out_coff + ch_off + ch * 8) = v;
    }
  };
  // 3x3 over an LDS-resident map: pixel fragment i reads 16 bytes at fp[i] + (r * SRCW + s) * PA (+ 64 for the second K half at
  // HID 64): one base pointer per fragment, every tap a compile-time offset.  K slab J of the stage's weights at woff.
  auto conv3_slab = [&](auto j_c, int woff, auto npx_c, auto srcw_c, auto nj_c, int ch0, const auto& fp, auto& acc) {
    constexpr int J = decltype(j_c)::value, NPX = decltype(npx_c)::value, SRCW = decltype(srcw_c)::value, NJ = decltype(nj_c)::value;
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
      // a 64-wide K slab is one tap (HID 64) or two (HID 32); split: [hi | lo] of one tap (HID 32: the k halves read the SAME 32
      // channels against the two planes) or one plane of one tap (HID 64: two slabs per tap)
      const int tap = SPLIT3 ? (HID == 64 ? J / 2 : J) : (HID == 64 ? J : 2 * J + kh);
      if (tap < 9) {
        const int r = tap / 3, s = tap - r * 3;
        uint4 xf[NPX], wf[NJ];
#pragma unroll
        for (int i = 0; i < NPX; ++i) xf[i] = *reinterpret_cast<const uint4*>(fp[i] + (r * SRCW + s) * PA + (HID == 64 ? kh * 64 : 0));
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) { const int n = (ch0 + jj) * 16 + fr; wf[jj] = *reinterpret_cast<const uint4*>(ldsb + woff + n * 128 + (((kh * 4 + fg) ^ swz<8>(n)) << 4)); }
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
          for (int i = 0; i < NPX; ++i) Mma<T>::run(wf[jj], xf[i], acc[jj][i]);
      }
    }
  };

  // ---- prologue ----------------------------------------------------------------------------------------------------------------
  int x_off = 0;                                                   // byte offset of the current tile's patch (T and U alias it)
  if (tid < G::BIAS_BYTES / 4) {                                                  // the four bias vectors -> LDS (published by the first barrier)
    const float* src = tid < C2 ? p.b12 + tid : tid < C2 + HID ? p.br + (tid - C2) : tid < C2 + 2 * HID ? p.bb + (tid - C2 - HID) : p.b3 + (tid - C2 - 2 * HID);
    reinterpret_cast<float*>(ldsb + OFF_BIAS)[tid] = *src;
  }
  if (tile < tile_end) issue_x(tile, 0);
  if constexpr (RES) {
    issue_w(p.w12, p.kw12, IC<C2>{}, 0, IC<NSLAB>{}, G::R_OFF_W12);
    issue_w(p.wr, p.kwr, IC<HID>{}, 0, IC<NS3>{}, G::R_OFF_WR);
    issue_w(p.wb, p.kwb, IC<HID>{}, 0, IC<NS3>{}, G::R_OFF_WB);
    issue_w(p.w3, p.kw3, IC<C2>{}, 0, IC<NSLAB>{}, G::R_OFF_W3);
  } else {
    issue_chunk(IC<0>{});
  }

  wait_vmcnt<0>();                                                 // first patch, resident weights / first chunk
  for (; tile < tile_end; tile += tile_step) {
    {
      b = fdiv(tile, p.tiles, p.inv_tiles);
      const int trem = tile - b * p.tiles, ty = fdiv(trem, p.tx, p.inv_tx), tx = trem - ty * p.tx;
      h0 = ty * G::TH; w0 = tx * G::TW;
    }
    const int off_t = x_off, off_u = x_off + G::T_BYTES;
    // every wave is done with the previous tile (its staging copy-out out of A, its reads of B and U) and the other patch buffer is
    // free.  This tile's patch has landed: the first one was waited for above, later ones at the opening of the previous tile's
    // stage 4 - so that the previous tile's output stores are NOT waited for here and drain under this tile.
    __syncthreads();
    if constexpr (RES) { if (tile + tile_step < tile_end) issue_x(tile + tile_step, G::X_BYTES - x_off); }
    else issue_chunk(IC<1>{});

    // ---- stage 1: [a | b] = SiLU(W12 x + b12); wave w owns patch-pixel fragments w and w + 8, all 2*HID channels ---------------
    {
      f32x4 acc[NJ1][2];
#pragma unroll
      for (int j = 0; j < NJ1; ++j) { acc[j][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[j][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
      int xrow[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) { const int row = 16 * (wave + 8 * i) + fr; xrow[i] = row < PR ? row : PR - 1; }   // fragment 15 is half empty: clamp
      static_for<NSLABW>([&](auto s_c) {
        constexpr int S = decltype(s_c)::value, XS = S % NSLAB;      // weight slab S multiplies x slab XS (split: the low plane walks x again)
        const int woff = open_slab(IC<0>{}, IC<S>{});
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
          uint4 xf[2], wf[NJ1];
#pragma unroll
          for (int i = 0; i < 2; ++i) xf[i] = *reinterpret_cast<const uint4*>(ldsb + x_off + XS * (PR * 128) + xrow[i] * 128 + (((kh * 4 + fg) ^ swz<8>(xrow[i])) << 4));
#pragma unroll
          for (int j = 0; j < NJ1; ++j) { const int n = 16 * j + fr; wf[j] = *reinterpret_cast<const uint4*>(ldsb + woff + n * 128 + (((kh * 4 + fg) ^ swz<8>(n)) << 4)); }
#pragma unroll
          for (int j = 0; j < NJ1; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i) Mma<T>::run(wf[j], xf[i], acc[j][i]);
        }
      });
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = 16 * (wave + 8 * i) + fr;                  // rows 240..255 land in A's padding and are never read
        const int py = row / PW, px = row - py * PW;
        const bool inimg = in_image(h0 - 2 + py, w0 - 2 + px);
        const bool inner = py >= 2 && py < 2 + G::TH && px >= 2 && px < 2 + G::TW;
        const int q = (py - 2) * G::TW + (px - 2);
#pragma unroll
        for (int j = 0; j < NJ1; ++j) {
          const int n = 16 * j + 4 * fg;
          if (j < HID / 16) *reinterpret_cast<uint2*>(ldsb + elem_off(OFF_A, row, n)) = act4(acc[j][i], os12, BI_12 + n * 4, inimg);
          else if (inner) *reinterpret_cast<uint2*>(ldsb + elem_off(OFF_B, q, n - HID)) = act4(acc[j][i], os12, BI_12 + n * 4, true);
        }
      }
    }
    if (p.dbg == 1) {
      wait_vmcnt<0>(); __syncthreads();
      dump(OFF_A, [](int q) { return ((q >> 4) + 2) * PW + (q & 15) + 2; }, 0);
      dump(OFF_B, [](int q) { return q; }, HID);
      if constexpr (RES) { x_off = G::X_BYTES - x_off; continue; } else return;
    }

    // ---- stage 2: t = SiLU(Wr * a + br) on the 10 x 18 ring; wave = (pixel fragments 3pg..3pg+2) x (channel half chh) ------------
    {
      f32x4 acc[NJ2][3];
      const char* fp[3]; int qq[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        qq[i] = 16 * (3 * pg + i) + fr;
        const int qc = qq[i] < QR ? qq[i] : QR - 1;
        const int ty_ = qc / QW, tx_ = qc - ty_ * QW;
        fp[i] = ldsb + OFF_A + (ty_ * PW + tx_) * PA + fg * 16;
#pragma unroll
        for (int jj = 0; jj < NJ2; ++jj) acc[jj][i] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      static_for<NS3>([&](auto j_c) {
        const int woff = open_slab(IC<1>{}, j_c);
        conv3_slab(j_c, woff, IC<3>{}, IC<PW>{}, IC<NJ2>{}, chh * NJ2, fp, acc);
      });
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int q = qq[i];
        if (q < QR) {
          const int ty_ = q / QW, tx_ = q - ty_ * QW;
          const bool inimg = in_image(h0 - 1 + ty_, w0 - 1 + tx_);
#pragma unroll
          for (int jj = 0; jj < NJ2; ++jj) {
            const int n = (chh * NJ2 + jj) * 16 + 4 * fg;
            *reinterpret_cast<uint2*>(ldsb + elem_off(off_t, q, n)) = act4(acc[jj][i], osr, BI_R + n * 4, inimg);
          }
        }
      }
    }
    if (p.dbg == 2) {
      wait_vmcnt<0>(); __syncthreads();
      dump(off_t, [](int q) { return ((q >> 4) + 1) * QW + (q & 15) + 1; }, 0);
      if constexpr (RES) { x_off = G::X_BYTES - x_off; continue; } else return;
    }

    // ---- stage 3: u = a + SiLU(Wb * t + bb) on the 8 x 16 inner pixels.  From here on a wave OWNS output row `wave` (one 16-pixel
    // fragment) with all its channels: stage 4 then reads only what its own wave wrote (u) or what stage 1 wrote two barriers ago
    // (b), and the store epilogue stages through the wave's own rows - no barrier between stage 3 and the end of the tile.
    constexpr int NJ3 = HID / 16;
    const int qrow = 16 * wave + fr;                               // this lane's inner pixel
    {
      f32x4 acc[NJ3][1];
      const char* fp[1] = {ldsb + off_t + (wave * QW + fr) * PA + fg * 16};
#pragma unroll
      for (int jj = 0; jj < NJ3; ++jj) acc[jj][0] = f32x4{0.f, 0.f, 0.f, 0.f};
      static_for<NS3>([&](auto j_c) {
        const int woff = open_slab(IC<2>{}, j_c);
        conv3_slab(j_c, woff, IC<1>{}, IC<QW>{}, IC<NJ3>{}, 0, fp, acc);
      });
      const int arow = (wave + 2) * PW + fr + 2;
#pragma unroll
      for (int jj = 0; jj < NJ3; ++jj) {
        const int n = 16 * jj + 4 * fg;
        const float4 b4 = *reinterpret_cast<const float4*>(ldsb + BI_B + n * 4);
        const f32x4 av = acc[jj][0];
        const uint2 ru = *reinterpret_cast<const uint2*>(ldsb + elem_off(OFF_A, arow, n));
        const T* rt = reinterpret_cast<const T*>(&ru);
        *reinterpret_cast<uint2*>(ldsb + elem_off(off_u, qrow, n)) =
            make_uint2(pack2<T>(to_f32<T>(rt[0]) + activate<T, 1>(__builtin_fmaf(av[0], osb, b4.x)), to_f32<T>(rt[1]) + activate<T, 1>(__builtin_fmaf(av[1], osb, b4.y))),
                       pack2<T>(to_f32<T>(rt[2]) + activate<T, 1>(__builtin_fmaf(av[2], osb, b4.z)), to_f32<T>(rt[3]) + activate<T, 1>(__builtin_fmaf(av[3], osb, b4.w))));
      }
    }
    if (p.dbg == 3) {
      wait_vmcnt<0>(); __syncthreads();
      dump(off_u, [](int q) { return q; }, 0);
      dump(OFF_B, [](int q) { return q; }, HID);
      if constexpr (RES) { x_off = G::X_BYTES - x_off; continue; } else return;
    }

    // ---- stage 4: out = SiLU(W3 [u | b] + b3) for the wave's 16 pixels, all 2*HID channels ----------------------------------------
    f32x4 acc4[NJ1][1];
#pragma unroll
    for (int jj = 0; jj < NJ1; ++jj) acc4[jj][0] = f32x4{0.f, 0.f, 0.f, 0.f};
    static_for<NSLABW>([&](auto s_c) {
      constexpr int S = decltype(s_c)::value % NSLAB;                          // the [u | b] slab this weight slab multiplies
      const int woff = open_slab(IC<3>{}, s_c);
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) {
        const int region = (HID == 64 ? S : kh) == 0 ? off_u : OFF_B;          // K order of cv3: the m-branch channels, then cv2's
        const int pc = HID == 64 ? kh * 4 + fg : fg;
        const uint4 xf = *reinterpret_cast<const uint4*>(ldsb + region + qrow * PA + pc * 16);
        uint4 wf[NJ1];
#pragma unroll
        for (int jj = 0; jj < NJ1; ++jj) { const int n = 16 * jj + fr; wf[jj] = *reinterpret_cast<const uint4*>(ldsb + woff + n * 128 + (((kh * 4 + fg) ^ swz<8>(n)) << 4)); }
#pragma unroll
        for (int jj = 0; jj < NJ1; ++jj) Mma<T>::run(wf[jj], xf, acc4[jj][0]);
      }
    });
    // bias + SiLU, then through the wave's OWN u rows (channels 0..HID-1) and b rows (HID..2*HID-1) - both dead for everybody else and,
    // after the MFMAs above, for this wave - so that every pixel's 2*HID channels leave as 16-byte-per-lane stores.  Same-wave LDS
    // traffic only: no barrier.
    {
      constexpr int OCPR = C2 / 8;                                 // 16-byte chunks per pixel
#pragma unroll
      for (int jj = 0; jj < NJ1; ++jj) {
        const int nl = 16 * jj + 4 * fg;
        *reinterpret_cast<uint2*>(ldsb + (jj < NJ3 ? elem_off(off_u, qrow, nl) : elem_off(OFF_B, qrow, nl - HID))) = act4(acc4[jj][0], os3, BI_3 + nl * 4, true);
      }
      T* outp = reinterpret_cast<T*>(p.out) + p.out_coff;
#pragma unroll
      for (int it = 0; it < 16 * OCPR / 64; ++it) {
        const int idx = lane + 64 * it, r = idx / OCPR, ch = idx - r * OCPR, q = 16 * wave + r;
        const long m = out_pix(q);
        const uint4 v = *reinterpret_cast<const uint4*>(ldsb + (ch < CPA ? off_u + q * PA + ch * 16 : OFF_B + q * PA + (ch - CPA) * 16));
        if (m >= 0) *reinterpret_cast<uint4*>(outp + m * p.out_cstride + ch * 8) = v;
      }
    }
    if constexpr (RES) x_off = G::X_BYTES - x_off;
  }
}

// csp_tile.hip (round 6): hidden width 32 with every weight resident in LDS in fragment order and 16 x 32-pixel tiles
bool csp_tile_supported(int dt, int hid, int split);
void launch_csp_tile(int dt, const CspP& p, hipStream_t stream);

bool csp_fused_supported(int dt, int hid, int split) { return (dt == F16 || (dt == BF16 && !split)) && (hid == 32 || hid == 64); }

template <class T, int HID, bool RES, int SPLIT = 0> static void launch_csp(const CspP& p, hipStream_t stream) {
  constexpr int lds = RES ? CspGeom<HID, SPLIT>::R_LDS_BYTES : CspGeom<HID, SPLIT>::S_LDS_BYTES;
  static PerDevice pd;                                 // attribute and CU count per device ordinal (common.h)
  const int pdi = pd.index();
  if (pd.first(pdi))
    CC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(csp_fused_kernel<T, HID, RES, SPLIT>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  const int cus = pd.cu_count(pdi);
  const int total = p.B * p.tiles;
  // persistent: one block per CU, a multiple of 8 so that every XCD gets the same number of walkers
  const int grid = RES ? std::max(8, std::min(cus, total) & ~7) : total;
  note_launch("csp_fused", csp_fused_kernel<T, HID, RES, SPLIT>, (long)total, 512, lds, RES ? grid : 0);
  hipLaunchKernelGGL((csp_fused_kernel<T, HID, RES, SPLIT>), dim3(grid), dim3(512), lds, stream, p);
}

void launch_csp_fused(int dt, const CspP& p0, hipStream_t stream) {
  CC_CHECK(csp_fused_supported(dt, p0.hid, p0.split), "fused RepNCSP: 16-bit storage (split weights: f16) and hidden width 32 or 64 only");
  CC_CHECK(p0.x_cstride % 8 == 0 && p0.x_coff % 8 == 0 && p0.out_cstride % 8 == 0 && p0.out_coff % 8 == 0 && (((uintptr_t)p0.x | (uintptr_t)p0.out) & 15) == 0,
           "fused RepNCSP: views must be 16-byte aligned");
  CC_CHECK(p0.split >= 0 && p0.split <= 2, "fused RepNCSP: split is 0 (one plane), 1 (two planes everywhere) or 2 (two planes in the 1x1 convs)");
  const int s1 = p0.split != 0, s3 = p0.split == 1;
  CC_CHECK(p0.kw12 % 64 == 0 && p0.kw3 % 64 == 0 && p0.kwr % 64 == 0 && p0.kwb % 64 == 0 && p0.kwr >= 9 * p0.hid * (1 + s3) && p0.kwb >= 9 * p0.hid * (1 + s3) &&
           p0.kw12 >= 2 * p0.hid * (1 + s1) && p0.kw3 >= 2 * p0.hid * (1 + s1), "fused RepNCSP: weight rows must cover whole K slabs");
  // CLEARCAM_CSP_TILE=0 keeps the round-2 kernel at hidden width 32 (A/B, tools/dev), =2 takes the round-6 kernel at every batch size
  // (tests: small batches, ragged tiles); read once per process
  static const int tile_mode = [] { const char* e = getenv("CLEARCAM_CSP_TILE"); return e ? atoi(e) : 1; }();
  const bool tile_on = tile_mode != 0;
  // ... from one round of its 16 x 32 tiles on (256 tiles: B >= 6 at 160 x 160; measured a tie from 200 tiles and 1.6 % ahead at 400,
  // profiles/r06n_csp_tile_threshold.txt): below that (a single frame: 50 tiles, +0.04 ms) this file's 128-pixel tiles spread over more CUs.
  // Both are bit-identical to the four launches they replace, so the choice may depend on the batch size.
  const long tiles32 = (long)p0.B * ((p0.H + 15) / 16) * ((p0.W + 31) / 32);
  if (tile_on && !p0.stream && !p0.dbg && (tiles32 >= 256 || tile_mode == 2) && csp_tile_supported(dt, p0.hid, p0.split)) { launch_csp_tile(dt, p0, stream); return; }
  CspP p = p0;
  p.tx = (p.W + 15) / 16; p.tiles = ((p.H + 7) / 8) * p.tx;
  p.inv_tiles = 1.0f / (float)p.tiles; p.inv_tx = 1.0f / (float)p.tx;
  CC_CHECK((long)p.B * p.tiles < (1L << 22), "fused RepNCSP: too many tiles");
  const bool res = p.hid == 32 && !p.stream;                        // 56 KB of weights stay in LDS; 208 KB (hidden 64) cannot
  if (p.split == 1) {                                               // two weight planes: streamed at both widths
    if (p.hid == 64) launch_csp<f16_t, 64, false, 1>(p, stream); else launch_csp<f16_t, 32, false, 1>(p, stream);
  } else if (p.split == 2) {                                        // ... in the 1x1 convs only (72 KB of weights at hidden 32: streamed as well)
    if (p.hid == 64) launch_csp<f16_t, 64, false, 2>(p, stream); else launch_csp<f16_t, 32, false, 2>(p, stream);
  } else if (dt == F16) {
    if (p.hid == 64) launch_csp<f16_t, 64, false>(p, stream);
    else if (res) launch_csp<f16_t, 32, true>(p, stream); else launch_csp<f16_t, 32, false>(p, stream);
  } else {
    if (p.hid == 64) launch_csp<bf16_t, 64, false>(p, stream);
    else if (res) launch_csp<bf16_t, 32, true>(p, stream); else launch_csp<bf16_t, 32, false>(p, stream);
  }
  CC_HIP(hipGetLastError());
}

}  // namespace cc

// 3x3 stride-1 convolution, 64 -> 64 channels, for the wide maps of the detector (the 3x3 that follows each RepNCSP of the first ELAN
// block at 160 x 160, DDetect's grouped box conv at 80 x 80): round 6, built from what csp_tile.hip taught.
//
// conv3x3_wave_kernel (one autonomous wave per 2 x 16 strip) runs these at 645 TFLOP/s: 6 ds_reads per 8 MFMAs with per-read swizzle
// arithmetic, a 2.25x patch over-read per strip, 8-byte stores.  Here a persistent block of eight waves works as TWO GROUPS of four, each
// with its own stream of 256-pixel tiles (16 x 16, or 8 x 32 where that covers the map with fewer), half a tile out of phase:
//  * the 72 KB of weights stay in LDS for the life of the block in MFMA-FRAGMENT ORDER (one 1 KB image per (k step, 16-row fragment):
//    an A operand is one conflict-free ds_read_b128 at a compile-time offset), shared by both groups;
//  * each group owns ONE 43.5 KB buffer for the 10 x 34 input patch of its tile (halo 1; zero page outside the image): 128-byte pixel rows
//    with chunk ^= ((pixel >> 1) & 3) << 1 - conflict-free for 16 consecutive pixels at ANY start, which is what the nine shifted windows are;
//  * phases alternate: while group 0 runs the K loop of its tile (18 k steps of 16 MFMAs + 8 ds_reads per wave, software-pipelined by
//    hand), group 1 requests its NEXT patch (its buffer is free: its own K loop ended with the previous phase), turns the accumulators of
//    its PREVIOUS tile into 16-bit outputs (bias + SiLU: 128 transcendentals per lane), waits for the patch and issues its stores; one
//    block barrier; roles swap.  Waves w and w + 4 share a SIMD, so every SIMD always holds one wave on the matrix pipe and one on the
//    vector ALU / memory side;
//  * measured (B = 64, 160 x 160, f16; profiles/r06q_tile64.txt): 149 us against 208 for the wave-autonomous kernel and 166 for the
//    one-phase form of this kernel (all eight waves in step).  In-kernel cycle stamps (development build, ABL 128): a K loop takes
//    3.9 us - it is paced by the wave's LDS reads, not by its MFMAs: ONE wave gets a ds_read_b128 per ~30 ticks of s_memtime whatever
//    the other waves do (tools/dev/lds_rate.hip: a batch of eight returns in 308 ticks with 4 or 8 waves per CU, swizzled pixel rows
//    and contiguous weight images alike), so 8 reads per 16 MFMAs (187 ticks of matrix pipe) leave the pipe half idle; a finish phase
//    takes 4.3 us beside a K loop: 0.7 requesting the patch (11 LDS-DMA issues), 2.7 in the activation arithmetic (which shares the
//    SIMD's issue port with the other wave's MFMAs), 0.8 issuing the stores; memory alone (no K loop) is 85 us = 5.6 TB/s, the K
//    loops alone 100 us.  Tried and measured slower: the stores moved to the start of the group's next K phase (171 us: eight store
//    issues stall the MFMA wave for 1-2 us), the K loop at priority 1 (159), reads two k steps ahead (no change: lgkmcnt counts to 15);
//  * wave w of a group owns output rows 2w, 2w + 1 of the tile (2 x 2 pixel fragments x four channel fragments = 64 accumulator registers);
//  * MFMA row i of fragment pair (2s, 2s+1) is channel 32s + (i >> 2) * 8 + h * 4 + (i & 3), so a lane ends up with eight consecutive
//    channels of its pixel: bias + activation in registers, ONE 16-byte store per 32 channels;
//  * a 16-bit residual (the bottleneck's shortcut) is read as one 16-byte piece per 8 channels at the start of the finish phase and added after the activation;
//  * a wave waits for its share of the next patch (vmcnt(0)) BEFORE it issues its output stores - those are never waited for on their
//    own (the next vmcnt(0) of the wave is a whole K loop later).
// K order (tap, 32-channel step) is that of every other conv kernel: same bits (tests/test_gpu_yolo.py::test_tile64_3x3_equals_generic).
#include <utility>
#include "conv_tile.h"

namespace cc {

namespace {
template <int V> using ICv = std::integral_constant<int, V>;
template <int... I, class F> __device__ __forceinline__ void sfor64_impl(std::integer_sequence<int, I...>, F&& f) { (f(ICv<I>{}), ...); }
template <int N, class F> __device__ __forceinline__ void sfor64(F&& f) { sfor64_impl(std::make_integer_sequence<int, N>{}, f); }
}  // namespace

struct Tile64Aux { float inv_tiles, inv_tx; int tiles, tx, total; };

// TWIDTH = 32: 8 x 32-pixel tiles (a wave of a group owns 2 rows x 2 fragments); 16: 16 x 16 (4 rows x 1 fragment) - for maps whose width 32 does not
// divide: an 80 x 80 map is 25 tiles of 16 x 16 against 30 of 8 x 32.  Same 256 pixels, same MFMAs and reads per wave, 324 patch pixels instead of 340.
template <int TWIDTH> struct Tile64Geom {
  static constexpr int TW = TWIDTH, TH = 256 / TW, PH = TH + 2, PW = TW + 2, PR = PH * PW;   // 10 x 34 = 340 / 18 x 18 = 324 patch pixels
  static constexpr int RW = TH / 4, FW = TW / 16;                                         // rows and 16-pixel fragments per row of one wave
  static constexpr int NPIECE = (PR * 8 + 63) / 64;                                       // 43 DMA pieces of 1 KB (64 lanes x 16 B)
  static constexpr int PPW = (NPIECE + 3) / 4;                                            // pieces per wave of a group: 11
  static constexpr int PATCH_BYTES = NPIECE * 1024;                                       // 44 032
  static constexpr int NKS = 18;                                                          // k steps: tap * 2 + k half
  static constexpr int OFF_W = 0, OFF_BIAS = NKS * 4 * 1024, OFF_P = OFF_BIAS + 256, LDS_BYTES = (OFF_P + 2 * PATCH_BYTES + 2047) & ~2047;
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

// ABL (development builds, -DCC_TILE64_ABLATIONS; timing only, results are WRONG with any bit set): 1 = no patch DMA after the first, 2 = no
// global stores, 4 = no activation arithmetic, 8 = no MFMA (fragment reads stay), 16 = no K loop at all, 32 = no finish phase (the accumulators
// are kept alive), 64 = fragment reads two k steps ahead instead of one, 256 = the K loop at priority 1,
// 512 / 1024 = the finish phase at priority 3 while it requests the patch / at priority 1 throughout, 128 = cycle counts per phase kind behind the output.
int g_tile64_abl = 0;
int g_tile64_w = -1;

template <class T, int ABL = 0, int TWIDTH = 32>
__global__ __launch_bounds__(512) void conv3x3_tile64_kernel(const ConvP p, const Tile64Aux a) {
  using G = Tile64Geom<TWIDTH>;
  constexpr int RW = G::RW, FW = G::FW;
  constexpr int PW = G::PW;
  static_assert(sizeof(T) == 2, "16-bit storage");
  extern __shared__ __attribute__((aligned(16))) uint4 lds[];
  char* ldsb = reinterpret_cast<char*>(lds);
  const unsigned lds_base = lds_addr(lds);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wg = wave & 3;                          // waves w and w + 4 share a SIMD: one of each group per SIMD
  int fr = lane & 15, fg = lane >> 4;

  // ---- tile walk: XCD x owns a contiguous range of tiles (neighbours share halo pixels in its L2); a block's tiles alternate between its groups
  int tile0, tile_step, n;
  {
    const int nwg = gridDim.x, bid = blockIdx.x, xcd = bid & 7, idx = bid >> 3;
    const int q = a.total >> 3, r = a.total & 7;
    const int start = xcd * q + (xcd < r ? xcd : r), tile_end = start + q + (xcd < r ? 1 : 0);
    tile0 = start + idx; tile_step = nwg >> 3;
    n = tile0 < tile_end ? (tile_end - tile0 + tile_step - 1) / tile_step : 0;
  }
  if (n == 0) return;                                                // (before any barrier: whole blocks only)
  const int ng = (n + 1 - grp) >> 1;                                 // this group's tiles: the block's tiles 2 it + grp
  auto tile_of = [&](int it) { return tile0 + (2 * it + grp) * tile_step; };

  // ---- prologue: weights -> LDS in fragment order; fragment (ks, j): lane l holds row n(j, l & 15), k = 32 ks + 8 (l >> 4) .. + 7 -----
  {
    const int nrow = (fr >> 2) * 8 + (fr & 3);
    for (int i = wave; i < G::NKS * 4; i += 8) {
      const int ks = i >> 2, j = i & 3;
      const int nn = (j >> 1) * 32 + (j & 1) * 4 + nrow;
      const char* src = reinterpret_cast<const char*>(p.w) + ((size_t)nn * p.Kw + ks * 32 + fg * 8) * 2;
      glds16_m0(src, __builtin_amdgcn_readfirstlane(lds_base + G::OFF_W + i * 1024));
    }
    if (tid < 64) reinterpret_cast<float*>(ldsb + G::OFF_BIAS)[tid] = p.bias ? p.bias[tid] : 0.0f;
  }
  auto tile_origin = [&](int t, int& b, int& h0, int& w0) {
    b = fdiv(t, a.tiles, a.inv_tiles);
    const int trem = t - b * a.tiles, ty = fdiv(trem, a.tx, a.inv_tx);
    h0 = ty * G::TH; w0 = (trem - ty * a.tx) * G::TW;
  };
  // patch -> the group's buffer: piece pi = wg + 4 k covers patch pixels 8 pi .. 8 pi + 7, lane l = pixel l >> 3, LDS chunk l & 7 (holding
  // global chunk (l & 7) ^ swizzle).  What does not depend on the tile is kept per lane: byte offset from the patch origin, (py, px).
  const char* xbase = reinterpret_cast<const char*>(p.s0.ptr) + (size_t)p.s0.coff * 2;
  int poff[G::PPW], ppos[G::PPW];
#pragma unroll
  for (int k = 0; k < G::PPW; ++k) {
    const int q = (wg + 4 * k) * 8 + (lane >> 3), py = q / PW, px = q - py * PW;
    const int chunk = (lane & 7) ^ (((q >> 1) & 3) << 1);
    poff[k] = ((py * p.s0.W + px) * p.s0.cstride + chunk * 8) * 2;
    ppos[k] = q < G::PR ? (py | (px << 16)) : 0x7fff;               // pixels past the patch: never in the image
  }
  const unsigned pbuf = __builtin_amdgcn_readfirstlane(lds_base + G::OFF_P + grp * G::PATCH_BYTES + wg * 1024);
  auto issue_patch = [&](int t) {
    int b, h0, w0; tile_origin(t, b, h0, w0);
    const char* org = xbase + (((long)b * p.s0.H + (h0 - 1)) * p.s0.W + (w0 - 1)) * (long)p.s0.cstride * 2;
#pragma unroll
    for (int k = 0; k < G::PPW; ++k) {
      if (wg + 4 * k < G::NPIECE) {
        const int ih = h0 - 1 + (ppos[k] & 0xffff), iw = w0 - 1 + (ppos[k] >> 16);
        const bool ok = (unsigned)ih < (unsigned)p.Hin && (unsigned)iw < (unsigned)p.Win;
        const void* src = ok ? static_cast<const void*>(org + poff[k]) : static_cast<const void*>(&g_zero16);
        glds16_m0(src, pbuf + k * 4096);
      }
    }
  };
  if (ng > 0) issue_patch(tile_of(0));
  wait_vmcnt<0>();                                                   // weights and the group's first patch
  __syncthreads();
  const float osc = out_scale(p);
  T* outp = reinterpret_cast<T*>(p.out) + p.out_coff;
  const int pbase = G::OFF_P + grp * G::PATCH_BYTES;

  unsigned long long tk = 0, tf = 0, tb = 0, t_start = 0;            // ABL 128: cycles in compute / finish / barrier, written behind the output by block 0
  if constexpr (ABL & 128) t_start = __builtin_readcyclecounter();
  f32x4 acc[RW][FW][4];                                                // [row][pixel fragment][channel fragment]: written by a compute phase, read by the next finish phase
  unsigned long long ts[3] = {0, 0, 0};                              // ABL 128: inside the finish phases - until the patch is requested / the outputs are packed / the patch has landed
  const int nph = n + 1;                                             // group g computes its tile `it` in phase 2 it + g and finishes it in phase 2 it + g + 1
  for (int ph = 0; ph < nph; ++ph) {
    asm volatile("" : "+v"(fr), "+v"(fg));                           // keeps tile-invariant per-lane addresses out of scratch (csp_tile.hip)
    const int it = ph >> 1;
    unsigned long long t0 = 0;
    if constexpr (ABL & 128) t0 = __builtin_readcyclecounter();
    if ((ph & 1) == grp) {
      // ---- compute: the K loop of tile `it` ------------------------------------------------------------------------------------------
      if (it < ng && !(ABL & 16)) {
#pragma unroll
        for (int r = 0; r < RW; ++r)
#pragma unroll
          for (int i = 0; i < FW; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[r][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int q0 = (RW * wg) * PW + fr;                           // patch pixel of tap (0, 0), row 0, fragment 0
        auto pix_addr = [&](int q, int c) { const int v = q * 128 + c * 16; return pbase + (v ^ ((v >> 3) & 0x60)); };
        constexpr int PD = (ABL & 64) ? 2 : 1, NB = PD + 1;         // fragment reads run PD k steps ahead of the MFMAs that use them
        uint4 xf[NB][RW][FW], wf[NB][4];
        auto rd = [&](auto ks_c) {
          constexpr int KS = decltype(ks_c)::value, BUF = KS % NB, TAP = KS >> 1, KH = KS & 1, R = TAP / 3, S = TAP - R * 3;
#pragma unroll
          for (int j = 0; j < 4; ++j) wf[BUF][j] = *reinterpret_cast<const uint4*>(ldsb + G::OFF_W + (KS * 4 + j) * 1024 + lane * 16);
#pragma unroll
          for (int r = 0; r < RW; ++r)
#pragma unroll
            for (int i = 0; i < FW; ++i) xf[BUF][r][i] = *reinterpret_cast<const uint4*>(ldsb + pix_addr(q0 + (r + R) * PW + S + 16 * i, 4 * KH + fg));
        };
        sfor64<PD>([&](auto k_c) { rd(k_c); });
        if constexpr (ABL & 256) __builtin_amdgcn_s_setprio(1);
        sfor64<G::NKS>([&](auto ks_c) {
          constexpr int KS = decltype(ks_c)::value, CUR = KS % NB;
          if constexpr (KS + PD < G::NKS) rd(ICv<KS + PD>{});
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int r = 0; r < RW; ++r)
#pragma unroll
            for (int i = 0; i < FW; ++i)
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                if constexpr (ABL & 8) asm volatile("" :: "v"(wf[CUR][j].x), "v"(wf[CUR][j].w), "v"(xf[CUR][r][i].x), "v"(xf[CUR][r][i].w));
                else Mma<T>::run(wf[CUR][j], xf[CUR][r][i], acc[r][i][j]);
              }
          __builtin_amdgcn_sched_barrier(0);
        });
        __builtin_amdgcn_s_setprio(0);
      }
    } else {
      // ---- finish: request the next patch, turn the previous tile's accumulators into outputs, wait for the patch, store ------------------
      const int pi = grp == 0 ? it : it - 1, ni = pi + 1;            // group 0 finishes in odd phases (tile it), group 1 in even ones (tile it - 1)
      if constexpr (ABL & 1024) __builtin_amdgcn_s_setprio(1);
      if constexpr (ABL & 512) __builtin_amdgcn_s_setprio(3);
      if (ni >= 1 && ni < ng && !(ABL & (1 | 32))) issue_patch(tile_of(ni));   // (a group's tile 0 was requested in the prologue)
      if constexpr (ABL & 512) __builtin_amdgcn_s_setprio(0);
      if constexpr (ABL & 128) ts[0] += __builtin_readcyclecounter() - t0;
      if constexpr (ABL & 32) {
#pragma unroll
        for (int r = 0; r < RW; ++r)
#pragma unroll
          for (int i = 0; i < FW; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("" :: "v"(acc[r][i][j]));
      } else if (pi >= 0 && pi < ng) {
        uint4 ov[RW][FW][2];                                           // [row][pixel fragment][channel half]
        int b, h0, w0; tile_origin(tile_of(pi), b, h0, w0);
        // the shortcut of a RepNBottleneck (detection/yolov9.py:82-89: x + cv2(cv1(x))): the residual's 16-byte pieces are requested now and
        // added after the activation arithmetic, when they have long landed.  Same order as every other epilogue: act(fma(acc, scale, bias)) + r.
        uint4 rv[RW][FW][2];
        if (p.res) {
          const T* resp = reinterpret_cast<const T*>(p.res) + p.res_coff + fg * 8;
#pragma unroll
          for (int r = 0; r < RW; ++r)
#pragma unroll
            for (int i = 0; i < FW; ++i) {
              const int ho = h0 + RW * wg + r, wo = w0 + 16 * i + fr;
              const bool ok = ho < p.Ho && wo < p.Wo;
              const T* src = resp + (((size_t)b * p.Ho + (ok ? ho : 0)) * p.Wo + (ok ? wo : 0)) * (size_t)p.res_cstride;
              rv[r][i][0] = *reinterpret_cast<const uint4*>(src); rv[r][i][1] = *reinterpret_cast<const uint4*>(src + 32);
            }
        }
        auto add2 = [](float a0, float a1, unsigned rr) -> unsigned {   // two activated values + the two residual values packed in rr
          const T* t = reinterpret_cast<const T*>(&rr);
          return pack2<T>(to_f32<T>(t[0]) + a0, to_f32<T>(t[1]) + a1);
        };
        auto outputs = [&](auto act_tag) {
          constexpr int ACT = decltype(act_tag)::value;
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) {
            const float4 b0 = *reinterpret_cast<const float4*>(ldsb + G::OFF_BIAS + s2 * 128 + fg * 32), b1 = *reinterpret_cast<const float4*>(ldsb + G::OFF_BIAS + s2 * 128 + fg * 32 + 16);
#pragma unroll
            for (int r = 0; r < RW; ++r)
#pragma unroll
              for (int i = 0; i < FW; ++i) {
                const f32x4 lo = acc[r][i][2 * s2], hi = acc[r][i][2 * s2 + 1];
                const float v0 = activate<T, ACT>(__builtin_fmaf(lo[0], osc, b0.x)), v1 = activate<T, ACT>(__builtin_fmaf(lo[1], osc, b0.y));
                const float v2 = activate<T, ACT>(__builtin_fmaf(lo[2], osc, b0.z)), v3 = activate<T, ACT>(__builtin_fmaf(lo[3], osc, b0.w));
                const float v4 = activate<T, ACT>(__builtin_fmaf(hi[0], osc, b1.x)), v5 = activate<T, ACT>(__builtin_fmaf(hi[1], osc, b1.y));
                const float v6 = activate<T, ACT>(__builtin_fmaf(hi[2], osc, b1.z)), v7 = activate<T, ACT>(__builtin_fmaf(hi[3], osc, b1.w));
                uint4 o;
                if (p.res) {
                  const uint4 q = rv[r][i][s2];
                  o.x = add2(v0, v1, q.x); o.y = add2(v2, v3, q.y); o.z = add2(v4, v5, q.z); o.w = add2(v6, v7, q.w);
                } else {
                  o.x = pack2<T>(v0, v1); o.y = pack2<T>(v2, v3); o.z = pack2<T>(v4, v5); o.w = pack2<T>(v6, v7);
                }
                ov[r][i][s2] = o;
              }
          }
        };
        if (p.act == 1 && !(ABL & 4)) outputs(ICv<1>{}); else outputs(ICv<0>{});
        if constexpr (ABL & 128) ts[1] += __builtin_readcyclecounter() - t0;
        wait_vmcnt<0>();                                             // the next patch (this wave's pieces) and the stores of the tile before
        if constexpr (ABL & 128) ts[2] += __builtin_readcyclecounter() - t0;
#pragma unroll
        for (int r = 0; r < RW; ++r) {
          const int ho = h0 + RW * wg + r;
#pragma unroll
          for (int i = 0; i < FW; ++i) {
            const int wo = w0 + 16 * i + fr;
            if constexpr (ABL & 2) asm volatile("" :: "v"(ov[r][i][0].x), "v"(ov[r][i][0].w), "v"(ov[r][i][1].x), "v"(ov[r][i][1].w));
            else if (ho < p.Ho && wo < p.Wo) {
              T* dst = outp + (((size_t)b * p.Ho + ho) * p.Wo + wo) * (size_t)p.out_cstride + fg * 8;
              *reinterpret_cast<uint4*>(dst) = ov[r][i][0];
              *reinterpret_cast<uint4*>(dst + 32) = ov[r][i][1];
            }
          }
        }
      } else {
        wait_vmcnt<0>();
      }
      if constexpr (ABL & 1024) __builtin_amdgcn_s_setprio(0);
    }
    if constexpr (ABL & 128) {
      const unsigned long long t1 = __builtin_readcyclecounter();
      if ((ph & 1) == grp) tk += t1 - t0; else tf += t1 - t0;
      t0 = t1;
    }
    if (ph + 1 < nph) __syncthreads();                               // patches landed for the whole group / the group's K loop is over: its buffer is free
    if constexpr (ABL & 128) tb += __builtin_readcyclecounter() - t0;
  }
  if constexpr (ABL & 128) {
    if (blockIdx.x == 0 && lane == 0) {
      unsigned long long* dbg = reinterpret_cast<unsigned long long*>(reinterpret_cast<T*>(p.out) + (size_t)p.B * p.Ho * p.Wo * p.out_cstride) + wave * 4;
      dbg[0] = tk; dbg[1] = tf; dbg[2] = tb; dbg[3] = (ts[0] << 40) | (ts[1] << 20) | ts[2];
    }
  }
}

bool conv_tile64_legal(const ConvP& p) {
  return !p.split && p.ks == 3 && p.stride == 1 && p.pad == 1 && p.s1.C == 0 && p.s0.shift == 0 && p.Cin == 64 && p.Cout == 64 && p.s0.C == 64 &&
         p.Hin == p.Ho && p.Win == p.Wo && (!p.res || (!p.res_f32 && p.res_cstride % 8 == 0 && p.res_coff % 8 == 0 && ((uintptr_t)p.res & 15) == 0)) && !p.out_f32 && p.act <= 1 && !p.slope && p.Kw >= 576 && p.Kw % 8 == 0 &&
         p.s0.cstride % 8 == 0 && p.s0.coff % 8 == 0 && p.out_cstride % 8 == 0 && p.out_coff % 8 == 0 &&
         (((uintptr_t)p.s0.ptr | (uintptr_t)p.w | (uintptr_t)p.out) & 15) == 0 &&
         (long)p.B * ((p.Ho + 7) / 8) * ((p.Wo + 31) / 32) < (1L << 22);
}

template <class T, int ABL = 0, int TWIDTH = 32> static void launch_tile64_g(const ConvP& p, hipStream_t stream) {
  using G = Tile64Geom<TWIDTH>;
  constexpr int lds = G::LDS_BYTES;
  static PerDevice pd;
  const int pdi = pd.index();
  if (pd.first(pdi))
    CC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_tile64_kernel<T, ABL, TWIDTH>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  const int cus = pd.cu_count(pdi);
  Tile64Aux a{};
  a.tx = (p.Wo + G::TW - 1) / G::TW; a.tiles = ((p.Ho + G::TH - 1) / G::TH) * a.tx; a.total = p.B * a.tiles;
  a.inv_tiles = 1.0f / (float)a.tiles; a.inv_tx = 1.0f / (float)a.tx;
  const int grid = std::max(8, std::min(cus, (a.total + 1) / 2) & ~7);   // persistent, one block per CU, two tiles or more per block (one per group), the same number of walkers on every XCD
  note_launch("conv3x3_tile64", conv3x3_tile64_kernel<T, ABL, TWIDTH>, (long)a.total, 512, lds, grid);
  hipLaunchKernelGGL((conv3x3_tile64_kernel<T, ABL, TWIDTH>), dim3(grid), dim3(512), lds, stream, p, a);
}

// the geometry with fewer tiles, 16 x 16 on a tie (its patch is 324 pixels against 340: 162 against 173 us at 160 x 160, 44.5 against 48.8 at 80 x 80,
// B = 64, profiles/r06q_tile64.txt); CLEARCAM_TILE64_W=32 / 16 forces one (tests, A/B)
template <class T, int ABL = 0> static void launch_tile64_t(const ConvP& p, hipStream_t stream) {
  static const int env_w = [] { const char* e = getenv("CLEARCAM_TILE64_W"); return e ? atoi(e) : 0; }();
  const int forced = g_tile64_w >= 0 ? g_tile64_w : env_w;      // cc_dev_set("tile64_w", 32 / 16 / -1)
  const long t32 = (long)((p.Ho + 7) / 8) * ((p.Wo + 31) / 32), t16 = (long)((p.Ho + 15) / 16) * ((p.Wo + 15) / 16);
  if constexpr (ABL == 0) {
    if (forced == 16 || (forced != 32 && t16 <= t32)) { launch_tile64_g<T, 0, 16>(p, stream); return; }
  }
  launch_tile64_g<T, ABL, 32>(p, stream);
}

void launch_conv_tile64(int dt, const ConvP& p, hipStream_t stream) {
  CC_CHECK(conv_tile64_legal(p) && dt != F32, "3x3 64 -> 64 tile kernel: shape not eligible");
#ifdef CC_TILE64_ABLATIONS
  if (g_tile64_abl && dt == F16) {
    switch (g_tile64_abl) {
      case 1: launch_tile64_t<f16_t, 1>(p, stream); return;
      case 2: launch_tile64_t<f16_t, 2>(p, stream); return;
      case 4: launch_tile64_t<f16_t, 4>(p, stream); return;
      case 6: launch_tile64_t<f16_t, 6>(p, stream); return;
      case 7: launch_tile64_t<f16_t, 7>(p, stream); return;
      case 8: launch_tile64_t<f16_t, 8>(p, stream); return;
      case 16: launch_tile64_t<f16_t, 16>(p, stream); return;
      case 32: launch_tile64_t<f16_t, 32>(p, stream); return;
      case 33: launch_tile64_t<f16_t, 33>(p, stream); return;
      case 40: launch_tile64_t<f16_t, 40>(p, stream); return;
      case 64: launch_tile64_t<f16_t, 64>(p, stream); return;
      case 128: launch_tile64_t<f16_t, 128>(p, stream); return;
      case 256: launch_tile64_t<f16_t, 256>(p, stream); return;
      case 512: launch_tile64_t<f16_t, 512>(p, stream); return;
      case 768: launch_tile64_t<f16_t, 768>(p, stream); return;
      case 1280: launch_tile64_t<f16_t, 1280>(p, stream); return;
      case 384: launch_tile64_t<f16_t, 384>(p, stream); return;
      case 640: launch_tile64_t<f16_t, 640>(p, stream); return;
      case 160: launch_tile64_t<f16_t, 160>(p, stream); return;
      case 144: launch_tile64_t<f16_t, 144>(p, stream); return;
      case 96: launch_tile64_t<f16_t, 96>(p, stream); return;
      case 104: launch_tile64_t<f16_t, 104>(p, stream); return;
      case 17: launch_tile64_t<f16_t, 17>(p, stream); return;
      case 18: launch_tile64_t<f16_t, 18>(p, stream); return;
      case 20: launch_tile64_t<f16_t, 20>(p, stream); return;
      case 23: launch_tile64_t<f16_t, 23>(p, stream); return;
      default: break;
    }
  }
#endif
  if (dt == F16) launch_tile64_t<f16_t>(p, stream); else launch_tile64_t<bf16_t>(p, stream);
  CC_HIP(hipGetLastError());
}

}  // namespace cc

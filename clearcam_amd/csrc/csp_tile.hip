// RepNCSP (detection/yolov9.py:92-105, one bottleneck) at hidden width 32 as ONE persistent kernel with every weight resident in
// LDS and a 16 x 32-pixel output tile - round 6's rewrite of csp_fused.hip for the two 160 x 160 blocks of the backbone, which the
// old kernel ran at 0.42 ms per launch (208 TFLOP/s, 8x its HBM roof: 128-pixel tiles, weights re-streamed per tile in storage
// mode "f16h", ten barriers per tile with an L2 round trip behind each).
//
//     a = SiLU(cv1 x)         1x1, 64 -> 32      on the 20 x 36 patch (halo 2); zero outside the image (the 3x3's zero padding)
//     t = SiLU(rep3x3 a)      3x3, 32 -> 32      on the 18 x 34 ring (halo 1); zero outside the image
//     u = a + SiLU(3x3 t)     3x3, 32 -> 32      on the 16 x 32 tile
//     b = SiLU(cv2 x)         1x1, 64 -> 32      on the tile
//     out = SiLU(cv3 [u | b]) 1x1, 64 -> 64
//
// What is different from csp_fused.hip:
//  * Weights live in LDS for the life of the block in MFMA-FRAGMENT ORDER: one 1 KB image per (k step, 16-row fragment), lane l's
//    sixteen bytes at l * 16 - an A operand is one ds_read_b128 at a compile-time offset, conflict-free by construction, and two
//    weight planes ("f16h": the 1x1 convs) fit beside the activations (68 KB + 85 KB).  The DMA that builds the images runs once.
//  * MFMA row i of fragment pair (2s, 2s+1) is output channel 32s + (i >> 2) * 8 + h * 4 + (i & 3): after the two fragments a lane holds
//    EIGHT CONSECUTIVE channels of its pixel - one 16-byte chunk, which is exactly the B operand of the next conv's k step.  So
//    u and b go from accumulators straight into cv3 (no LDS round trip), a and t are written with one ds_write_b128 per pixel
//    fragment, and the output leaves as one 16-byte store per lane per 32 channels (four lanes = one 64-byte segment).
//  * x is not staged in LDS: the 1x1 convs read their pixel fragments from global memory in operand layout (16 bytes per lane),
//    prefetched a stage ahead (the next tile's patch during this tile's 3x3 stages).
//  * Three barriers per tile (a complete / t complete / tile done), 2400 MFMAs between them.
// The intermediates use 64-byte pixel rows with the 16-byte chunk index XORed by ((pixel >> 2) & 1) << 1: conflict-free for a
// ds_read_b128 of sixteen consecutive pixels at ANY starting pixel (the taps of a 3x3 shift the window by single pixels).
// Hidden width 64 (208 / 272 KB of weights: they must stream) was built the same way with 16 x 16 tiles and a two-slot 32 KB ring fed
// through registers, one barrier per 24-32 KB chunk: bit-identical and SLOWER than csp_fused.hip's streaming variant (0.30 / 0.40 ms
// against 0.22 / 0.26; 0.19 / 0.25 even with the weight stream switched off: nine barriers per 256-pixel tile, waves parked 44-63 %,
// profiles/r06e_*) - removed; csp_fused.hip keeps hidden width 64.
// K order of every accumulation (tap, plane, 32-channel step) and the rounding points are those of the four launches it replaces:
// bit-identical (tests/test_gpu_yolo.py::test_fused_csp_equals_unfused).
#include <utility>
#include "conv_tile.h"

namespace cc {

namespace {
template <int V> using ICt = std::integral_constant<int, V>;
template <int... I, class F> __device__ __forceinline__ void sfor_impl(std::integer_sequence<int, I...>, F&& f) { (f(ICt<I>{}), ...); }
template <int N, class F> __device__ __forceinline__ void sfor(F&& f) { sfor_impl(std::make_integer_sequence<int, N>{}, f); }
}  // namespace

template <int SPLIT> struct CspTileGeom {
  static constexpr int HID = 32, C2 = 64;
  static constexpr int TH = 16, TW = 32, PH = TH + 4, PW = TW + 4, PR = PH * PW;      // 20 x 36 = 720 patch pixels = 45 fragments
  static constexpr int QH = TH + 2, QW = TW + 2, QR = QH * QW;                          // 18 x 34 = 612 ring pixels
  static constexpr int NF1 = PR / 16, NF2 = (QR + 15) / 16, NF3 = TH * TW / 16;         // 45 / 39 / 32 fragments
  static constexpr int NK1 = SPLIT ? 4 : 2, NK3 = 9;                                    // 32-wide k steps of a 1x1 (hi c0-31, hi c32-63, lo ...) / of a 3x3
  static constexpr int OFF_W12 = 0, OFF_WR = OFF_W12 + NK1 * 4 * 1024, OFF_WB = OFF_WR + NK3 * 2 * 1024, OFF_W3 = OFF_WB + NK3 * 2 * 1024,
                       OFF_BIAS = OFF_W3 + NK1 * 4 * 1024, OFF_A = OFF_BIAS + 1024, OFF_T = OFF_A + PR * 64,
                       LDS_BYTES = (OFF_T + NF2 * 16 * 64 + 2047) & ~2047;
  static_assert(PR % 16 == 0 && LDS_BYTES <= 160 * 1024, "geometry");
};

template <class T, int SPLIT>
__global__ __launch_bounds__(512) void csp_tile_kernel(const CspP p) {
  using G = CspTileGeom<SPLIT>;
  constexpr int PW = G::PW, QW = G::QW, NK1 = G::NK1, NK3 = G::NK3;
  static_assert(sizeof(T) == 2 && (SPLIT == 0 || SPLIT == 2), "16-bit storage; one plane, or two planes in the 1x1 convs");
  const float os12 = SPLIT ? p.os12 : 1.0f, os3 = SPLIT ? p.os3 : 1.0f;           // exact 2^-e output scales of the split 1x1 convs
  extern __shared__ __attribute__((aligned(16))) uint4 lds[];
  char* ldsb = reinterpret_cast<char*>(lds);
  const unsigned lds_base = lds_addr(lds);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int fr = lane & 15, fg = lane >> 4;
  const int total = p.B * p.tiles;

  // ---- tile walk: XCD x owns a contiguous range of tiles (neighbours share halo pixels in its L2) -----------------------------
  int tile, tile_end, tile_step;
  {
    const int nwg = gridDim.x, bid = blockIdx.x, xcd = bid & 7, idx = bid >> 3;
    const int q = total >> 3, r = total & 7;
    const int start = xcd * q + (xcd < r ? xcd : r);
    tile = start + idx; tile_end = start + q + (xcd < r ? 1 : 0); tile_step = nwg >> 3;
  }
  if (tile >= tile_end) return;                                      // (before any barrier: whole blocks only)

  // ---- prologue: weights -> LDS in fragment order, biases -> LDS -------------------------------------------------------------------
  // fragment (k step ks, row fragment j) of a [rows][kw] matrix: lane l fetches row n(j, l & 15), k = 32 ks + 8 (l >> 4) .. + 7
  {
    const int nrow = (fr >> 2) * 8 + (fr & 3);                       // + 32 s + 4 h for fragment j = 2 s + h
    auto issue = [&](const void* wbase, int kw, int ks, int j, unsigned dst) {
      const int n = (j >> 1) * 32 + (j & 1) * 4 + nrow;
      const char* src = reinterpret_cast<const char*>(wbase) + ((size_t)n * kw + ks * 32 + fg * 8) * 2;
      glds16(src, __builtin_amdgcn_readfirstlane(lds_base + dst));
    };
    constexpr int N12 = NK1 * 4, N3x3 = NK3 * 2, NALL = 2 * N12 + 2 * N3x3;
    for (int i = wave; i < NALL; i += 8) {
      if (i < N12) issue(p.w12, p.kw12, i >> 2, i & 3, G::OFF_W12 + i * 1024);
      else if (i < N12 + N3x3) { const int k = i - N12; issue(p.wr, p.kwr, k >> 1, k & 1, G::OFF_WR + k * 1024); }
      else if (i < N12 + 2 * N3x3) { const int k = i - N12 - N3x3; issue(p.wb, p.kwb, k >> 1, k & 1, G::OFF_WB + k * 1024); }
      else { const int k = i - N12 - 2 * N3x3; issue(p.w3, p.kw3, k >> 2, k & 3, G::OFF_W3 + k * 1024); }
    }
    if (tid < 192) {                                                 // b12[64] | br[32] | bb[32] | b3[64] as f32
      const float* src = tid < 64 ? p.b12 + tid : tid < 96 ? p.br + (tid - 64) : tid < 128 ? p.bb + (tid - 96) : p.b3 + (tid - 128);
      reinterpret_cast<float*>(ldsb + G::OFF_BIAS)[tid] = *src;
    }
  }
  constexpr int BI_12 = G::OFF_BIAS, BI_R = BI_12 + 256, BI_B = BI_R + 128, BI_3 = BI_B + 128;

  // ---- helpers ---------------------------------------------------------------------------------------------------------------------
  auto wfrag = [&](int off) { return *reinterpret_cast<const uint4*>(ldsb + off + lane * 16); };
  // byte address of chunk `fg` of pixel q in a 64-byte-per-pixel map: chunk ^= ((q >> 2) & 1) << 1
  auto pix_addr = [&](int region, int q) { const int v = q * 64 + fg * 16; return region + (v ^ ((v >> 3) & 32)); };
  // the 8 consecutive channels a lane holds after fragments (2s, 2s+1): bias + activation -> one 16-byte chunk
  struct Bias8 { float4 b0, b1; };
  auto bias8 = [&](int bias_off) { return Bias8{*reinterpret_cast<const float4*>(ldsb + bias_off + fg * 32), *reinterpret_cast<const float4*>(ldsb + bias_off + fg * 32 + 16)}; };
  auto act8 = [&](const f32x4& lo, const f32x4& hi, float osc, const Bias8& bi) {
    uint4 r;
    r.x = pack2<T>(activate<T, 1>(__builtin_fmaf(lo[0], osc, bi.b0.x)), activate<T, 1>(__builtin_fmaf(lo[1], osc, bi.b0.y)));
    r.y = pack2<T>(activate<T, 1>(__builtin_fmaf(lo[2], osc, bi.b0.z)), activate<T, 1>(__builtin_fmaf(lo[3], osc, bi.b0.w)));
    r.z = pack2<T>(activate<T, 1>(__builtin_fmaf(hi[0], osc, bi.b1.x)), activate<T, 1>(__builtin_fmaf(hi[1], osc, bi.b1.y)));
    r.w = pack2<T>(activate<T, 1>(__builtin_fmaf(hi[2], osc, bi.b1.z)), activate<T, 1>(__builtin_fmaf(hi[3], osc, bi.b1.w)));
    return r;
  };
  const char* xbase = reinterpret_cast<const char*>(p.x) + (size_t)p.x_coff * 2 + fg * 16;
  // the two k-step chunks (channels 8 fg .. + 7 and 32 + 8 fg .. + 7) of input pixel (b, ih, iw), clamped into the image
  auto load_x = [&](int b, int ih, int iw, uint4 (&dst)[2]) {
    const int ch = ih < 0 ? 0 : (ih >= p.H ? p.H - 1 : ih), cw = iw < 0 ? 0 : (iw >= p.W ? p.W - 1 : iw);
    const char* src = xbase + (((size_t)b * p.H + ch) * p.W + cw) * (size_t)p.x_cstride * 2;
    dst[0] = *reinterpret_cast<const uint4*>(src); dst[1] = *reinterpret_cast<const uint4*>(src + 64);
  };
  auto tile_origin = [&](int t, int& b, int& h0, int& w0) {
    b = fdiv(t, p.tiles, p.inv_tiles);
    const int trem = t - b * p.tiles, ty = fdiv(trem, p.tx, p.inv_tx);
    h0 = ty * G::TH; w0 = (trem - ty * p.tx) * G::TW;
  };
  constexpr int F1 = (G::NF1 + 7) / 8, F2 = (G::NF2 + 7) / 8;      // 6 patch / 5 ring fragments per wave at most; 4 tile fragments exactly
  uint4 x1[F1][2];                                                   // stage 1 operands of the tile about to start
  auto load_patch = [&](int t) {
    int b, h0, w0; tile_origin(t, b, h0, w0);
#pragma unroll
    for (int i = 0; i < F1; ++i) {
      const int f = wave + 8 * i;
      if (f < G::NF1) { const int q = 16 * f + fr, py = q / PW, px = q - py * PW; load_x(b, h0 - 2 + py, w0 - 2 + px, x1[i]); }
    }
  };
  load_patch(tile);
  wait_vmcnt<0>();                                                   // weights (LDS-DMA) and the first patch
  __syncthreads();

  for (; tile < tile_end; tile += tile_step) {
    // fr / fg are laundered once per tile: otherwise hipcc hoists every tile-invariant per-lane address (4 + 5 + 6 fragments x 9 taps)
    // out of the tile loop and spills ~100 of them to scratch
    asm volatile("" : "+v"(fr), "+v"(fg));
    int b, h0, w0; tile_origin(tile, b, h0, w0);
    auto in_image = [&](int ih, int iw) { return (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W; };

    // Every stage is software-pipelined BY HAND (operands of step k + 1 requested before the MFMAs of step k, sched_barrier between the
    // blocks): left alone, hipcc keeps one or two ds_reads in flight to save registers and every pair of MFMAs waits a full LDS latency
    // (first build: 0.39 ms per launch, 3x the issue-bound time).
    // ---- stage 1: a = SiLU(cv1 x) on the patch; wave w owns patch fragments w, w + 8, ... ------------------------------------------
    {
      uint4 wv[NK1][2];
#pragma unroll
      for (int ks = 0; ks < NK1; ++ks) { wv[ks][0] = wfrag(G::OFF_W12 + (ks * 4) * 1024); wv[ks][1] = wfrag(G::OFF_W12 + (ks * 4 + 1) * 1024); }
      const Bias8 bi = bias8(BI_12);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < F1; ++i) {
        const int f = wave + 8 * i;
        if (f < G::NF1) {
          f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
          for (int ks = 0; ks < NK1; ++ks) { Mma<T>::run(wv[ks][0], x1[i][ks & 1], acc[0]); Mma<T>::run(wv[ks][1], x1[i][ks & 1], acc[1]); }
          const int q = 16 * f + fr, py = q / PW, px = q - py * PW;
          uint4 v = act8(acc[0], acc[1], os12, bi);
          if (!in_image(h0 - 2 + py, w0 - 2 + px)) v = make_uint4(0u, 0u, 0u, 0u);
          *reinterpret_cast<uint4*>(ldsb + pix_addr(G::OFF_A, q)) = v;
        }
      }
    }
    // operands of cv2 on this wave's four tile fragments (rows 2w, 2w+1; two 16-pixel halves each)
    uint4 x4[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) load_x(b, h0 + 2 * wave + (i >> 1), w0 + 16 * (i & 1) + fr, x4[i]);
    __syncthreads();                                                 // a complete

    // ---- stage 2: t = SiLU(rep3x3 a) on the ring; wave w owns ring fragments w, w + 8, ... -----------------------------------------
    {
      f32x4 acc[F2][2];
      int base[F2];
#pragma unroll
      for (int i = 0; i < F2; ++i) {
        const int f = wave + 8 * i < G::NF2 ? wave + 8 * i : G::NF2 - 1;    // (wave 7's fifth fragment repeats the last one: same values)
        const int q = 16 * f + fr, qc = q < G::QR ? q : G::QR - 1, ty = qc / QW, tx = qc - ty * QW;
        base[i] = ty * PW + tx;                                     // patch pixel of tap (0, 0)
        acc[i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      uint4 xf[2][F2], wf[2][2];
      auto rd = [&](auto tap_c, auto buf_c) {
        constexpr int TAP = decltype(tap_c)::value, BUF = decltype(buf_c)::value, R = TAP / 3, S = TAP - R * 3;
        wf[BUF][0] = wfrag(G::OFF_WR + (TAP * 2) * 1024); wf[BUF][1] = wfrag(G::OFF_WR + (TAP * 2 + 1) * 1024);
#pragma unroll
        for (int i = 0; i < F2; ++i) xf[BUF][i] = *reinterpret_cast<const uint4*>(ldsb + pix_addr(G::OFF_A, base[i] + R * PW + S));
      };
      rd(ICt<0>{}, ICt<0>{});
      const Bias8 bi = bias8(BI_R);
      sfor<NK3>([&](auto tap_c) {
        constexpr int TAP = decltype(tap_c)::value, CUR = TAP & 1;
        if constexpr (TAP + 1 < NK3) rd(ICt<TAP + 1>{}, ICt<CUR ^ 1>{});
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < F2; ++i) { Mma<T>::run(wf[CUR][0], xf[CUR][i], acc[i][0]); Mma<T>::run(wf[CUR][1], xf[CUR][i], acc[i][1]); }
        __builtin_amdgcn_sched_barrier(0);
      });
#pragma unroll
      for (int i = 0; i < F2; ++i) {
        const int f = wave + 8 * i < G::NF2 ? wave + 8 * i : G::NF2 - 1;
        const int q = 16 * f + fr, ty = q / QW, tx = q - ty * QW;
        uint4 v = act8(acc[i][0], acc[i][1], 1.0f, bi);
        if (!in_image(h0 - 1 + ty, w0 - 1 + tx)) v = make_uint4(0u, 0u, 0u, 0u);
        *reinterpret_cast<uint4*>(ldsb + pix_addr(G::OFF_T, q)) = v;       // (pixels 612..623 of the last fragment land in T's padding)
      }
    }
    if (tile + tile_step < tile_end) load_patch(tile + tile_step);   // the next tile's stage 1 operands travel under stages 3 and 4
    __syncthreads();                                                 // t complete

    // ---- stage 3: u = a + SiLU(3x3 t) on this wave's four tile fragments, kept in registers as cv3's first k step --------------------
    uint4 uf[4];
    {
      f32x4 acc[4][2];
      int base[4];
      uint4 av[4];                                                   // the shortcut: a at the fragment's pixels, the same eight channels per lane
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        base[i] = (2 * wave + (i >> 1)) * QW + 16 * (i & 1) + fr;   // ring pixel of tap (0, 0)
        acc[i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        av[i] = *reinterpret_cast<const uint4*>(ldsb + pix_addr(G::OFF_A, (2 * wave + (i >> 1) + 2) * PW + 16 * (i & 1) + fr + 2));
      }
      uint4 xf[2][4], wf[2][2];
      auto rd = [&](auto tap_c, auto buf_c) {
        constexpr int TAP = decltype(tap_c)::value, BUF = decltype(buf_c)::value, R = TAP / 3, S = TAP - R * 3;
        wf[BUF][0] = wfrag(G::OFF_WB + (TAP * 2) * 1024); wf[BUF][1] = wfrag(G::OFF_WB + (TAP * 2 + 1) * 1024);
#pragma unroll
        for (int i = 0; i < 4; ++i) xf[BUF][i] = *reinterpret_cast<const uint4*>(ldsb + pix_addr(G::OFF_T, base[i] + R * QW + S));
      };
      rd(ICt<0>{}, ICt<0>{});
      const Bias8 bi = bias8(BI_B);
      sfor<NK3>([&](auto tap_c) {
        constexpr int TAP = decltype(tap_c)::value, CUR = TAP & 1;
        if constexpr (TAP + 1 < NK3) rd(ICt<TAP + 1>{}, ICt<CUR ^ 1>{});
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i) { Mma<T>::run(wf[CUR][0], xf[CUR][i], acc[i][0]); Mma<T>::run(wf[CUR][1], xf[CUR][i], acc[i][1]); }
        __builtin_amdgcn_sched_barrier(0);
      });
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const T* at = reinterpret_cast<const T*>(&av[i]);
        const f32x4 lo = acc[i][0], hi = acc[i][1];
        uf[i].x = pack2<T>(to_f32<T>(at[0]) + activate<T, 1>(lo[0] + bi.b0.x), to_f32<T>(at[1]) + activate<T, 1>(lo[1] + bi.b0.y));
        uf[i].y = pack2<T>(to_f32<T>(at[2]) + activate<T, 1>(lo[2] + bi.b0.z), to_f32<T>(at[3]) + activate<T, 1>(lo[3] + bi.b0.w));
        uf[i].z = pack2<T>(to_f32<T>(at[4]) + activate<T, 1>(hi[0] + bi.b1.x), to_f32<T>(at[5]) + activate<T, 1>(hi[1] + bi.b1.y));
        uf[i].w = pack2<T>(to_f32<T>(at[6]) + activate<T, 1>(hi[2] + bi.b1.z), to_f32<T>(at[7]) + activate<T, 1>(hi[3] + bi.b1.w));
      }
    }

    // ---- stage 4: b = SiLU(cv2 x), out = SiLU(cv3 [u | b]) -> global, all in registers ------------------------------------------------
    {
      uint4 bf[4];
      {
        uint4 wv[NK1][2];
#pragma unroll
        for (int ks = 0; ks < NK1; ++ks) { wv[ks][0] = wfrag(G::OFF_W12 + (ks * 4 + 2) * 1024); wv[ks][1] = wfrag(G::OFF_W12 + (ks * 4 + 3) * 1024); }
        const Bias8 bi = bias8(BI_12 + 128);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          f32x4 accb[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
          for (int ks = 0; ks < NK1; ++ks) { Mma<T>::run(wv[ks][0], x4[i][ks & 1], accb[0]); Mma<T>::run(wv[ks][1], x4[i][ks & 1], accb[1]); }
          bf[i] = act8(accb[0], accb[1], os12, bi);
        }
      }
      __builtin_amdgcn_sched_barrier(0);                             // cv2's weight fragments are dead before cv3's sixteen are read
      uint4 wv[NK1][4];
#pragma unroll
      for (int ks = 0; ks < NK1; ++ks)
#pragma unroll
        for (int j = 0; j < 4; ++j) wv[ks][j] = wfrag(G::OFF_W3 + (ks * 4 + j) * 1024);
      const Bias8 bi0 = bias8(BI_3), bi1 = bias8(BI_3 + 128);
      __builtin_amdgcn_sched_barrier(0);
      T* outp = reinterpret_cast<T*>(p.out) + p.out_coff + fg * 8;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        f32x4 acc[4] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int ks = 0; ks < NK1; ++ks)
#pragma unroll
          for (int j = 0; j < 4; ++j) Mma<T>::run(wv[ks][j], (ks & 1) ? bf[i] : uf[i], acc[j]);
        const int ho = h0 + 2 * wave + (i >> 1), wo = w0 + 16 * (i & 1) + fr;
        if (ho < p.H && wo < p.W) {
          T* dst = outp + (((size_t)b * p.H + ho) * p.W + wo) * (size_t)p.out_cstride;
          *reinterpret_cast<uint4*>(dst) = act8(acc[0], acc[1], os3, bi0);
          *reinterpret_cast<uint4*>(dst + 32) = act8(acc[2], acc[3], os3, bi1);
        }
      }
    }
    __syncthreads();                                                 // every wave is done with a and t: the next tile may overwrite them
  }
}

bool csp_tile_supported(int dt, int hid, int split) { return (dt == F16 || (dt == BF16 && !split)) && hid == 32 && (split == 0 || split == 2); }

template <class T, int SPLIT> static void launch_csp_tile_t(const CspP& p, hipStream_t stream) {
  constexpr int lds = CspTileGeom<SPLIT>::LDS_BYTES;
  static PerDevice pd;
  const int pdi = pd.index();
  if (pd.first(pdi))
    CC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(csp_tile_kernel<T, SPLIT>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  const int cus = pd.cu_count(pdi);
  const int total = p.B * p.tiles;
  const int grid = std::max(8, std::min(cus, total) & ~7);          // persistent, one block per CU, the same number of walkers on every XCD
  note_launch("csp_tile", csp_tile_kernel<T, SPLIT>, (long)total, 512, lds, grid);
  hipLaunchKernelGGL((csp_tile_kernel<T, SPLIT>), dim3(grid), dim3(512), lds, stream, p);
}

void launch_csp_tile(int dt, const CspP& p0, hipStream_t stream) {
  CC_CHECK(csp_tile_supported(dt, p0.hid, p0.split), "tiled RepNCSP: 16-bit storage, hidden width 32, one weight plane or two in the 1x1 convs");
  CC_CHECK(p0.x_cstride % 8 == 0 && p0.x_coff % 8 == 0 && p0.out_cstride % 8 == 0 && p0.out_coff % 8 == 0 && (((uintptr_t)p0.x | (uintptr_t)p0.out) & 15) == 0,
           "tiled RepNCSP: views must be 16-byte aligned");
  const int s1 = p0.split != 0;
  CC_CHECK(p0.kw12 >= 2 * p0.hid * (1 + s1) && p0.kw3 >= 2 * p0.hid * (1 + s1) && p0.kwr >= 9 * p0.hid && p0.kwb >= 9 * p0.hid, "tiled RepNCSP: weight rows too short");
  CspP p = p0;
  using G = CspTileGeom<0>;
  p.tx = (p.W + G::TW - 1) / G::TW; p.tiles = ((p.H + G::TH - 1) / G::TH) * p.tx;
  p.inv_tiles = 1.0f / (float)p.tiles; p.inv_tx = 1.0f / (float)p.tx;
  CC_CHECK((long)p.B * p.tiles < (1L << 22), "tiled RepNCSP: too many tiles");
  if (p.split == 2) launch_csp_tile_t<f16_t, 2>(p, stream);
  else if (dt == F16) launch_csp_tile_t<f16_t, 0>(p, stream);
  else launch_csp_tile_t<bf16_t, 0>(p, stream);
  CC_HIP(hipGetLastError());
}

}  // namespace cc

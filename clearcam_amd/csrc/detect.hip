// Detector pre/post-processing kernels.
//   preprocess : letterbox (tinygrad interpolate/lerp semantics, incl. the uint8 7-bit fixed point),
//                zero pad, BGR->RGB, /255            detection/yolov9.py:376-379,390-404; helpers.py:127-131
//   decode     : DFL softmax-expectation, dist2bbox, *stride, sigmoid, xywh->xyxy, class max/argmax,
//                confidence threshold                detection/yolov9.py:209-220,263-282,440-448
//   topk_nms   : stable top-300 by score (radix select + bitonic sort), 300x300 mask NMS,
//                scale_boxes/clip_boxes              detection/yolov9.py:406-458
// Integer/byte work (the uint8 resize) is bit-exact against the oracle; float work follows the
// reference's operation order in f32.
#include <type_traits>
#include "kernels.h"
#include "mfma.h"

namespace cc {

// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int lerp_u8(int a, int b, int w7) {
  // tinygrad Tensor.lerp for uint8: int8-wrapped difference, 7-bit weight, +64, >>7, mod 256.
  const int d = (int)(int8_t)(uint8_t)(b - a);
  const int t = (int)(((uint16_t)(int16_t)(__mul24(d, w7) + 64)) >> 7);    // |d| < 2^7, |w7| <= 2^7: a 24-bit multiply (full rate)
  return (a + t) & 0xff;
}

// One letterboxed pixel from its interpolation taps: rows y0/y1, columns x0/x1, fractions fx/fy (tinygrad interpolate,
// uint8 frames through the 7-bit fixed-point lerp).  `u8(row, col, ch)` / `f32(row, col, ch)` fetch a source sample.
template <class FetchU8, class FetchF32>
__device__ __forceinline__ void letterbox_taps(const PreP& p, int x0, int x1, int y0, int y1, float fx, float fy, FetchU8 u8, FetchF32 f32, float (&rgb)[3]) {
  if (!p.frame_f32) {
    const int wx = (int)(int16_t)__fadd_rn(__fmul_rn(fx, 128.0f), 0.5f);
    const int wy = (int)(int16_t)__fadd_rn(__fmul_rn(fy, 128.0f), 0.5f);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int top = lerp_u8(u8(y0, x0, c), u8(y0, x1, c), wx);
      const int bot = lerp_u8(u8(y1, x0, c), u8(y1, x1, c), wx);
      rgb[p.flip ? 2 - c : c] = __fsub_rn(__fdiv_rn((float)lerp_u8(top, bot, wy), p.div), p.sub);
    }
  } else {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float a0 = f32(y0, x0, c), b0 = f32(y0, x1, c);
      const float a1 = f32(y1, x0, c), b1 = f32(y1, x1, c);
      const float top = __fadd_rn(a0, __fmul_rn(__fsub_rn(b0, a0), fx));
      const float bot = __fadd_rn(a1, __fmul_rn(__fsub_rn(b1, a1), fx));
      rgb[p.flip ? 2 - c : c] = __fsub_rn(__fdiv_rn(__fadd_rn(top, __fmul_rn(__fsub_rn(bot, top), fy)), p.div), p.sub);
    }
  }
}

// One pixel (b, y, x) of the letterboxed network input, as float RGB-or-BGR after v / div - sub (pad_val in the padding).
__device__ __forceinline__ void letterbox_rgb(const PreP& p, int b, int y, int x, float (&rgb)[3]) {
  rgb[0] = rgb[1] = rgb[2] = p.pad_val;
  const int yy = y - p.pad_y, xx = x - p.pad_x;
  if ((unsigned)yy < (unsigned)p.nh && (unsigned)xx < (unsigned)p.nw) {
    const size_t img = (size_t)b * p.H;
    auto u8 = [&](int r, int c, int ch) { return (int)reinterpret_cast<const uint8_t*>(p.frames)[((img + r) * p.W + c) * 3 + ch]; };
    auto f32 = [&](int r, int c, int ch) { return reinterpret_cast<const float*>(p.frames)[((img + r) * p.W + c) * 3 + ch]; };
    letterbox_taps(p, p.xlo[xx], p.xhi[xx], p.ylo[yy], p.yhi[yy], p.xfr[xx], p.yfr[yy], u8, f32, rgb);
  }
}

template <class T>
__global__ __launch_bounds__(256) void preprocess_kernel(const PreP p) {
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, b = blockIdx.z;      // grid = (row tiles, row, image)
  if (x >= p.Wn) return;
  const size_t idx = ((size_t)b * p.Hn + y) * p.Wn + x;
  float rgb[3];
  letterbox_rgb(p, b, y, x, rgb);
  T* o = reinterpret_cast<T*>(p.out) + idx * p.out_c;
  if (p.out_c * (int)sizeof(T) == 16) {                         // the stem's one 16-byte channel chunk: a single store
    uint4 v = make_uint4(0, 0, 0, 0);
    T* t = reinterpret_cast<T*>(&v);
#pragma unroll
    for (int c = 0; c < 3; ++c) t[c] = from_f32<T>(rgb[c]);
    *reinterpret_cast<uint4*>(o) = v;
  } else {
    for (int c = 0; c < p.out_c; ++c) o[c] = from_f32<T>(c < 3 ? rgb[c] : 0.f);
  }
}

// ---- letterbox + first conv ---------------------------------------------------------------------------------------
// Conv 3x3 stride 2 pad 1, 3 -> COUT, bias, SiLU (detection/yolov9.py:302 / :330, Conv :33-38) computed straight from the
// camera frames.  A block owns a 16x16 tile of output pixels: it letterboxes the 33x33 input pixels the tile needs into
// LDS (rounded to the storage type exactly as preprocess_kernel stores them; zeros outside the network input = the
// conv's padding), then every output pixel is ONE v_mfma_f32_16x16x32 per 16 output channels: K = 27 (row r, column s,
// channel c) - 9 contiguous LDS elements per row r - laid into the 32 k slots so that a lane reads whole dwords (see the
// B-operand comment below).  The result goes through LDS so each pixel's COUT channels leave as 16-byte stores.  HBM traffic: the frames in, the activations out;
// the (B,Hn,Wn,8) input tensor of the unfused path (a 16-byte write and read per pixel) does not exist.
// BIG: 36 KB of scratch for the source-byte stage (camera frames scaled down by up to ~3x: 1080p -> 384x640); otherwise 18 KB, which
// is what the output stage needs (each wave passes its 64 pixels through its own 32-pixel area in two halves) and lets six blocks
// share a CU instead of three.
// SPLIT: the weights are two f16 planes (ConvP::split): a second MFMA per channel fragment on the low plane, the accumulator scaled by the
// exact 2^-e in the epilogue.
template <class T, int COUT, bool BIG, bool SPLIT = false>
__global__ __launch_bounds__(256) void stem_fused_kernel(const StemP p) {
  constexpr int PW = 33, PROW = 104;                     // patch row: 33 pixels x 3 channels (+ pad) in storage type
  constexpr int NT = COUT / 16, OROW = COUT + 8;         // staged output row: COUT channels + 16 bytes (bank spread)
  constexpr int SCRATCH = (BIG ? 256 : 128) * 72 * 2;    // bytes: the source-pixel stage (phase 1) and the output stage (phase 3) share it
  __shared__ T patch[PW * PROW];
  __shared__ __attribute__((aligned(16))) unsigned char scratch[SCRATCH];
  __shared__ int tab_i[4][PW];                           // xlo, xhi, ylo, yhi of the patch's columns / rows
  __shared__ float tab_f[2][PW];                         // xfr, yfr
  __shared__ int tab_w[2][PW];                           // the 7-bit fixed-point weights of xfr, yfr (uint8 frames)
  __shared__ T lut[256];                                 // uint8 frames: value -> value / div - sub, rounded to T
  T* ostage = reinterpret_cast<T*>(scratch);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ox0 = blockIdx.x * 16, oy0 = blockIdx.y * 16, b = blockIdx.z;

  // weights: A operand, row = output channel, this lane's eight k values = one 16-byte load; kept in registers for the tile
  const int kg = lane >> 4, row = lane & 15;
  uint4 afrag[NT], alo[SPLIT ? NT : 1];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) afrag[nt] = reinterpret_cast<const uint4*>(p.w)[(nt * 16 + row) * 4 + kg];
  if constexpr (SPLIT) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) alo[nt] = reinterpret_cast<const uint4*>(p.w_lo)[(nt * 16 + row) * 4 + kg];
  }
  const float osc = SPLIT ? p.oscale : 1.0f;

  // ---- phase 1: the 33x33 letterboxed pixels of this tile -> patch ------------------------------------------------
  // Per-pixel byte loads from HBM are what bounds the stand-alone letterbox kernel (12 load instructions per pixel), so the
  // tile's interpolation tables and - for uint8 frames whose source rectangle fits - the source bytes themselves are
  // staged in LDS with aligned dword loads first; the taps are then LDS reads.  Arithmetic is letterbox_taps either way.
  const PreP& q = p.pre;
  const int Y0 = 2 * oy0 - 1, X0 = 2 * ox0 - 1;
  if (tid < 2 * PW) {
    const bool isx = tid < PW;
    const int j = isx ? tid : tid - PW;
    const int v = (isx ? X0 - q.pad_x : Y0 - q.pad_y) + j, n = isx ? q.nw : q.nh;
    const bool ok = (unsigned)v < (unsigned)n;
    const int vc = min(max(v, 0), n - 1);
    tab_i[isx ? 0 : 2][j] = ok ? (isx ? q.xlo : q.ylo)[vc] : -1;          // -1: this column / row is padding
    tab_i[isx ? 1 : 3][j] = ok ? (isx ? q.xhi : q.yhi)[vc] : -1;
    const float fr = ok ? (isx ? q.xfr : q.yfr)[vc] : 0.f;
    tab_f[isx ? 0 : 1][j] = fr;
    tab_w[isx ? 0 : 1][j] = (int)(int16_t)__fadd_rn(__fmul_rn(fr, 128.0f), 0.5f);       // as letterbox_taps
  }
  lut[tid] = from_f32<T>(__fsub_rn(__fdiv_rn((float)tid, q.div), q.sub));
  // source rectangle of the tile (tables are monotone): rows [sr0, sr1], columns [sc0, sc1]
  const int yy0 = max(Y0 - q.pad_y, 0), yy1 = min(Y0 + PW - 1 - q.pad_y, q.nh - 1);
  const int xx0 = max(X0 - q.pad_x, 0), xx1 = min(X0 + PW - 1 - q.pad_x, q.nw - 1);
  const bool any = yy0 <= yy1 && xx0 <= xx1;
  const int sr0 = any ? q.ylo[yy0] : 0, sr1 = any ? q.yhi[yy1] : -1, sc0 = any ? q.xlo[xx0] : 0, sc1 = any ? q.xhi[xx1] : -1;
  const int nrows = sr1 - sr0 + 1, rowbytes = (sc1 - sc0 + 1) * 3;
  const int dpr = (rowbytes + 6) / 4;                                   // dwords per staged row, whatever the row's misalignment
  const bool staged = !q.frame_f32 && any && nrows * dpr * 4 <= SCRATCH;
  if (staged && !(p.abl & 4)) {
    const uintptr_t base = reinterpret_cast<uintptr_t>(q.frames);
    const size_t total = (size_t)q.B * q.H * q.W * 3;
    // four independent dword loads per thread in flight (unconditional, index clamped), then the LDS writes
    const int n = nrows * dpr;
    const float inv_dpr = 1.0f / (float)dpr;
    for (int i0 = tid; i0 < n; i0 += 1024) {
      unsigned v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = min(i0 + 256 * u, n - 1);
        int r = (int)((float)i * inv_dpr);                               // i / dpr without an integer divide
        r += (r + 1) * dpr <= i ? 1 : 0; r -= r * dpr > i ? 1 : 0;
        const int d = i - r * dpr;
        // absolute addresses: an aligned dword never straddles a page, so the words holding the buffer's first and last
        // byte are readable whatever the alignment of the caller's pointer or the size of the buffer
        const uintptr_t a0 = base + (((size_t)b * q.H + sr0 + r) * q.W + sc0) * 3;   // first byte this row needs
        const uintptr_t al = min((a0 & ~(uintptr_t)3) + 4 * (uintptr_t)d, (base + total - 1) & ~(uintptr_t)3);
        v[u] = *reinterpret_cast<const unsigned*>(al);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) if (i0 + 256 * u < n) reinterpret_cast<unsigned*>(scratch)[i0 + 256 * u] = v[u];
    }
  }
  __syncthreads();
  // thread -> (column tid & 31, rows tid >> 5, +8, ...): whatever depends on the column only is read once; column 32 of the
  // patch is done by the first 33 threads afterwards
  const T zero = from_f32<T>(0.f), padv = from_f32<T>(q.pad_val);
  const size_t img = (size_t)b * q.H;
  const unsigned fr_lo = (unsigned)reinterpret_cast<uintptr_t>(q.frames);
  auto pixel = [&](int pr, int pc, int x0, int x1, int wx, float fx) {
    T o[3] = {zero, zero, zero};                                        // outside the network input: the conv's zero padding
    const int y = Y0 + pr, x = X0 + pc;
    if ((unsigned)y < (unsigned)q.Hn && (unsigned)x < (unsigned)q.Wn) {
      const int y0 = tab_i[2][pr];
      if (x0 < 0 || y0 < 0) o[0] = o[1] = o[2] = padv;
      else {
        const int y1 = tab_i[3][pr];
        if (!q.frame_f32) {
          // uint8 frames: the interpolated value is an integer 0..255, so v / div - sub rounded to T is a 256-entry table
          const int wy = tab_w[1][pr];
          int v[3];
          if (staged) {
            auto rowoff = [&](int r) {                                   // LDS byte offset of source column sc0 in row r
              const unsigned sh = (fr_lo + ((unsigned)(b * q.H + r) * (unsigned)q.W + (unsigned)sc0) * 3u) & 3u;
              return (r - sr0) * dpr * 4 + (int)sh;                      // the row's first byte, modulo its aligned dword
            };
            const int r0 = rowoff(y0), r1 = rowoff(y1), c0 = (x0 - sc0) * 3, c1 = (x1 - sc0) * 3;
            if ((wx | wy) == 0) {                                        // both weights zero (frames already at the network's size: the letterbox is
#pragma unroll                                                           // the identity): lerp_u8(a, b, 0) IS a - one byte per channel instead of four
              for (int c = 0; c < 3; ++c) v[c] = scratch[r0 + c0 + c];
            } else {
#pragma unroll
            for (int c = 0; c < 3; ++c)
              v[c] = lerp_u8(lerp_u8(scratch[r0 + c0 + c], scratch[r0 + c1 + c], wx), lerp_u8(scratch[r1 + c0 + c], scratch[r1 + c1 + c], wx), wy);
            }
          } else {
            const uint8_t* f = reinterpret_cast<const uint8_t*>(q.frames);
            const size_t a00 = ((img + y0) * q.W + x0) * 3, a01 = ((img + y0) * q.W + x1) * 3, a10 = ((img + y1) * q.W + x0) * 3, a11 = ((img + y1) * q.W + x1) * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) v[c] = lerp_u8(lerp_u8(f[a00 + c], f[a01 + c], wx), lerp_u8(f[a10 + c], f[a11 + c], wx), wy);
          }
#pragma unroll
          for (int c = 0; c < 3; ++c) o[q.flip ? 2 - c : c] = lut[v[c]];
        } else {
          float rgb[3];
          auto g8 = [&](int, int, int) { return 0; };
          auto gf = [&](int r, int c, int ch) { return reinterpret_cast<const float*>(q.frames)[((img + r) * q.W + c) * 3 + ch]; };
          letterbox_taps(q, x0, x1, y0, y1, fx, tab_f[1][pr], g8, gf, rgb);
          o[0] = from_f32<T>(rgb[0]); o[1] = from_f32<T>(rgb[1]); o[2] = from_f32<T>(rgb[2]);
        }
      }
    }
    T* d = patch + pr * PROW + pc * 3;
    d[0] = o[0]; d[1] = o[1]; d[2] = o[2];
  };
  if (!(p.abl & 1)) {
    {
      const int pc = tid & 31;
      const int x0 = tab_i[0][pc], x1 = tab_i[1][pc], wx = tab_w[0][pc];
      const float fx = tab_f[0][pc];
#pragma unroll
      for (int pass = 0; pass < 5; ++pass) {
        const int pr = (tid >> 5) + 8 * pass;
        if (pr < PW) pixel(pr, pc, x0, x1, wx, fx);
      }
    }
    if (tid < PW) pixel(tid, 32, tab_i[0][32], tab_i[1][32], tab_w[0][32], tab_f[0][32]);
  }
  __syncthreads();

  float bias[NT][4];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int i = 0; i < 4; ++i) bias[nt][i] = p.bias[nt * 16 + kg * 4 + i];

  // B operand (pixels).  k slots: lane groups 0..2 take row r = group, the first 8 of that row's 9 contiguous (s, c)
  // elements (four aligned dwords); group 3 takes element 8 of rows 0..2 and five zero slots.  stem_pack_weights uses the
  // same order.  Every lane issues the same four ds_read_b32; group 3 masks and repacks.
  const unsigned* patch32 = reinterpret_cast<const unsigned*>(patch);
  int boff[4];                                                        // dword offsets relative to the pixel's first element
#pragma unroll
  for (int j = 0; j < 4; ++j) boff[j] = kg < 3 ? kg * (PROW / 2) + j : min(j, 2) * (PROW / 2) + 4;
  // A wave owns tile rows 4*wave .. 4*wave+3 and passes them, two rows at a time, through its own 32-pixel staging area so that every
  // pixel's COUT channels leave as 16-byte stores: same-wave LDS traffic only, no block barrier after the patch is complete.
  T* wstage = ostage + wave * (32 * OROW);
  constexpr int CPP = COUT / 8;                                     // 16-byte chunks per pixel
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const int ty = wave * 4 + rr, tx = row;                         // this lane's pixel of the 16-pixel row
    const int base = (2 * ty) * (PROW / 2) + 3 * tx;                // element 2*tx*3 of patch row 2*ty, in dwords
    const unsigned d0 = patch32[base + boff[0]], d1 = patch32[base + boff[1]], d2 = patch32[base + boff[2]], d3 = patch32[base + boff[3]];
    uint4 bfrag;
    bfrag.x = kg < 3 ? d0 : ((d0 & 0xffffu) | (d1 << 16));
    bfrag.y = kg < 3 ? d1 : (d2 & 0xffffu);
    bfrag.z = kg < 3 ? d2 : 0u;
    bfrag.w = kg < 3 ? d3 : 0u;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      Mma<T>::run(afrag[nt], bfrag, acc);
      if constexpr (SPLIT) Mma<T>::run(alo[nt], bfrag, acc);
      float o[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float xv = __builtin_fmaf(acc[i], osc, bias[nt][i]);
        o[i] = (p.abl & 2) ? xv : xv * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(xv * -1.4426950408889634f));   // SiLU, as conv_mfma's 16-bit epilogue
      }
      *reinterpret_cast<uint2*>(wstage + ((rr & 1) * 16 + tx) * OROW + nt * 16 + kg * 4) = make_uint2(pack2<T>(o[0], o[1]), pack2<T>(o[2], o[3]));
    }
    if (rr & 1) {                                                   // two rows staged: store them
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int j = lane; j < 32 * CPP; j += 64) {
        const int px = j / CPP, ch = j - px * CPP;
        const int oy = oy0 + wave * 4 + (rr - 1) + (px >> 4), ox = ox0 + (px & 15);
        if (oy < p.Ho && ox < p.Wo)
          *reinterpret_cast<uint4*>(reinterpret_cast<T*>(p.out) + (((size_t)b * p.Ho + oy) * p.Wo + ox) * p.out_cstride + p.out_coff + ch * 8) =
              *reinterpret_cast<const uint4*>(wstage + px * OROW + ch * 8);
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
}

template <class T>
__global__ void stem_pack_kernel(const T* __restrict__ w, int w_row, int tap_stride, int plane_off, int n, T* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n * 32) return;
  // k slot -> (row r, element q = s*3 + c): slots 0..23 = rows 0..2 x elements 0..7, slots 24..26 = element 8 of rows 0..2
  const int co = i >> 5, k = i & 31, g = k >> 3, e = k & 7;
  const bool live = g < 3 || e < 3;
  const int r = g < 3 ? g : min(e, 2), q = g < 3 ? e : 8, s = q / 3, c = q - s * 3;
  out[i] = live ? w[(size_t)co * w_row + (r * 3 + s) * tap_stride + plane_off + c] : from_f32<T>(0.f);
}

void stem_pack_weights(int dt, const void* w_packed, int w_row, int tap_stride, int plane_off, int Cout, void* out, hipStream_t stream) {
  const dim3 grid((Cout * 32 + 255) / 256), block(256);
  if (dt == F16) hipLaunchKernelGGL(stem_pack_kernel<f16_t>, grid, block, 0, stream, (const f16_t*)w_packed, w_row, tap_stride, plane_off, Cout, (f16_t*)out);
  else hipLaunchKernelGGL(stem_pack_kernel<bf16_t>, grid, block, 0, stream, (const bf16_t*)w_packed, w_row, tap_stride, plane_off, Cout, (bf16_t*)out);
  CC_HIP(hipGetLastError());
}

bool stem_fused_supported(int dt, int Cout) { return dt != F32 && (Cout == 16 || Cout == 32 || Cout == 64); }

template <class T, bool BIG> static void launch_stem_b(const StemP& p, hipStream_t stream) {
  const dim3 grid((unsigned)((p.Wo + 15) / 16), (unsigned)((p.Ho + 15) / 16), (unsigned)p.pre.B), block(256);
  if constexpr (std::is_same<T, f16_t>::value) {
    if (p.w_lo) {                                                     // split weights (f16 storage only)
      if (p.Cout == 64) hipLaunchKernelGGL((stem_fused_kernel<T, 64, BIG, true>), grid, block, 0, stream, p);
      else if (p.Cout == 32) hipLaunchKernelGGL((stem_fused_kernel<T, 32, BIG, true>), grid, block, 0, stream, p);
      else hipLaunchKernelGGL((stem_fused_kernel<T, 16, BIG, true>), grid, block, 0, stream, p);
      return;
    }
  }
  if (p.Cout == 64) hipLaunchKernelGGL((stem_fused_kernel<T, 64, BIG>), grid, block, 0, stream, p);
  else if (p.Cout == 32) hipLaunchKernelGGL((stem_fused_kernel<T, 32, BIG>), grid, block, 0, stream, p);
  else hipLaunchKernelGGL((stem_fused_kernel<T, 16, BIG>), grid, block, 0, stream, p);
}
template <class T> static void launch_stem_t(const StemP& p, hipStream_t stream) {
  // bytes the source rectangle of a 33x33-pixel patch can take in the stage (the kernel checks the exact figure per tile and reads
  // the frame directly where it does not fit): small scratch, twice the blocks per CU, when that is at most 18 KB
  const PreP& q = p.pre;
  const long nr = (33L * q.H + q.nh - 1) / q.nh + 2, nc = (33L * q.W + q.nw - 1) / q.nw + 2;
  const bool big = !q.frame_f32 && nr * ((nc * 3 + 6) / 4) * 4 > 128 * 72 * 2;
  if (big) launch_stem_b<T, true>(p, stream); else launch_stem_b<T, false>(p, stream);
}

void launch_stem_fused(int dt, const StemP& p0, hipStream_t stream) {
  StemP p = p0;
  { static const int abl = [] { const char* e = getenv("CLEARCAM_STEM_ABL"); return e ? atoi(e) : 0; }(); p.abl = abl; }
  CC_CHECK(stem_fused_supported(dt, p.Cout) && p.out_coff % 8 == 0 && p.out_cstride % 8 == 0 && (!p.w_lo || dt == F16), "fused stem: unsupported dtype / channel count");
  if (dt == F16) launch_stem_t<f16_t>(p, stream); else launch_stem_t<bf16_t>(p, stream);
  CC_HIP(hipGetLastError());
}

void launch_preprocess(int dt, const PreP& p, hipStream_t stream) {
  const dim3 grid((unsigned)((p.Wn + 255) / 256), (unsigned)p.Hn, (unsigned)p.B), block(256);
  if (dt == F32) hipLaunchKernelGGL(preprocess_kernel<float>, grid, block, 0, stream, p);
  else if (dt == F16) hipLaunchKernelGGL(preprocess_kernel<f16_t>, grid, block, 0, stream, p);
  else hipLaunchKernelGGL(preprocess_kernel<bf16_t>, grid, block, 0, stream, p);
  CC_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------
// decode arithmetic, shared by decode_kernel and the fused head_tail_kernel (one definition: the same f32 operations in the same
// order, so the two paths produce the same bits)
// DFL (detection/yolov9.py:273-282): softmax over the 16 bins of one box side, then the 1x1 conv with the loaded 16 weights
__device__ __forceinline__ float dfl_expect(float (&v)[16], const float (&w16)[16]) {
  float mx = v[0];
#pragma unroll
  for (int i = 1; i < 16; ++i) mx = fmaxf(mx, v[i]);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) { v[i] = expf(v[i] - mx); s += v[i]; }
  float e = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) e = __fadd_rn(e, __fmul_rn(__fdiv_rn(v[i], s), w16[i]));
  return e;
}
// dist2bbox(xywh) * stride -> xyxy (:263-271, :440-441) for anchor cell (a % W, a / W)
__device__ __forceinline__ void dist_to_box(const float (&d)[4], int a, int W, float stride, float (&o)[4]) {
  const float ax = (float)(a % W) + 0.5f, ay = (float)(a / W) + 0.5f;
  const float lx = ax - d[0], ly = ay - d[1], rx = ax + d[2], ry = ay + d[3];
  const float cx = ((lx + rx) / 2.f) * stride, cy = ((ly + ry) / 2.f) * stride;
  const float bw = (rx - lx) * stride, bh = (ry - ly) * stride;
  o[0] = cx - bw / 2.f; o[1] = cy - bh / 2.f; o[2] = cx + bw / 2.f; o[3] = cy + bh / 2.f;
}
__device__ __forceinline__ float class_sigmoid(float l) { return 1.0f / (1.0f + expf(-l)); }

__global__ __launch_bounds__(256) void decode_kernel(const DecodeP p) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)p.B * p.A) return;
  const int b = (int)(idx / p.A);
  int a = (int)(idx - (size_t)b * p.A);
  int lvl = 0;
  while (lvl < 2 && a >= p.H[lvl] * p.W[lvl]) { a -= p.H[lvl] * p.W[lvl]; ++lvl; }
  const int H = p.H[lvl], W = p.W[lvl];
  const float stride = lvl == 0 ? 8.f : (lvl == 1 ? 16.f : 32.f);
  const float4* r = reinterpret_cast<const float4*>(p.raw[lvl] + ((size_t)b * H * W + a) * 144);
  float w16[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) w16[i] = p.dfl_w[i];
  float d[4];
#pragma unroll
  for (int side = 0; side < 4; ++side) {
    float v[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) { const float4 t = r[side * 4 + q]; v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w; }
    d[side] = dfl_expect(v, w16);
  }
  float box[4];
  dist_to_box(d, a, W, stride, box);
  float best = -1.f; int bi = 0;
#pragma unroll 4
  for (int q = 0; q < 20; ++q) {
    const float4 t = r[16 + q];
    const float l[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float sg = class_sigmoid(l[e]);
      if (sg > best) { best = sg; bi = q * 4 + e; }
    }
  }
  float* o = p.det + idx * 6;
  o[0] = box[0]; o[1] = box[1]; o[2] = box[2]; o[3] = box[3];
  o[4] = best >= p.conf ? best : 0.f;
  o[5] = (float)bi;
  // every finite logit has a sigmoid above -1: `best` still at its initial value, or a box that is not a number, means the activations
  // left the storage type's range somewhere upstream (f16 saturates at 65504) - counted, so that the caller is told instead of getting no detections
  if (p.nonfinite && (!(best >= 0.f) || !(box[0] - box[0] == 0.f) || !(box[2] - box[2] == 0.f))) atomicAdd(p.nonfinite, 1);
}

// ---- DDetect tail: last 1x1 convs of both branches + decode, logits on chip ---------------------------------------------------
// One block = four waves; a wave owns 16 pixels at a time (one MFMA pixel fragment) and walks TPB tiles of 64 pixels with the
// block.  Weights of the level (box 64 x 64, class 80 x CH) sit in LDS for the block's lifetime (rows padded by 16 bytes: fragment
// reads spread over the banks); pixel fragments come straight from global memory into registers (each is used by one wave only).
//   box branch:   4 channel fragments x K 64      class branch: 5 channel fragments x K CH     (ascending K, one accumulator each:
//   the accumulation order of every conv kernel in this library, so the logits equal the two conv launches' bit for bit)
//   class scores: sigmoid + running (max, argmax) over the lane's 20 logits in registers, then over the four lanes that share a
//                 pixel (ties -> lower class index = decode_kernel's first-maximum scan)
//   box:          the 64 box logits of a pixel go through LDS so that lane (pixel, side) runs decode_kernel's own 16-bin sequence
constexpr int kTailTPB = 8;            // 64-pixel tiles per block
// SPLIT: weight rows are [hi(Cin) | lo(Cin)] (ConvP::split): the K walk passes the pixel fragments twice, high plane first - the order of
// the conv kernels' virtual taps - and the accumulator is scaled by the level's exact 2^-e before the bias.
template <class T, int CH, bool SPLIT = false>
__global__ __launch_bounds__(256) void head_tail_kernel(const HeadTailP p, int blk1, int blk2) {
  constexpr int PL = SPLIT ? 2 : 1;
  constexpr int KC3 = CH / 32, ROW3 = PL * CH * 2 + 16, ROW2 = PL * 64 * 2 + 16, LROW = 68;      // LDS row strides: bytes, bytes, floats
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* w3s = smem;                                    // 80 rows x ROW3
  char* w2s = smem + 80 * ROW3;                        // 64 rows x ROW2
  float* lg = reinterpret_cast<float*>(smem + 80 * ROW3 + 64 * ROW2);   // [4 waves][16 pixels][LROW]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fg = lane >> 4;
  const int lvl = (int)blockIdx.x >= blk2 ? 2 : ((int)blockIdx.x >= blk1 ? 1 : 0);
  const int blk = (int)blockIdx.x - (lvl == 2 ? blk2 : (lvl == 1 ? blk1 : 0));
  const int H = p.H[lvl], W = p.W[lvl], hw = H * W, M = p.B * hw;
  const float stride = lvl == 0 ? 8.f : (lvl == 1 ? 16.f : 32.f);
  int aoff = 0;
  for (int l = 0; l < lvl; ++l) aoff += p.H[l] * p.W[l];
  // weights -> LDS (16-byte chunks; rows are kw elements apart in global memory)
  {
    const char* g3 = reinterpret_cast<const char*>(p.w3[lvl]);
    for (int c = tid; c < 80 * (PL * CH / 8); c += 256) {
      const int row = c / (PL * CH / 8), ch = c - row * (PL * CH / 8);
      *reinterpret_cast<uint4*>(w3s + row * ROW3 + ch * 16) = *reinterpret_cast<const uint4*>(g3 + ((size_t)row * p.kw3 + ch * 8) * sizeof(T));
    }
    const char* g2 = reinterpret_cast<const char*>(p.w2[lvl]);
    for (int c = tid; c < 64 * 8 * PL; c += 256) {
      const int row = c / (8 * PL), ch = c - row * (8 * PL);
      *reinterpret_cast<uint4*>(w2s + row * ROW2 + ch * 16) = *reinterpret_cast<const uint4*>(g2 + ((size_t)row * p.kw2 + ch * 8) * sizeof(T));
    }
  }
  float w16[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) w16[i] = p.dfl_w[i];
  float4 b2v[4], b3v[5];
#pragma unroll
  for (int j = 0; j < 4; ++j) b2v[j] = *reinterpret_cast<const float4*>(p.b2[lvl] + j * 16 + fg * 4);
#pragma unroll
  for (int j = 0; j < 5; ++j) b3v[j] = *reinterpret_cast<const float4*>(p.b3[lvl] + j * 16 + fg * 4);
  const float os2 = SPLIT ? p.os2[lvl] : 1.0f, os3 = SPLIT ? p.os3[lvl] : 1.0f;
  __syncthreads();
  const T* bxp = reinterpret_cast<const T*>(p.bx[lvl]);
  const T* clp = reinterpret_cast<const T*>(p.cl[lvl]);
  float* lgw = lg + wave * 16 * LROW;
  // pixel fragments of tile t+1 are requested before tile t is computed and decoded (the decode is ~4 k VALU cycles per tile, an
  // HBM round trip about as long: without the prefetch every tile opened with a dependent load and nothing to do)
  auto load_frags = [&](int t, uint4 (&xb)[2], uint4 (&xc)[KC3]) {
    const int m0 = (blk * kTailTPB + t) * 64 + wave * 16;
    const int mc = min(m0 + fr, M - 1);                // past the last pixel (ragged fragment, or no tile t at all): clamped, never stored
#pragma unroll
    for (int kc = 0; kc < 2; ++kc) xb[kc] = *reinterpret_cast<const uint4*>(bxp + (size_t)mc * 64 + (kc * 4 + fg) * 8);
#pragma unroll
    for (int kc = 0; kc < KC3; ++kc) xc[kc] = *reinterpret_cast<const uint4*>(clp + (size_t)mc * CH + (kc * 4 + fg) * 8);
  };
  uint4 xb[2], xc[KC3], nb[2], nc[KC3];
  load_frags(0, xb, xc);
  for (int t = 0; t < kTailTPB; ++t) {
    const int m0 = (blk * kTailTPB + t) * 64 + wave * 16;
    if (m0 >= M) break;                                // wave-uniform
    const int m = m0 + fr;
    if (t + 1 < kTailTPB) load_frags(t + 1, nb, nc);
    // ---- box branch: logits -> LDS --------------------------------------------------------------------------------------
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kc = 0; kc < 2 * PL; ++kc) Mma<T>::run(*reinterpret_cast<const uint4*>(w2s + (j * 16 + fr) * ROW2 + (kc * 4 + fg) * 16), xb[kc & 1], acc);
      *reinterpret_cast<float4*>(lgw + fr * LROW + j * 16 + fg * 4) = make_float4(__builtin_fmaf(acc[0], os2, b2v[j].x), __builtin_fmaf(acc[1], os2, b2v[j].y),
                                                                                  __builtin_fmaf(acc[2], os2, b2v[j].z), __builtin_fmaf(acc[3], os2, b2v[j].w));
    }
    // ---- class branch: sigmoid + max in registers ---------------------------------------------------------------------------
    float best = -1.f; int bi = 0;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kc = 0; kc < KC3 * PL; ++kc) Mma<T>::run(*reinterpret_cast<const uint4*>(w3s + (j * 16 + fr) * ROW3 + (kc * 4 + fg) * 16), xc[kc % KC3], acc);
      const float l[4] = {__builtin_fmaf(acc[0], os3, b3v[j].x), __builtin_fmaf(acc[1], os3, b3v[j].y), __builtin_fmaf(acc[2], os3, b3v[j].z), __builtin_fmaf(acc[3], os3, b3v[j].w)};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float sg = class_sigmoid(l[e]);
        if (sg > best) { best = sg; bi = j * 16 + fg * 4 + e; }
      }
    }
#pragma unroll
    for (int off = 16; off <= 32; off <<= 1) {          // lanes fr, fr+16, fr+32, fr+48 hold the same pixel
      const float ob = __shfl_xor(best, off, 64); const int oi = __shfl_xor(bi, off, 64);
      if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    // ---- DFL: lane (pixel fr, side fg) -----------------------------------------------------------------------------------
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's logits are in LDS (same-wave writes, in order)
    float v[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 tt = *reinterpret_cast<const float4*>(lgw + fr * LROW + fg * 16 + q * 4);
      v[4 * q] = tt.x; v[4 * q + 1] = tt.y; v[4 * q + 2] = tt.z; v[4 * q + 3] = tt.w;
    }
    const float dme = dfl_expect(v, w16);
    float d[4];
#pragma unroll
    for (int sd = 0; sd < 4; ++sd) d[sd] = __shfl(dme, fr + 16 * sd, 64);
    const int b = m / hw, a = m - b * hw;
    float box[4];
    dist_to_box(d, a, W, stride, box);
    if (m < M) {
      float* o = p.det + ((size_t)b * p.A + aoff + a) * 6;
      if (fg == 0) { o[0] = box[0]; o[1] = box[1]; o[2] = box[2]; o[3] = box[3]; }
      else if (fg == 1) { o[4] = best >= p.conf ? best : 0.f; o[5] = (float)bi; }
      if (p.nonfinite && ((fg == 1 && !(best >= 0.f)) || (fg == 0 && (!(box[0] - box[0] == 0.f) || !(box[2] - box[2] == 0.f))))) atomicAdd(p.nonfinite, 1);   // as decode_kernel
    }
    __builtin_amdgcn_wave_barrier();                    // the next tile overwrites this wave's logits
#pragma unroll
    for (int kc = 0; kc < 2; ++kc) xb[kc] = nb[kc];
#pragma unroll
    for (int kc = 0; kc < KC3; ++kc) xc[kc] = nc[kc];
  }
}

bool head_tail_supported(int dt, int ch, int split) { return dt != F32 && (ch == 128 || ch == 256) && (!split || dt == F16); }

template <class T, int CH, bool SPLIT = false> static void launch_head_tail_t(const HeadTailP& p, hipStream_t stream) {
  constexpr int PL = SPLIT ? 2 : 1;
  constexpr size_t lds = (size_t)80 * (PL * CH * 2 + 16) + 64 * (PL * 64 * 2 + 16) + 4 * 16 * 68 * 4;
  static_assert(lds <= 160 * 1024, "the level's weights must fit in LDS");
  // (per device: a process may hold handles on several GPUs, and the attribute is per device)
  static PerDevice once;
  if (once.first(once.index()))
    CC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(head_tail_kernel<T, CH, SPLIT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  int nb[3];
  for (int l = 0; l < 3; ++l) { const long M = (long)p.B * p.H[l] * p.W[l]; nb[l] = (int)((M + 64 * kTailTPB - 1) / (64 * kTailTPB)); }
  hipLaunchKernelGGL((head_tail_kernel<T, CH, SPLIT>), dim3(nb[0] + nb[1] + nb[2]), dim3(256), lds, stream, p, nb[0], nb[0] + nb[1]);
}

void launch_head_tail(int dt, const HeadTailP& p, hipStream_t stream) {
  CC_CHECK(head_tail_supported(dt, p.ch, p.split) && p.kw2 >= 64 * (1 + p.split) && p.kw3 >= p.ch * (1 + p.split), "fused DDetect tail: unsupported dtype / class-branch width");
  if (p.split) { if (p.ch == 256) launch_head_tail_t<f16_t, 256, true>(p, stream); else launch_head_tail_t<f16_t, 128, true>(p, stream); CC_HIP(hipGetLastError()); return; }
  if (dt == F16) { if (p.ch == 256) launch_head_tail_t<f16_t, 256>(p, stream); else launch_head_tail_t<f16_t, 128>(p, stream); }
  else { if (p.ch == 256) launch_head_tail_t<bf16_t, 256>(p, stream); else launch_head_tail_t<bf16_t, 128>(p, stream); }
  CC_HIP(hipGetLastError());
}

void launch_decode(const DecodeP& p, hipStream_t stream) {
  const size_t total = (size_t)p.B * p.A;
  hipLaunchKernelGGL(decode_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, p);
  CC_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------
// One 1024-thread workgroup per image.  key = score_bits<<32 | ~anchor  (unique; descending key order
// == descending score, ties by ascending anchor index == a stable descending sort).
constexpr int kMaxDet = 300;
constexpr int kSortN = 512;

__device__ __forceinline__ unsigned long long det_key(const float* det, int a) {
  return ((unsigned long long)__float_as_uint(det[(size_t)a * 6 + 4]) << 32) | (unsigned)(~(unsigned)a);
}

__global__ __launch_bounds__(1024) void topk_nms_kernel(const NmsP p) {
  __shared__ unsigned hist[256];
  __shared__ unsigned long long s_prefix;
  __shared__ int s_k;
  __shared__ unsigned s_cnt;
  __shared__ unsigned long long keys[kSortN];
  __shared__ float bx[kMaxDet][6];
  const int tid = threadIdx.x, b = blockIdx.x;
  const float* det = p.det + (size_t)b * p.A * 6;
  const int K = p.A < kMaxDet ? p.A : kMaxDet;

  __shared__ int s_done;
  __shared__ unsigned s_wsum[16];
  if (tid == 0) { s_prefix = 0ull; s_k = K; s_cnt = 0; s_done = 0; }
  if (tid < kSortN) keys[tid] = 0ull;
  __syncthreads();
  // ---- fast path: the scores are thresholded (0 below conf), so a frame has a few hundred positive keys among its 8400 anchors.
  // One pass compacts them; if all of them fit the 512-key sort, the top K are its first K - when there are fewer than K, joined by
  // the K - nz zero-score anchors with the lowest indices (key = score_bits<<32 | ~anchor: among equal scores the lower anchor is
  // the larger key), found with one block-wide scan over the first 1024 anchors, which hold at least 1024 - nz >= K - nz zero-score
  // ones.  Same set, same order after the sort as the radix select below, which now only runs for frames with more than 512 positive
  // anchors (eight passes of 8400 LDS atomics, most of them on one bucket: 71 us of a 1.4 ms single frame).
  for (int a = tid; a < p.A; a += 1024) {
    const unsigned long long key = det_key(det, a);
    if (key >> 32) { const unsigned pos = atomicAdd(&s_cnt, 1u); if (pos < kSortN) keys[pos] = key; }
  }
  __syncthreads();
  const unsigned nz = s_cnt;
  if (nz <= (unsigned)kSortN) {                        // every positive key is in keys[]: the sort below ranks them all, the first K win
    const unsigned need = nz < (unsigned)K ? (unsigned)K - nz : 0u;
    const bool zero = tid < p.A && (det_key(det, tid) >> 32) == 0ull;
    const unsigned long long bal = __ballot(zero);
    const int lane = tid & 63, wave = tid >> 6;
    const unsigned before = (unsigned)__popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) s_wsum[wave] = (unsigned)__popcll(bal);
    __syncthreads();
    unsigned woff = 0;
    for (int w = 0; w < wave; ++w) woff += s_wsum[w];
    const unsigned rank = woff + before;
    if (zero && rank < need) keys[nz + rank] = det_key(det, tid);
    __syncthreads();
  } else {
  __syncthreads();
  if (tid < kSortN) keys[tid] = 0ull;
  if (tid == 0) s_cnt = 0;
  __syncthreads();
  // ---- radix select (MSB first, 8 bits per pass) of the K-th largest key; stops as soon as the bucket it lands in is
  // wanted whole (distinct scores: after the four score bytes - the anchor-id bytes only matter for ties)
  for (int byte = 7; byte >= 0; --byte) {
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    const unsigned long long prefix = s_prefix;
    const int sh = 8 * (byte + 1);
    for (int a = tid; a < p.A; a += 1024) {
      const unsigned long long key = det_key(det, a);
      const bool match = byte == 7 ? true : ((key >> sh) == (prefix >> sh));
      if (match) atomicAdd(&hist[(unsigned)(key >> (8 * byte)) & 255u], 1u);
    }
    __syncthreads();
    if (tid < 64) {   // wave 0: suffix sums over buckets, 4 buckets per lane
      const unsigned h0 = hist[4 * tid], h1 = hist[4 * tid + 1], h2 = hist[4 * tid + 2], h3 = hist[4 * tid + 3];
      const unsigned s = h0 + h1 + h2 + h3;
      unsigned suf = s;   // inclusive suffix sum over lanes >= tid
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) { const unsigned t = __shfl_down(suf, off, 64); if (tid + off < 64) suf += t; }
      const unsigned above = suf - s;
      const unsigned kk = (unsigned)s_k;
      if (above < kk && kk <= suf) {
        unsigned cum = above; int v = 4 * tid + 3;
        const unsigned hh[4] = {h0, h1, h2, h3};
        for (int q = 3; q >= 0; --q) { if (cum + hh[q] >= kk) { v = 4 * tid + q; break; } cum += hh[q]; }
        s_prefix = prefix | ((unsigned long long)v << (8 * byte));
        s_k = (int)(kk - cum);
        if (hist[v] == kk - cum) s_done = 1;
      }
    }
    __syncthreads();
    if (s_done) break;
  }
  // ---- collect the K keys >= threshold (the slots were zeroed above), then bitonic sort descending
  const unsigned long long thr = s_prefix;
  if (K > 0)
    for (int a = tid; a < p.A; a += 1024) {
      const unsigned long long key = det_key(det, a);
      if (key >= thr) { const unsigned pos = atomicAdd(&s_cnt, 1u); if (pos < kSortN) keys[pos] = key; }
    }
  __syncthreads();
  }
  for (int k = 2; k <= kSortN; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      if (tid < kSortN) {
        const int ixj = tid ^ j;
        if (ixj > tid) {
          const unsigned long long x = keys[tid], y = keys[ixj];
          const bool desc = (tid & k) == 0;
          if (desc ? (x < y) : (x > y)) { keys[tid] = y; keys[ixj] = x; }
        }
      }
      __syncthreads();
    }
  // ---- gather boxes
  if (tid < kMaxDet) {
    if (tid < K) {
      const unsigned a = ~(unsigned)(keys[tid] & 0xffffffffull);
      for (int c = 0; c < 6; ++c) bx[tid][c] = det[(size_t)a * 6 + c];
    } else
      for (int c = 0; c < 6; ++c) bx[tid][c] = 0.f;
  }
  __syncthreads();
  // ---- mask NMS: row j is suppressed by ANY earlier row i (suppressed or not) of the same class.  Row 299 has 299 candidates and
  // every test carries an IEEE divide (the quotient is compared as the reference compares it): one thread per row made the last
  // row's serial loop ~30 us of the kernel, so three threads share a row (i = g, g+3, ...) and OR their verdicts through LDS.
  __shared__ unsigned s_sup[kMaxDet];
  if (tid < kMaxDet) s_sup[tid] = 0u;
  __syncthreads();
  {
    const int g = tid / kMaxDet, j = tid - g * kMaxDet;
    if (g < 3 && j < K) {
      const float jx1 = bx[j][0], jy1 = bx[j][1], jx2 = bx[j][2], jy2 = bx[j][3], jcl = bx[j][5];
      const float jarea = (jx2 - jx1) * (jy2 - jy1);
      bool hit = false;
      for (int i = g; i < j; i += 3) {
        const float ix1 = fmaxf(bx[i][0], jx1), iy1 = fmaxf(bx[i][1], jy1);
        const float ix2 = fminf(bx[i][2], jx2), iy2 = fminf(bx[i][3], jy2);
        const float inter = fmaxf(0.f, ix2 - ix1) * fmaxf(0.f, iy2 - iy1);
        const float ai = (bx[i][2] - bx[i][0]) * (bx[i][3] - bx[i][1]);
        const float iou = inter / (ai + jarea - inter);
        if (iou > p.iou_thr && bx[i][5] == jcl) hit = true;
      }
      if (hit) s_sup[j] = 1u;
    }
  }
  __syncthreads();
  if (tid < kMaxDet) {
    const float x1 = bx[tid][0], y1 = bx[tid][1], x2 = bx[tid][2], y2 = bx[tid][3], cl = bx[tid][5];
    const bool sup = s_sup[tid] != 0u;
    const float keep = (sup || tid >= K) ? 0.f : 1.f;
    float* o = p.out + ((size_t)b * kMaxDet + tid) * 6;
    // scale_boxes + clip_boxes (applied to zeroed rows too)
    o[0] = fminf(fmaxf((x1 * keep - p.pad_x) / p.gain, 0.f), p.src_w);
    o[1] = fminf(fmaxf((y1 * keep - p.pad_y) / p.gain, 0.f), p.src_h);
    o[2] = fminf(fmaxf((x2 * keep - p.pad_x) / p.gain, 0.f), p.src_w);
    o[3] = fminf(fmaxf((y2 * keep - p.pad_y) / p.gain, 0.f), p.src_h);
    o[4] = bx[tid][4] * keep;
    o[5] = cl * keep;
  }
}

void launch_topk_nms(const NmsP& p, hipStream_t stream) {
  hipLaunchKernelGGL(topk_nms_kernel, dim3(p.B), dim3(1024), 0, stream, p);
  CC_HIP(hipGetLastError());
}

}  // namespace cc

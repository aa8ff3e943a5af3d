// Crop preprocessing for the CLIP image tower: OpenCV-compatible INTER_CUBIC resize of variable-size uint8 HWC crops
// to S x S, then /255, (x-0.5)/0.5 and HWC -> CHW.  Replaces the host-side `ObjectFinder.preprocess`
// (models/objects.py:237-242: cv2.resize(img,(224,224),INTER_CUBIC) -> f32/255 -> (x-0.5)/0.5 -> transpose), which at
// 10 k crops per search re-index is the step in front of `precompute_embedding`.
//
// The arithmetic follows cv::resize's portable 8-bit cubic path (OpenCV 4.10 imgproc/src/resize.cpp, the version
// requirements.txt:3 pins): source coordinate fx = (float)((d+0.5)*scale-0.5) with scale = 1/(dst/src) in double,
// four float cubic weights (A = -0.75) rounded to 11-bit fixed point (short), borders replicated, horizontal pass in
// int32 without rounding, vertical pass (sum + 2^21) >> 22, saturate to uint8.  Integer accumulation makes the two
// separable passes equal to one 16-tap double sum, which is what each thread evaluates.  HBM-bound byte work: no LDS,
// one thread per output pixel, taps served by L1/L2 (a crop is a few hundred KB).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <string>
#include <vector>
#include "common.h"

namespace cc {
namespace {

struct CropTab { long long off; int h, w; int pad; };

__device__ __forceinline__ void cubic_coeffs(float x, short (&c)[4]) {
  const float A = -0.75f;
  float f[4];
  f[0] = ((A * (x + 1.f) - 5.f * A) * (x + 1.f) + 8.f * A) * (x + 1.f) - 4.f * A;
  f[1] = ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f;
  f[2] = ((A + 2.f) * (1.f - x) - (A + 3.f)) * (1.f - x) * (1.f - x) + 1.f;
  f[3] = 1.f - f[0] - f[1] - f[2];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    int v = __float2int_rn(f[k] * 2048.f);                 // saturate_cast<short>(float) = cvRound then clamp
    v = v < -32768 ? -32768 : (v > 32767 ? 32767 : v);
    c[k] = (short)v;
  }
}

// one axis: destination index d -> first source tap (s-1) and the four fixed-point weights
__device__ __forceinline__ int axis(int d, int src, int dst, short (&c)[4]) {
  const double inv_scale = (double)dst / (double)src;
  const double scale = 1.0 / inv_scale;
  float f = (float)(((double)d + 0.5) * scale - 0.5);
  const int s = (int)floorf(f);
  f -= (float)s;
  cubic_coeffs(f, c);
  return s;
}

__global__ __launch_bounds__(256) void crop_cubic_kernel(const uint8_t* __restrict__ pix, const CropTab* __restrict__ tab,
                                                         int S, float* __restrict__ out) {
  const int b = blockIdx.y;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= S * S) return;
  const int dy = idx / S, dx = idx - dy * S;
  const CropTab t = tab[b];
  const uint8_t* src = pix + t.off;
  short ax[4], ay[4];
  const int sx = axis(dx, t.w, S, ax), sy = axis(dy, t.h, S, ay);
  int acc[3] = {0, 0, 0};
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    int y = sy - 1 + r; y = y < 0 ? 0 : (y >= t.h ? t.h - 1 : y);
    const uint8_t* row = src + (size_t)y * t.w * 3;
    int hs[3] = {0, 0, 0};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      int x = sx - 1 + k; x = x < 0 ? 0 : (x >= t.w ? t.w - 1 : x);
      const uint8_t* p = row + x * 3;
      hs[0] += (int)p[0] * ax[k]; hs[1] += (int)p[1] * ax[k]; hs[2] += (int)p[2] * ax[k];
    }
    acc[0] += hs[0] * ay[r]; acc[1] += hs[1] * ay[r]; acc[2] += hs[2] * ay[r];
  }
  float* o = out + (size_t)b * 3 * S * S + idx;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    int v = (acc[c] + (1 << 21)) >> 22;
    v = v < 0 ? 0 : (v > 255 ? 255 : v);
    o[(size_t)c * S * S] = __fdiv_rn(__fdiv_rn((float)v, 255.0f) - 0.5f, 0.5f);
  }
}

// ---- cv2.resize(INTER_LINEAR) and cv2.warpAffine(INTER_LINEAR, BORDER_CONSTANT 0) for 8-bit 3-channel images ----------
// The two OpenCV calls `ObjectFinder.img_to_face` makes around BlazeFace (models/objects.py:247,318,332), restated from
// OpenCV 4.10's fixed-point paths (oracle/cv_warp_oracle.py has the derivation); one thread per output pixel.
struct ResizeP { const uint8_t* src; int H, W; uint8_t* dst; int dh, dw; int area2; };
struct WarpP { const uint8_t* src; int H, W; uint8_t* dst; int dh, dw; double m[6]; };   // m = the INVERTED 2x3 matrix

__device__ __forceinline__ void linear_axis(int d, int src, int dst, int& s0, int& s1, int& a0, int& a1) {
  const double inv_scale = (double)dst / (double)src;
  const double scale = 1.0 / inv_scale;
  float f = (float)(((double)d + 0.5) * scale - 0.5);
  int s = (int)floorf(f);
  f -= (float)s;
  if (s < 0) { f = 0.f; s = 0; }
  if (s >= src - 1) { f = 0.f; s = src - 1; }
  s0 = s; s1 = min(s + 1, src - 1);
  a0 = max(-32768, min(32767, __float2int_rn((1.0f - f) * 2048.f)));
  a1 = max(-32768, min(32767, __float2int_rn(f * 2048.f)));
}

__global__ __launch_bounds__(256) void resize_linear_kernel(const ResizeP p) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= p.dh * p.dw) return;
  const int dy = idx / p.dw, dx = idx - dy * p.dw;
  uint8_t* o = p.dst + (size_t)idx * 3;
  if (p.area2) {                                               // exact 2x2 decimation: OpenCV switches INTER_LINEAR to INTER_AREA
    const uint8_t* r0 = p.src + ((size_t)(2 * dy) * p.W + 2 * dx) * 3;
    const uint8_t* r1 = r0 + (size_t)p.W * 3;
    for (int c = 0; c < 3; ++c) o[c] = (uint8_t)(((int)r0[c] + r0[3 + c] + r1[c] + r1[3 + c] + 2) >> 2);
    return;
  }
  int x0, x1, ax0, ax1, y0, y1, ay0, ay1;
  linear_axis(dx, p.W, p.dw, x0, x1, ax0, ax1);
  linear_axis(dy, p.H, p.dh, y0, y1, ay0, ay1);
  const uint8_t* r0 = p.src + (size_t)y0 * p.W * 3;
  const uint8_t* r1 = p.src + (size_t)y1 * p.W * 3;
  for (int c = 0; c < 3; ++c) {
    const int h0 = (int)r0[x0 * 3 + c] * ax0 + (int)r0[x1 * 3 + c] * ax1;
    const int h1 = (int)r1[x0 * 3 + c] * ax0 + (int)r1[x1 * 3 + c] * ax1;
    const int v = (((ay0 * (h0 >> 4)) >> 16) + ((ay1 * (h1 >> 4)) >> 16) + 2) >> 2;
    o[c] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
  }
}

__global__ __launch_bounds__(256) void warp_affine_kernel(const WarpP p) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= p.dh * p.dw) return;
  const int dy = idx / p.dw, dx = idx - dy * p.dw;
  const long long adelta = __double2ll_rn(p.m[0] * (double)dx * 1024.0), bdelta = __double2ll_rn(p.m[3] * (double)dx * 1024.0);
  const long long X0 = __double2ll_rn((p.m[1] * (double)dy + p.m[2]) * 1024.0) + 16, Y0 = __double2ll_rn((p.m[4] * (double)dy + p.m[5]) * 1024.0) + 16;
  const long long X = (X0 + adelta) >> 5, Y = (Y0 + bdelta) >> 5;
  long long sxl = X >> 5, syl = Y >> 5;
  sxl = sxl < -32768 ? -32768 : (sxl > 32767 ? 32767 : sxl); syl = syl < -32768 ? -32768 : (syl > 32767 ? 32767 : syl);
  const int sx = (int)sxl, sy = (int)syl, fx = (int)(X & 31), fy = (int)(Y & 31);
  const int w00 = 32 * (32 - fy) * (32 - fx), w01 = 32 * (32 - fy) * fx, w10 = 32 * fy * (32 - fx), w11 = 32 * fy * fx;
  auto tap = [&](int yy, int xx, int c) -> int {
    return ((unsigned)xx < (unsigned)p.W && (unsigned)yy < (unsigned)p.H) ? (int)p.src[((size_t)yy * p.W + xx) * 3 + c] : 0;
  };
  uint8_t* o = p.dst + (size_t)idx * 3;
  for (int c = 0; c < 3; ++c) {
    const int v = (tap(sy, sx, c) * w00 + tap(sy, sx + 1, c) * w01 + tap(sy + 1, sx, c) * w10 + tap(sy + 1, sx + 1, c) * w11 + (1 << 14)) >> 15;
    o[c] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
  }
}

}  // namespace
}  // namespace cc

#define CC_API_BEGIN try {
#define CC_API_END                                                         \
  return 0; }                                                              \
  catch (const cc::Error& e) { cc::set_error(e.what()); return e.code; }   \
  catch (const std::exception& e) { cc::set_error(e.what()); return -1; }

extern "C" int cc_crop_preprocess(const uint8_t* pixels, const int64_t* offsets, const int32_t* heights, const int32_t* widths,
                                  int B, int pixels_on_device, int out_size, float* out_dev, int device, void* stream) {
  using namespace cc;
  CC_API_BEGIN
  CC_CHECK(pixels && offsets && heights && widths && out_dev && B > 0 && out_size > 0, "bad argument");
  int n = 0; CC_HIP(hipGetDeviceCount(&n));
  CC_CHECK(n > 0 && device >= 0 && device < n, "no such HIP device");
  CC_HIP(hipSetDevice(device));
  hipStream_t s = (hipStream_t)stream;
  std::vector<CropTab> tab(B);
  size_t total = 0;
  for (int b = 0; b < B; ++b) {
    CC_CHECK(heights[b] > 0 && widths[b] > 0 && offsets[b] >= 0, "empty crop");
    tab[b] = CropTab{(long long)offsets[b], heights[b], widths[b], 0};
    total = std::max(total, (size_t)offsets[b] + (size_t)heights[b] * widths[b] * 3);
  }
  CropTab* dtab = nullptr; uint8_t* dpix = nullptr;
  CC_HIP(hipMallocAsync((void**)&dtab, sizeof(CropTab) * B, s));
  CC_HIP(hipMemcpyAsync(dtab, tab.data(), sizeof(CropTab) * B, hipMemcpyHostToDevice, s));
  const uint8_t* src = pixels;
  if (!pixels_on_device) {
    CC_HIP(hipMallocAsync((void**)&dpix, total, s));
    CC_HIP(hipMemcpyAsync(dpix, pixels, total, hipMemcpyHostToDevice, s));
    src = dpix;
  }
  const int px = out_size * out_size;
  hipLaunchKernelGGL(crop_cubic_kernel, dim3((px + 255) / 256, B), dim3(256), 0, s, src, dtab, out_size, out_dev);
  CC_HIP(hipGetLastError());
  CC_HIP(hipFreeAsync(dtab, s));
  if (dpix) CC_HIP(hipFreeAsync(dpix, s));
  CC_HIP(hipStreamSynchronize(s));          // `tab` and the caller's host buffers may go away on return
  CC_API_END
}


namespace {
// host image in -> kernel -> host image out, on `device`; small images, synchronous by design (the reference's cv2 calls are too)
template <class Launch>
void run_image_op(int device, const uint8_t* src, size_t src_bytes, uint8_t* dst, size_t dst_bytes, Launch launch) {
  using namespace cc;
  int n = 0; CC_HIP(hipGetDeviceCount(&n));
  CC_CHECK(n > 0 && device >= 0 && device < n, "no such HIP device");
  CC_HIP(hipSetDevice(device));
  uint8_t *ds = nullptr, *dd = nullptr;
  CC_HIP(hipMalloc((void**)&ds, src_bytes + 16));
  if (hipMalloc((void**)&dd, dst_bytes + 16) != hipSuccess) { hipFree(ds); throw Error(-5, "hipMalloc failed"); }
  try {
    CC_HIP(hipMemcpy(ds, src, src_bytes, hipMemcpyHostToDevice));
    launch(ds, dd);
    CC_HIP(hipGetLastError());
    CC_HIP(hipMemcpy(dst, dd, dst_bytes, hipMemcpyDeviceToHost));
  } catch (...) { hipFree(ds); hipFree(dd); throw; }
  hipFree(ds); hipFree(dd);
}
}  // namespace

extern "C" int cc_cv_resize_linear_u8(const uint8_t* src, int H, int W, uint8_t* dst, int dh, int dw, int device) {
  using namespace cc;
  CC_API_BEGIN
  CC_CHECK(src && dst && H > 0 && W > 0 && dh > 0 && dw > 0, "bad argument");
  const double sx = 1.0 / ((double)dw / W), sy = 1.0 / ((double)dh / H);
  const int area2 = std::fabs(sx - 2.0) < 2.220446049250313e-16 && std::fabs(sy - 2.0) < 2.220446049250313e-16 && W % 2 == 0 && H % 2 == 0;
  run_image_op(device, src, (size_t)H * W * 3, dst, (size_t)dh * dw * 3, [&](uint8_t* ds, uint8_t* dd) {
    if (dh == H && dw == W) { CC_HIP(hipMemcpy(dd, ds, (size_t)H * W * 3, hipMemcpyDeviceToDevice)); return; }
    const ResizeP p{ds, H, W, dd, dh, dw, area2};
    hipLaunchKernelGGL(resize_linear_kernel, dim3((dh * dw + 255) / 256), dim3(256), 0, 0, p);
  });
  CC_API_END
}

extern "C" int cc_cv_warp_affine_u8(const uint8_t* src, int H, int W, const double* M, uint8_t* dst, int dh, int dw, int device) {
  using namespace cc;
  CC_API_BEGIN
  CC_CHECK(src && dst && M && H > 0 && W > 0 && dh > 0 && dw > 0, "bad argument");
  double m[6]; for (int i = 0; i < 6; ++i) m[i] = M[i];
  double D = m[0] * m[4] - m[1] * m[3];                        // cv::warpAffine inverts the forward matrix (imgwarp.cpp)
  D = D != 0 ? 1.0 / D : 0.0;
  const double A11 = m[4] * D, A22 = m[0] * D;
  m[0] = A11; m[1] *= -D; m[3] *= -D; m[4] = A22;
  const double b1 = -m[0] * m[2] - m[1] * m[5], b2 = -m[3] * m[2] - m[4] * m[5];
  m[2] = b1; m[5] = b2;
  run_image_op(device, src, (size_t)H * W * 3, dst, (size_t)dh * dw * 3, [&](uint8_t* ds, uint8_t* dd) {
    WarpP p{ds, H, W, dd, dh, dw, {m[0], m[1], m[2], m[3], m[4], m[5]}};
    hipLaunchKernelGGL(warp_affine_kernel, dim3((dh * dw + 255) / 256), dim3(256), 0, 0, p);
  });
  CC_API_END
}

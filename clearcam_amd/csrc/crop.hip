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

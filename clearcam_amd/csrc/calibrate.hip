// Calibration-aware weight rounding for the detector's 1x1 convs (C-ABI dtype 5, "f16c"; round 5).
//
// Plain f16 weights (11 bits) move boxes by up to a pixel on a float32 checkpoint; dtype "f16h" buys the tolerance back with a second
// weight plane in the backbone's 1x1 convs at 8 % of the frame rate.  Here every 1x1 conv keeps ONE f16 plane, but its weights are rounded
// with the rounding errors steered by the second moments of the conv's own input, H = E[x x^T] (the GPTQ recursion: Frantar et al. 2022,
// restated from the paper; columns in input-channel order, error fed forward through the Cholesky factor of H^-1): the error of the
// conv's OUTPUT on inputs that look like the calibration inputs shrinks 10-15x against nearest rounding (tools/dev/gptq_emulation.py,
// profiles/r04w_gptq_emulation.txt; on the real kernels: profiles/r05a_gptq_gpu.txt).
//   * sample_rows_kernel  gathers S pixel rows of a conv's input view (channel slice of an NHWC f32 tensor, two-source Concat, Upsample as
//                         an index shift - the ConvP::s0 / s1 views of the f32 calibration plan) into a dense (S, Cin) f32 matrix;
//   * gptq_round_f16      host, double precision, no LAPACK (K <= 1024): H + damp mean(diag) I = L L^T, H^-1 = L^-T L^-1 = G G^T, then the
//                         column walk.  Bit-compatible with oracle/lowprec_oracle.py::gptq_f16 up to the factorisations' rounding
//                         (tests/test_abi_and_host.py: identical on >= 99.5 % of the weights, same expected output error).
// The reference (detection/yolov9.py:372-373) loads float32 safetensors and computes in float32; which 16-bit values stand in for them is
// this library's business, and the parity bar (tests/test_gpu_yolo.py::test_calibrated_mode_*) is the one "f16h" is held to.
#include <cmath>
#include <cstring>
#include <vector>
#include "kernels.h"

namespace cc {

// out[j][c] = input channel c (over s0 then s1) of output pixel rows[j] of a 1x1 stride-1 conv over f32 activations
__global__ __launch_bounds__(256) void sample_rows_kernel(const ConvP p, const int* rows, int S, float* out) {
  const int j = blockIdx.x;
  if (j >= S) return;
  const int m = rows[j], hw = p.Ho * p.Wo;
  const int b = m / hw, rem = m - b * hw, ho = rem / p.Wo, wo = rem - ho * p.Wo;
  const float* a0 = reinterpret_cast<const float*>(p.s0.ptr) + (((size_t)b * p.s0.H + (ho >> p.s0.shift)) * p.s0.W + (wo >> p.s0.shift)) * p.s0.cstride + p.s0.coff;
  const float* a1 = p.s1.C ? reinterpret_cast<const float*>(p.s1.ptr) + (((size_t)b * p.s1.H + (ho >> p.s1.shift)) * p.s1.W + (wo >> p.s1.shift)) * p.s1.cstride + p.s1.coff : nullptr;
  for (int c = threadIdx.x; c < p.Cin; c += 256) out[(size_t)j * p.Cin + c] = c < p.s0.C ? a0[c] : a1[c - p.s0.C];
}

void launch_sample_rows(const ConvP& p, const int* rows_dev, int S, float* out_dev, hipStream_t stream) {
  CC_CHECK(p.ks == 1 && p.stride == 1 && p.pad == 0 && p.s0.shift >= 0 && p.s1.shift >= 0, "sample_rows: 1x1 stride-1 convs only");
  hipLaunchKernelGGL(sample_rows_kernel, dim3(S), dim3(256), 0, stream, p, rows_dev, S, out_dev);
  CC_HIP(hipGetLastError());
}

static inline float f16_round(float f) { return (float)(f16_t)f; }   // round to nearest even, overflow to inf, subnormals exact

// lower Cholesky factor of the symmetric positive definite n x n matrix a (row major), in place in the lower triangle; false if not PD
static bool cholesky_lower(std::vector<double>& a, int n) {
  for (int j = 0; j < n; ++j) {
    double d = a[(size_t)j * n + j];
    for (int k = 0; k < j; ++k) d -= a[(size_t)j * n + k] * a[(size_t)j * n + k];
    if (!(d > 0)) return false;
    d = std::sqrt(d); a[(size_t)j * n + j] = d;
    for (int i = j + 1; i < n; ++i) {
      double s = a[(size_t)i * n + j];
      const double *ri = &a[(size_t)i * n], *rj = &a[(size_t)j * n];
      for (int k = 0; k < j; ++k) s -= ri[k] * rj[k];
      a[(size_t)i * n + j] = s / d;
    }
  }
  return true;
}

// H (ci x ci, row major) = X^T X / rows for X (rows x ci) f32, accumulated in double
void second_moments(const float* X, int rows, int ci, std::vector<double>& H) {
  H.assign((size_t)ci * ci, 0.0);
  std::vector<double> xr(ci);
  for (int r = 0; r < rows; ++r) {
    const float* x = X + (size_t)r * ci;
    for (int i = 0; i < ci; ++i) xr[i] = x[i];
    for (int i = 0; i < ci; ++i) {
      const double xi = xr[i];
      if (xi == 0.0) continue;
      double* h = &H[(size_t)i * ci];
      for (int j = i; j < ci; ++j) h[j] += xi * xr[j];
    }
  }
  const double inv = rows > 0 ? 1.0 / rows : 0.0;
  for (int i = 0; i < ci; ++i)
    for (int j = i; j < ci; ++j) { const double v = H[(size_t)i * ci + j] * inv; H[(size_t)i * ci + j] = v; H[(size_t)j * ci + i] = v; }
}

// w (co x ci) f32 -> out: f16-representable f32 values.  0 ok, -1 / -2: H (damped) or its inverse not positive definite - the caller then
// leaves the conv to the library's controlled rounding.
int gptq_round_f16(const float* w, int co, int ci, const double* H, double damp, float* out) {
  const int n = ci;
  std::vector<double> A((size_t)n * n);
  double mean_diag = 0;
  for (int i = 0; i < n; ++i) mean_diag += H[(size_t)i * n + i];
  mean_diag /= n;
  if (!(mean_diag > 0) || !std::isfinite(mean_diag)) return -1;
  for (size_t i = 0; i < (size_t)n * n; ++i) A[i] = H[i];
  for (int i = 0; i < n; ++i) A[(size_t)i * n + i] += damp * mean_diag;
  if (!cholesky_lower(A, n)) return -1;
  std::vector<double> Li((size_t)n * n, 0.0);              // Li = L^-1 (lower): forward substitution, column c of the identity
  for (int c = 0; c < n; ++c)
    for (int i = c; i < n; ++i) {
      double s = i == c ? 1.0 : 0.0;
      for (int k = c; k < i; ++k) s -= A[(size_t)i * n + k] * Li[(size_t)k * n + c];
      Li[(size_t)i * n + c] = s / A[(size_t)i * n + i];
    }
  std::vector<double>& Hinv = A;                           // H^-1 = Li^T Li (A is done)
  std::fill(Hinv.begin(), Hinv.end(), 0.0);
  for (int i = 0; i < n; ++i)
    for (int j = 0; j <= i; ++j) {
      double s = 0;
      for (int k = i; k < n; ++k) s += Li[(size_t)k * n + i] * Li[(size_t)k * n + j];
      Hinv[(size_t)i * n + j] = Hinv[(size_t)j * n + i] = s;
    }
  if (!cholesky_lower(Hinv, n)) return -2;                 // H^-1 = G G^T, G lower; U = G^T is the upper factor the walk uses: U[i][j] = G[j][i]
  std::vector<double> W((size_t)co * n);
  for (size_t i = 0; i < (size_t)co * n; ++i) W[i] = w[i];
  for (int i = 0; i < n; ++i) {
    const double uii = Hinv[(size_t)i * n + i];
    for (int o = 0; o < co; ++o) {
      double* row = &W[(size_t)o * n];
      const float q = f16_round((float)row[i]);
      out[(size_t)o * n + i] = q;
      if (!std::isfinite(q)) continue;
      const double err = (row[i] - (double)q) / uii;
      for (int j = i + 1; j < n; ++j) row[j] -= err * Hinv[(size_t)j * n + i];
    }
  }
  return 0;
}

}  // namespace cc

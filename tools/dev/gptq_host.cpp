// dev prototype (round 4, for round 5; NOT part of libclearcam_hip): the host side of calibration-aware weight rounding - DESIGN.md section 8.
// Rounds the weights of one 1x1 conv (co x ci, f32) to f16-representable values column by column, feeding each column's rounding error
// forward through the inverse of H = E[x x^T] (ci x ci, f64) of the conv's input activations: the GPTQ recursion, in the order and with the
// factorisations of tools/dev/gptq_gpu.py::gptq_f16 (H + damp * mean(diag) I = L L^T;  H^-1 = L^-T L^-1 = U^T U with U upper triangular).
// Plain C++, no LAPACK: K <= 1024 here, the three O(K^3) steps take well under a second.   g++ -O2 -shared -fPIC -o gptq_host.so gptq_host.cpp
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

static inline float f16_round(float f) {                 // f32 -> nearest f16 (ties to even) -> f32; overflow to inf, subnormals exact
  uint32_t u; std::memcpy(&u, &f, 4);
  const uint32_t sign = u & 0x80000000u; u &= 0x7fffffffu;
  if (u >= 0x7f800000u) return f;                        // inf / nan
  float a; std::memcpy(&a, &u, 4);
  if (a >= 65520.0f) { u = 0x7f800000u | sign; std::memcpy(&f, &u, 4); return f; }
  if (a < 6.103515625e-5f) {                             // below the smallest normal f16: multiples of 2^-24
    const float q = std::nearbyintf(a * 16777216.0f) / 16777216.0f;      // round-to-nearest-even in the default rounding mode
    std::memcpy(&u, &q, 4); u |= sign; std::memcpy(&f, &u, 4); return f;
  }
  const uint32_t rem = u & 0x1fffu, keep = u & ~0x1fffu;                 // 13 mantissa bits go
  u = keep + ((rem > 0x1000u || (rem == 0x1000u && (keep & 0x2000u))) ? 0x2000u : 0u);
  u |= sign; std::memcpy(&f, &u, 4); return f;
}

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

extern "C" int gptq_round_f16(const float* w, int co, int ci, const double* H, double damp, float* out) {
  const int n = ci;
  std::vector<double> A((size_t)n * n);
  double mean_diag = 0; for (int i = 0; i < n; ++i) mean_diag += H[(size_t)i * n + i]; mean_diag /= n;
  for (size_t i = 0; i < (size_t)n * n; ++i) A[i] = H[i];
  for (int i = 0; i < n; ++i) A[(size_t)i * n + i] += damp * mean_diag;
  if (!cholesky_lower(A, n)) return -1;
  // Li = L^-1 (lower), then Hinv = Li^T Li
  std::vector<double> Li((size_t)n * n, 0.0);
  for (int c = 0; c < n; ++c) {                            // forward substitution, column c of the identity
    for (int i = c; i < n; ++i) {
      double s = i == c ? 1.0 : 0.0;
      for (int k = c; k < i; ++k) s -= A[(size_t)i * n + k] * Li[(size_t)k * n + c];
      Li[(size_t)i * n + c] = s / A[(size_t)i * n + i];
    }
  }
  std::vector<double> Hinv((size_t)n * n, 0.0);
  for (int i = 0; i < n; ++i)
    for (int j = 0; j <= i; ++j) {
      double s = 0; for (int k = i; k < n; ++k) s += Li[(size_t)k * n + i] * Li[(size_t)k * n + j];
      Hinv[(size_t)i * n + j] = Hinv[(size_t)j * n + i] = s;
    }
  if (!cholesky_lower(Hinv, n)) return -2;                 // Hinv = G G^T, G lower; U = G^T is the upper factor the recursion walks
  std::vector<double> W((size_t)co * n);
  for (size_t i = 0; i < (size_t)co * n; ++i) W[i] = w[i];
  for (int i = 0; i < n; ++i) {
    const double uii = Hinv[(size_t)i * n + i];
    for (int o = 0; o < co; ++o) {
      double* row = &W[(size_t)o * n];
      const float q = f16_round((float)row[i]);
      out[(size_t)o * n + i] = q;
      const double err = (row[i] - (double)q) / uii;
      for (int j = i + 1; j < n; ++j) row[j] -= err * Hinv[(size_t)j * n + i];     // U[i][j] = G[j][i]
    }
  }
  return 0;
}

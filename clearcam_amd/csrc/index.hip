// Device-resident embedding index: brute-force cosine scan + top-k.
//
// Stands behind the scoring loop of ObjectFinder.search (models/objects.py:365-376), where the reference
// evaluates `(img_embedding @ text_embedding.T).item()` once per stored crop in a Python loop.  Here the
// (N, dim) float32 matrix lives in HBM and one query costs one streaming pass over it (HBM-bound:
// N*dim*4 bytes), followed by a two-stage radix-select top-k.  Scores are exact f32 dot products
// (sequential fma per lane + wave tree), so ranking ties break by row id exactly as a stable sort would.
#include <algorithm>
#include <cstring>
#include <memory>
#include <vector>
#include "kernels.h"
#include "../../include/clearcam_hip.h"

using namespace cc;

struct cc_index {
  int dim = 0, device = 0; int64_t capacity = 0, n = 0;
  int storage = F32;                              // F32: exact f32 dot products; BF16: rows rounded to bf16 (half the scan bytes, scores within ~1e-3)
  float* emb = nullptr;                           // (capacity, dim) rows in the storage type
  int* grp = nullptr;                             // (capacity) group id of each row (camera/day bucket chosen by the caller); 0 by default
  unsigned char* allowed = nullptr; int allowed_cap = 0;   // per-group filter of the current search (device copy)
  size_t row_bytes() const { return (size_t)dim * (storage == F32 ? 4 : 2); }
  hipStream_t stream = nullptr;
  float* q_dev = nullptr; int q_cap = 0;
  uint16_t* q16 = nullptr; int q16_cap = 0;     // bf16 copy of the queries (bf16 index, many queries: one MFMA GEMM pass)
  float* scores = nullptr; size_t scores_cap = 0;
  unsigned long long* cand = nullptr; size_t cand_cap = 0;
  int* idx_dev = nullptr; float* sc_dev = nullptr; size_t out_cap = 0;
  char* pin = nullptr; size_t pin_cap = 0;      // pinned, device-visible host staging: queries in, small results out
};

namespace {

constexpr int kChunk = 16384;     // rows per stage-1 block
constexpr int kMaxK = 1024;

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// scores[q][row] = <emb[row], query[q]>; one wave per row (grid-stride), up to 4 queries per pass.
template <int QB>
__global__ __launch_bounds__(256) void scores_kernel(const float* __restrict__ emb, const float* __restrict__ q, float* __restrict__ out,
                                                      long n, int dim, int nq) {
  const int lane = threadIdx.x & 63;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long)gridDim.x * 4;
  for (long row = wave; row < n; row += nwaves) {
    const float* e = emb + row * dim;
    float acc[QB];
#pragma unroll
    for (int c = 0; c < QB; ++c) acc[c] = 0.f;
    for (int j = lane * 4; j < dim; j += 256) {
      const float4 v = *reinterpret_cast<const float4*>(e + j);
#pragma unroll
      for (int c = 0; c < QB; ++c) {
        if (c < nq) {
          const float4 w = *reinterpret_cast<const float4*>(q + (long)c * dim + j);
          acc[c] = fmaf(v.x, w.x, acc[c]); acc[c] = fmaf(v.y, w.y, acc[c]); acc[c] = fmaf(v.z, w.z, acc[c]); acc[c] = fmaf(v.w, w.w, acc[c]);
        }
      }
    }
#pragma unroll
    for (int c = 0; c < QB; ++c) {
      const float s = wsum(acc[c]);
      if (lane == 0 && c < nq) out[(long)c * n + row] = s;
    }
  }
}

// The same scores with more memory parallelism (dim % 256 == 0, e.g. 768 / 512 / 1024): a row is covered by 16 lanes,
// so one wave reads four rows per load instruction and keeps four independent 16-byte loads per lane in flight (16 KB
// per wave); the queries sit in LDS; the per-row reduction is 4 shuffle steps instead of 6.  Each lane accumulates its
// 16-byte slots in ascending order with fma, then the 16 partial sums are added as a butterfly: a fixed order, so
// scores are deterministic and ties still break by row id.
template <int QB>
__global__ __launch_bounds__(256) void scan_kernel(const float* __restrict__ emb, const float* __restrict__ q, float* __restrict__ out,
                                                    long n, int dim) {
  extern __shared__ float4 qs[];                                  // QB x dim/4
  const int nchunk = dim >> 2;
  for (int i = threadIdx.x; i < QB * nchunk; i += 256) qs[i] = reinterpret_cast<const float4*>(q)[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, sub = lane & 15, grp = lane >> 4;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long)gridDim.x * 4;
  const int steps = dim >> 6;                                     // 16 lanes x 4 floats per step; a multiple of 4
  for (long r0 = wave * 4; r0 < n; r0 += nwaves * 4) {
    const long row = r0 + grp;
    const bool live = row < n;
    const float4* e = reinterpret_cast<const float4*>(emb + (live ? row : n - 1) * dim) + sub;
    float acc[QB];
#pragma unroll
    for (int c = 0; c < QB; ++c) acc[c] = 0.f;
    for (int t0 = 0; t0 < steps; t0 += 4) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = e[(t0 + u) * 16];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int c = 0; c < QB; ++c) {
          const float4 w = qs[c * nchunk + (t0 + u) * 16 + sub];
          acc[c] = fmaf(v[u].x, w.x, acc[c]); acc[c] = fmaf(v[u].y, w.y, acc[c]); acc[c] = fmaf(v[u].z, w.z, acc[c]); acc[c] = fmaf(v[u].w, w.w, acc[c]);
        }
    }
#pragma unroll
    for (int c = 0; c < QB; ++c) {
      float sum = acc[c];
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
      if (sub == 0 && live) out[(long)c * n + row] = sum;
    }
  }
}

template <int QB> void launch_scan(cc_index* h, const float* q, float* out, hipStream_t s) {
  const int blocks = (int)std::min<int64_t>((h->n + 15) / 16, 256 * 8);
  hipLaunchKernelGGL(scan_kernel<QB>, dim3(blocks), dim3(256), (size_t)QB * h->dim * 4, s, h->emb, q, out, (long)h->n, h->dim);
}

// bf16-stored rows: 16 lanes x 8 halfs per step, two independent groups of four rows per wave iteration (four 16-byte loads per
// lane in flight, like the f32 kernel); f32 queries from LDS, f32 fma accumulation in a fixed order.  dim % 256 == 0.
template <int QB>
__global__ __launch_bounds__(256) void scan_bf16_kernel(const uint16_t* __restrict__ emb, const float* __restrict__ q, float* __restrict__ out,
                                                         long n, int dim) {
  extern __shared__ float4 qs[];
  const int nchunk = dim >> 2;
  for (int i = threadIdx.x; i < QB * nchunk; i += 256) qs[i] = reinterpret_cast<const float4*>(q)[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, sub = lane & 15, grp = lane >> 4;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long)gridDim.x * 4;
  const int steps = dim >> 7;                                     // 16 lanes x 8 halfs per step; even
  for (long r0 = wave * 8; r0 < n; r0 += nwaves * 8) {
    const long rowa = r0 + grp, rowb = r0 + 4 + grp;
    const uint4* ea = reinterpret_cast<const uint4*>(emb + (rowa < n ? rowa : n - 1) * dim) + sub;
    const uint4* eb = reinterpret_cast<const uint4*>(emb + (rowb < n ? rowb : n - 1) * dim) + sub;
    float acca[QB], accb[QB];
#pragma unroll
    for (int c = 0; c < QB; ++c) { acca[c] = 0.f; accb[c] = 0.f; }
    for (int t0 = 0; t0 < steps; t0 += 2) {
      uint4 va[2], vb[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) { va[u] = ea[(t0 + u) * 16]; vb[u] = eb[(t0 + u) * 16]; }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const unsigned wa[4] = {va[u].x, va[u].y, va[u].z, va[u].w}, wb[4] = {vb[u].x, vb[u].y, vb[u].z, vb[u].w};
#pragma unroll
        for (int c = 0; c < QB; ++c) {
          const float4 q0 = qs[c * nchunk + ((t0 + u) * 16 + sub) * 2], q1 = qs[c * nchunk + ((t0 + u) * 16 + sub) * 2 + 1];
          const float qq[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            acca[c] = fmaf(__uint_as_float(wa[e] << 16), qq[2 * e], acca[c]); acca[c] = fmaf(__uint_as_float(wa[e] & 0xffff0000u), qq[2 * e + 1], acca[c]);
            accb[c] = fmaf(__uint_as_float(wb[e] << 16), qq[2 * e], accb[c]); accb[c] = fmaf(__uint_as_float(wb[e] & 0xffff0000u), qq[2 * e + 1], accb[c]);
          }
        }
      }
    }
#pragma unroll
    for (int c = 0; c < QB; ++c) {
      float sa = acca[c], sb = accb[c];
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) { sa += __shfl_xor(sa, o, 64); sb += __shfl_xor(sb, o, 64); }
      if (sub == 0 && rowa < n) out[(long)c * n + rowa] = sa;
      if (sub == 0 && rowb < n) out[(long)c * n + rowb] = sb;
    }
  }
}

template <int QB> void launch_scan_bf16(cc_index* h, const float* q, float* out, hipStream_t s) {
  const int blocks = (int)std::min<int64_t>((h->n + 31) / 32, 256 * 8);
  hipLaunchKernelGGL(scan_bf16_kernel<QB>, dim3(blocks), dim3(256), (size_t)QB * h->dim * 4, s, reinterpret_cast<const uint16_t*>(h->emb), q, out, (long)h->n, h->dim);
}

// f32 rows -> bf16 rows (round to nearest even), for cc_index_add on a bf16-stored index
__global__ void to_bf16_kernel(const float* __restrict__ in, uint16_t* __restrict__ out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = f32_to_bf16_bits(in[i]);
}

__device__ __forceinline__ unsigned ordered(float f) {          // monotone float -> uint
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float unordered(unsigned u) {
  return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

// Block-wide exact top-K of `count` unique 64-bit keys (key(i) for i in [0,count)): MSB-first radix select of the
// K-th largest key, then collection of every key >= it into `sel` (unsorted; K entries, zero padded).
template <class KeyFn>
__device__ void block_topk(KeyFn key, long count, int K, unsigned long long* sel, unsigned* hist,
                           unsigned long long* s_prefix, int* s_k, unsigned* s_cnt) {
  const int tid = threadIdx.x, nt = blockDim.x;
  for (int i = tid; i < K; i += nt) sel[i] = 0ull;
  __shared__ int s_done;
  if (tid == 0) { *s_prefix = 0ull; *s_k = (int)(count < K ? count : K); *s_cnt = 0; s_done = 0; }
  __syncthreads();
  if (count <= 0) return;
  for (int byte = 7; byte >= 0; --byte) {
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    const unsigned long long prefix = *s_prefix;
    const int sh = 8 * (byte + 1);
    for (long i = tid; i < count; i += nt) {
      const unsigned long long k = key(i);
      if (byte == 7 || (k >> sh) == (prefix >> sh)) atomicAdd(&hist[(unsigned)(k >> (8 * byte)) & 255u], 1u);
    }
    __syncthreads();
    if (tid < 64) {
      const unsigned h0 = hist[4 * tid], h1 = hist[4 * tid + 1], h2 = hist[4 * tid + 2], h3 = hist[4 * tid + 3];
      const unsigned s = h0 + h1 + h2 + h3;
      unsigned suf = s;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) { const unsigned t = __shfl_down(suf, off, 64); if (tid + off < 64) suf += t; }
      const unsigned above = suf - s, kk = (unsigned)*s_k;
      if (above < kk && kk <= suf) {
        unsigned cum = above; int v = 4 * tid + 3;
        const unsigned hh[4] = {h0, h1, h2, h3};
        for (int qd = 3; qd >= 0; --qd) { if (cum + hh[qd] >= kk) { v = 4 * tid + qd; break; } cum += hh[qd]; }
        *s_prefix = prefix | ((unsigned long long)v << (8 * byte));
        *s_k = (int)(kk - cum);
        // the whole bucket is wanted: every key at or above the prefix (lower bytes zero) is selected and the remaining
        // passes would only confirm it.  With distinct scores this fires after the four score bytes (row-id bytes skipped).
        if (hist[v] == kk - cum) s_done = 1;
      }
    }
    __syncthreads();
    if (s_done) break;
  }
  const unsigned long long thr = *s_prefix;
  for (long i = tid; i < count; i += nt) {
    const unsigned long long k = key(i);
    if (k >= thr && k != 0ull) { const unsigned pos = atomicAdd(s_cnt, 1u); if (pos < (unsigned)K) sel[pos] = k; }   // zero = padding / filtered-out row
  }
  __syncthreads();
}

// stage 1: per (query, chunk of rows) top-K keys -> cand[q][chunk][K]
// `allowed` (one byte per group) filters rows by their group id: a row of a group that is not allowed contributes the padding key.
__global__ __launch_bounds__(1024) void topk_stage1(const float* __restrict__ scores, long n, int K, unsigned long long* __restrict__ cand,
                                                    const int* __restrict__ grp, const unsigned char* __restrict__ allowed, int n_allowed) {
  __shared__ unsigned hist[256]; __shared__ unsigned long long s_prefix; __shared__ int s_k; __shared__ unsigned s_cnt;
  __shared__ unsigned long long sel[kMaxK];
  const int chunk = blockIdx.x, q = blockIdx.y;
  const long base = (long)chunk * kChunk, count = min((long)kChunk, n - base);
  const float* s = scores + (long)q * n + base;
  auto key = [&](long i) -> unsigned long long {
    if (allowed) { const unsigned g = (unsigned)grp[base + i]; if (g >= (unsigned)n_allowed || !allowed[g]) return 0ull; }
    return ((unsigned long long)ordered(s[i]) << 32) | (unsigned)(~(unsigned)(base + i));
  };
  block_topk(key, count, K, sel, hist, &s_prefix, &s_k, &s_cnt);
  unsigned long long* o = cand + ((size_t)q * gridDim.x + chunk) * K;
  for (int i = threadIdx.x; i < K; i += blockDim.x) o[i] = sel[i];
}

// stage 2: merge candidates of one query, sort descending, emit (row id, score)
__global__ __launch_bounds__(1024) void topk_stage2(const unsigned long long* __restrict__ cand, long ncand, int K, int* __restrict__ idx, float* __restrict__ score) {
  __shared__ unsigned hist[256]; __shared__ unsigned long long s_prefix; __shared__ int s_k; __shared__ unsigned s_cnt;
  __shared__ unsigned long long sel[kMaxK];
  const int q = blockIdx.x, tid = threadIdx.x;
  const unsigned long long* c = cand + (size_t)q * ncand;
  auto key = [&](long i) { return c[i]; };      // zero keys (padding) sort last and never beat a real key
  for (int i = tid; i < kMaxK; i += blockDim.x) sel[i] = 0ull;
  __syncthreads();
  block_topk(key, ncand, K, sel, hist, &s_prefix, &s_k, &s_cnt);
  int P = 2;                                                     // the K selected keys sit in sel[0..K): sort the next power of two
  while (P < K) P <<= 1;
  for (int k = 2; k <= P; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      const int ixj = tid ^ j;
      if (ixj > tid && ixj < P) {
        const unsigned long long x = sel[tid], y = sel[ixj];
        const bool desc = (tid & k) == 0;
        if (desc ? (x < y) : (x > y)) { sel[tid] = y; sel[ixj] = x; }
      }
      __syncthreads();
    }
  if (tid < K) {
    const unsigned long long k = sel[tid];
    if (k == 0ull) { idx[(size_t)q * K + tid] = -1; score[(size_t)q * K + tid] = -INFINITY; }
    else { idx[(size_t)q * K + tid] = (int)(~(unsigned)(k & 0xffffffffull)); score[(size_t)q * K + tid] = unordered((unsigned)(k >> 32)); }
  }
}

void ensure(void** p, size_t* cap, size_t bytes) {
  if (*cap >= bytes) return;
  if (*p) hipFree(*p);
  CC_HIP(hipMalloc(p, bytes + 256));
  *cap = bytes;
}

constexpr size_t kPinResultBytes = 1 << 20;      // results up to this size are written by the kernel straight into pinned host memory

void ensure_pinned(cc_index* h, size_t bytes) {
  if (h->pin_cap >= bytes) return;
  if (h->pin) hipHostFree(h->pin);
  h->pin = nullptr; h->pin_cap = 0;
  CC_HIP(hipHostMalloc((void**)&h->pin, bytes, hipHostMallocDefault));
  h->pin_cap = bytes;
}

// Host queries go through the pinned buffer (a pageable source makes the runtime stage and block; 3 KB per query).
void upload_queries(cc_index* h, const float* q, int Q, int on_device, hipStream_t s) {
  if (h->q_cap < Q) { if (h->q_dev) hipFree(h->q_dev); CC_HIP(hipMalloc((void**)&h->q_dev, (size_t)Q * h->dim * 4 + 256)); h->q_cap = Q; }
  const size_t bytes = (size_t)Q * h->dim * 4;
  if (!on_device && bytes <= h->pin_cap) {
    memcpy(h->pin, q, bytes);
    CC_HIP(hipMemcpyAsync(h->q_dev, h->pin, bytes, hipMemcpyHostToDevice, s));
  } else {
    CC_HIP(hipMemcpyAsync(h->q_dev, q, bytes, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s));
  }
}

void compute_scores(cc_index* h, int Q, hipStream_t s) {
  ensure((void**)&h->scores, &h->scores_cap, (size_t)Q * h->n * 4);
  // More than four queries: one pass over the matrix as a GEMM on the exact-f32 MFMA path (queries = the 128-row tile,
  // index rows = output channels, so every embedding is read from HBM once and scores come out (Q, N) row-major).
  // Sixteen GEMV passes for 64 queries become one.  Products are exact f32 either way; only the summation order differs
  // from the GEMV kernel (last-bit differences between the two paths).
  static const bool wide = [] { const char* e = getenv("CLEARCAM_SCAN"); return e ? atoi(e) != 0 : true; }();
  if (h->storage == BF16 && Q > 16 && h->n % 8 == 0 && h->n < (1L << 31)) {
    // many queries against bf16 rows: ONE pass as a bf16 MFMA GEMM (queries rounded to bf16 as well: scores within ~2e-3 of f32)
    if (h->q16_cap < Q) { if (h->q16) hipFree(h->q16); CC_HIP(hipMalloc((void**)&h->q16, (size_t)Q * h->dim * 2 + 256)); h->q16_cap = Q; }
    const size_t cnt = (size_t)Q * h->dim;
    hipLaunchKernelGGL(to_bf16_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, s, h->q_dev, h->q16, cnt);
    const ConvP g = gemm_params(h->q16, h->dim, Q, h->dim, h->emb, h->dim, nullptr, (int)h->n, h->scores, (int)h->n, 1, 0, nullptr, 0, 0);
    if (conv_mfma_supported(BF16, g)) { launch_conv_mfma(BF16, g, s); return; }
  }
  if (h->storage == BF16) {                         // bf16 rows: 8 queries per pass over half the bytes (dim % 256 == 0 is checked at create)
    for (int q0 = 0; q0 < Q; q0 += 8) {
      const float* q = h->q_dev + (size_t)q0 * h->dim; float* out = h->scores + (size_t)q0 * h->n;
      switch (std::min(8, Q - q0)) {
        case 1: launch_scan_bf16<1>(h, q, out, s); break;  case 2: launch_scan_bf16<2>(h, q, out, s); break;
        case 3: launch_scan_bf16<3>(h, q, out, s); break;  case 4: launch_scan_bf16<4>(h, q, out, s); break;
        case 5: launch_scan_bf16<5>(h, q, out, s); break;  case 6: launch_scan_bf16<6>(h, q, out, s); break;
        case 7: launch_scan_bf16<7>(h, q, out, s); break;  default: launch_scan_bf16<8>(h, q, out, s); break;
      }
    }
    CC_HIP(hipGetLastError());
    return;
  }
  const bool scan8 = wide && h->dim % 256 == 0 && h->dim <= 4096;   // scan_kernel: up to 8 queries per pass at HBM speed
  // measured on a 125 k x 768 shard: a scan pass costs ~65 us whatever its query count (1..8), the GEMM ~220 us flat up to 128 queries
  if (Q > (scan8 ? 16 : 4) && h->n % 4 == 0 && h->dim % 32 == 0) {
    const ConvP g = gemm_params(h->q_dev, h->dim, Q, h->dim, h->emb, h->dim, nullptr, (int)h->n, h->scores, (int)h->n, 1, 0, nullptr, 0, 0);
    if (h->n < (1L << 31) && conv_mfma_supported(F32, g)) { launch_conv_mfma(F32, g, s); return; }
  }
  if (scan8) {
    for (int q0 = 0; q0 < Q; q0 += 8) {
      const float* q = h->q_dev + (size_t)q0 * h->dim; float* out = h->scores + (size_t)q0 * h->n;
      switch (std::min(8, Q - q0)) {
        case 1: launch_scan<1>(h, q, out, s); break;  case 2: launch_scan<2>(h, q, out, s); break;
        case 3: launch_scan<3>(h, q, out, s); break;  case 4: launch_scan<4>(h, q, out, s); break;
        case 5: launch_scan<5>(h, q, out, s); break;  case 6: launch_scan<6>(h, q, out, s); break;
        case 7: launch_scan<7>(h, q, out, s); break;  default: launch_scan<8>(h, q, out, s); break;
      }
    }
    CC_HIP(hipGetLastError());
    return;
  }
  const int blocks = (int)std::min<int64_t>((h->n + 3) / 4, 256 * 16);
  for (int q0 = 0; q0 < Q; q0 += 4) {
    const int nq = std::min(4, Q - q0);
    hipLaunchKernelGGL(scores_kernel<4>, dim3(blocks), dim3(256), 0, s, h->emb, h->q_dev + (size_t)q0 * h->dim, h->scores + (size_t)q0 * h->n,
                       (long)h->n, h->dim, nq);
  }
  CC_HIP(hipGetLastError());
}

}  // namespace

#define CC_API_BEGIN try {
#define CC_API_END                                                         \
  return 0; }                                                              \
  catch (const cc::Error& e) { cc::set_error(e.what()); return e.code; }   \
  catch (const std::exception& e) { cc::set_error(e.what()); return -1; }

extern "C" {

int cc_index_create_ex(cc_index** h, int dim, int64_t capacity, int device, int storage) {
  CC_API_BEGIN
  CC_CHECK(h && dim > 0 && dim % 4 == 0 && capacity > 0, "bad argument (dim must be a multiple of 4)");
  CC_CHECK(storage == F32 || storage == BF16, "index storage must be 0 (f32) or 2 (bf16)");
  CC_CHECK(storage == F32 || dim % 256 == 0, "a bf16 index needs dim % 256 == 0");
  int n = 0; CC_HIP(hipGetDeviceCount(&n));
  CC_CHECK(n > 0 && device >= 0 && device < n, "no such HIP device");
  CC_HIP(hipSetDevice(device));
  std::unique_ptr<cc_index> x(new cc_index());
  x->dim = dim; x->capacity = capacity; x->device = device; x->storage = storage;
  CC_HIP(hipMalloc((void**)&x->emb, (size_t)capacity * x->row_bytes() + 256));
  CC_HIP(hipMalloc((void**)&x->grp, (size_t)capacity * 4 + 256));
  x->stream = pool_stream_get(device);
  *h = x.release();
  CC_API_END
}

int cc_index_create(cc_index** h, int dim, int64_t capacity, int device) { return cc_index_create_ex(h, dim, capacity, device, F32); }

// Append n rows (f32 in; rounded to bf16 for a bf16 index) with their group ids (null: group 0).  The matrix grows
// geometrically when it is full (new allocation, device-to-device copy, swap), so callers never see a capacity error.
int cc_index_add_grouped(cc_index* h, const float* emb, int64_t n, int on_device, const int32_t* groups) {
  CC_API_BEGIN
  CC_CHECK(h && (emb || n == 0) && n >= 0, "bad argument");
  CC_HIP(hipSetDevice(h->device));
  CC_HIP(hipStreamSynchronize(h->stream));            // a search on our stream may still be reading the matrix
  if (h->n + n > h->capacity) {
    int64_t cap = h->capacity;
    while (cap < h->n + n) cap *= 2;
    CC_CHECK((size_t)cap * h->row_bytes() < ((size_t)1 << 40), "index would exceed 1 TiB");
    float* emb2 = nullptr; int* grp2 = nullptr;
    CC_HIP(hipMalloc((void**)&emb2, (size_t)cap * h->row_bytes() + 256));
    if (hipMalloc((void**)&grp2, (size_t)cap * 4 + 256) != hipSuccess) { hipFree(emb2); throw cc::Error(-12, "out of device memory growing the index"); }
    CC_HIP(hipMemcpy(emb2, h->emb, (size_t)h->n * h->row_bytes(), hipMemcpyDeviceToDevice));
    CC_HIP(hipMemcpy(grp2, h->grp, (size_t)h->n * 4, hipMemcpyDeviceToDevice));
    hipFree(h->emb); hipFree(h->grp);
    h->emb = emb2; h->grp = grp2; h->capacity = cap;
  }
  if (n) {
    char* dst = reinterpret_cast<char*>(h->emb) + (size_t)h->n * h->row_bytes();
    if (h->storage == F32) {
      CC_HIP(hipMemcpy(dst, emb, (size_t)n * h->dim * 4, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
    } else {
      const size_t cnt = (size_t)n * h->dim;
      const float* src = emb; float* tmp = nullptr;
      if (!on_device) { CC_HIP(hipMalloc((void**)&tmp, cnt * 4)); CC_HIP(hipMemcpy(tmp, emb, cnt * 4, hipMemcpyHostToDevice)); src = tmp; }
      // Device-resident rows may still be being written by the caller's stream (torch's, not ours): the conversion kernel runs on
      // the index's private non-blocking stream, which nothing orders behind that producer.  The f32 path is ordered by its
      // blocking hipMemcpy; here the producer is waited for explicitly (add is not a hot path).
      else CC_HIP(hipDeviceSynchronize());
      hipLaunchKernelGGL(to_bf16_kernel, dim3((unsigned)std::min<size_t>((cnt + 255) / 256, 65535)), dim3(256), 0, h->stream, src, reinterpret_cast<uint16_t*>(dst), cnt);
      CC_HIP(hipStreamSynchronize(h->stream));
      if (tmp) hipFree(tmp);
    }
    if (groups) CC_HIP(hipMemcpy(h->grp + h->n, groups, (size_t)n * 4, hipMemcpyHostToDevice));
    else CC_HIP(hipMemset(h->grp + h->n, 0, (size_t)n * 4));
  }
  h->n += n;
  CC_API_END
}

int cc_index_add(cc_index* h, const float* emb, int64_t n, int on_device) { return cc_index_add_grouped(h, emb, n, on_device, nullptr); }

int cc_index_info(cc_index* h, int64_t* capacity, int* storage, int* dim) {
  CC_API_BEGIN
  CC_CHECK(h, "null argument");
  if (capacity) *capacity = h->capacity;
  if (storage) *storage = h->storage;
  if (dim) *dim = h->dim;
  CC_API_END
}

int cc_index_size(cc_index* h, int64_t* n) {
  CC_API_BEGIN
  CC_CHECK(h && n, "null argument");
  *n = h->n;
  CC_API_END
}

int cc_index_scores(cc_index* h, const float* q, int Q, float* scores, int on_device, void* stream) {
  CC_API_BEGIN
  CC_CHECK(h && q && scores && Q > 0, "bad argument");
  CC_HIP(hipSetDevice(h->device));
  if (h->n == 0) return 0;
  hipStream_t s = stream ? (hipStream_t)stream : h->stream;
  upload_queries(h, q, Q, on_device, s);
  compute_scores(h, Q, s);
  CC_HIP(hipMemcpyAsync(scores, h->scores, (size_t)Q * h->n * 4, on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, s));
  if (!on_device) CC_HIP(hipStreamSynchronize(s));
  CC_API_END
}

// Top-k among the rows whose group is allowed (allowed: n_groups bytes on the HOST, one per group id, non-zero = keep;
// null = every row).  Rows of a group id >= n_groups are filtered out.  Everything else as cc_index_search.
int cc_index_search_groups(cc_index* h, const float* q, int Q, int k, const uint8_t* allowed, int n_groups, int32_t* idx, float* score,
                           int on_device, void* stream) {
  CC_API_BEGIN
  CC_CHECK(h && q && idx && score && Q > 0, "bad argument");
  CC_CHECK(k > 0 && k <= kMaxK, "k must be in [1, 1024]");
  CC_HIP(hipSetDevice(h->device));
  hipStream_t s = stream ? (hipStream_t)stream : h->stream;
  const size_t ob = (size_t)Q * k;
  if (h->out_cap < ob) {
    if (h->idx_dev) hipFree(h->idx_dev);
    if (h->sc_dev) hipFree(h->sc_dev);
    CC_HIP(hipMalloc((void**)&h->idx_dev, ob * 4 + 256)); CC_HIP(hipMalloc((void**)&h->sc_dev, ob * 4 + 256)); h->out_cap = ob;
  }
  const int nchunk = (int)std::max<int64_t>(1, (h->n + kChunk - 1) / kChunk);
  ensure((void**)&h->cand, &h->cand_cap, (size_t)Q * nchunk * k * 8);
  // Host caller with a small result (the interactive case: one query, k = 100): the merge kernel writes (id, score) into
  // pinned host memory and the only host-side wait is the stream synchronize - no pageable copies in either direction.
  const size_t qbytes = ((size_t)Q * h->dim * 4 + 255) & ~(size_t)255;
  const bool zero_copy = !on_device && ob * 8 <= kPinResultBytes;
  if (zero_copy) ensure_pinned(h, qbytes + ob * 8);
  int* idx_out = zero_copy ? reinterpret_cast<int*>(h->pin + qbytes) : h->idx_dev;
  float* sc_out = zero_copy ? reinterpret_cast<float*>(h->pin + qbytes + ob * 4) : h->sc_dev;
  const unsigned char* allowed_dev = nullptr;
  if (allowed) {
    CC_CHECK(n_groups > 0 && n_groups <= (1 << 24), "bad group count");
    // the kernel indexes the bitmap with the row's group id: size it for every id the caller may have used (ids are < 2^24)
    if (h->allowed_cap < n_groups) {
      if (h->allowed) hipFree(h->allowed);
      h->allowed = nullptr; h->allowed_cap = 0;
      int cap = 1024; while (cap < n_groups) cap *= 2;
      CC_HIP(hipMalloc((void**)&h->allowed, (size_t)cap)); h->allowed_cap = cap;
    }
    CC_HIP(hipMemsetAsync(h->allowed, 0, (size_t)h->allowed_cap, s));
    CC_HIP(hipMemcpyAsync(h->allowed, allowed, (size_t)n_groups, hipMemcpyHostToDevice, s));
    allowed_dev = h->allowed;
  }
  if (h->n > 0) {
    upload_queries(h, q, Q, on_device, s);
    compute_scores(h, Q, s);
    hipLaunchKernelGGL(topk_stage1, dim3(nchunk, Q), dim3(1024), 0, s, h->scores, (long)h->n, k, h->cand, h->grp, allowed_dev, h->allowed_cap);
  } else {
    CC_HIP(hipMemsetAsync(h->cand, 0, (size_t)Q * nchunk * k * 8, s));
  }
  hipLaunchKernelGGL(topk_stage2, dim3(Q), dim3(1024), 0, s, h->cand, (long)nchunk * k, k, idx_out, sc_out);
  CC_HIP(hipGetLastError());
  if (zero_copy) {
    CC_HIP(hipStreamSynchronize(s));
    memcpy(idx, idx_out, ob * 4);
    memcpy(score, sc_out, ob * 4);
  } else {
    const hipMemcpyKind kind = on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
    CC_HIP(hipMemcpyAsync(idx, h->idx_dev, ob * 4, kind, s));
    CC_HIP(hipMemcpyAsync(score, h->sc_dev, ob * 4, kind, s));
    if (!on_device) CC_HIP(hipStreamSynchronize(s));
  }
  CC_API_END
}

int cc_index_search(cc_index* h, const float* q, int Q, int k, int32_t* idx, float* score, int on_device, void* stream) {
  return cc_index_search_groups(h, q, Q, k, nullptr, 0, idx, score, on_device, stream);
}

void cc_index_destroy(cc_index* h) {
  if (!h) return;
  hipSetDevice(h->device);
  if (h->stream) hipStreamSynchronize(h->stream);
  for (void* p : {(void*)h->emb, (void*)h->grp, (void*)h->allowed, (void*)h->q16, (void*)h->q_dev, (void*)h->scores, (void*)h->cand, (void*)h->idx_dev, (void*)h->sc_dev}) if (p) hipFree(p);
  if (h->pin) hipHostFree(h->pin);
  pool_stream_put(h->device, h->stream);                  // parked, never destroyed (kernels.h)
  delete h;
}

}  // extern "C"

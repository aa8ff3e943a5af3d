// Shared host/device helpers for libclearcam_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <map>
#include <memory>
#include <cstdlib>
#include <string>
#include <stdexcept>

namespace cc {

enum DType : int { F32 = 0, F16 = 1, BF16 = 2 };
// C-ABI dtype 3 ("f16s"): f16 activations, every conv weight carried as TWO f16 planes W = W_hi + W_lo (ConvP::split). Storage type F16.
constexpr int F16S = 3;
// C-ABI dtype 4 ("f16h"): the low plane only where it is needed - the 1x1 convs of the detector's backbone (blocks 0-9) and the stem conv;
// every other conv carries one f16 plane with controlled rounding, which balances a 3x3 filter's nine taps against each other and has
// nothing to balance in a 1x1 ("Which layers need the low plane").
constexpr int F16H = 4;
// C-ABI dtype 5 ("f16c", round 5): ONE f16 plane everywhere except the stem conv (two planes), the 1x1 convs' weights rounded with the
// errors steered by the second moments of their own inputs on a few calibration frames (calibrate.hip; cc_yolo_calibrate): the frame rate of
// plain f16, the tolerance of "f16h" on inputs like the calibration frames.
constexpr int F16C = 5;
inline int storage_dtype(int dt) { return dt == F16S || dt == F16H || dt == F16C ? (int)F16 : dt; }

inline size_t dtype_size(int dt) { return dt == F32 ? 4 : 2; }

// ---- error plumbing -------------------------------------------------------------------------
void set_error(const std::string& msg);          // api.cpp (thread-local)
struct Error : std::runtime_error { int code; Error(int c, const std::string& m) : std::runtime_error(m), code(c) {} };

#define CC_HIP(expr)                                                                              \
  do {                                                                                            \
    hipError_t _e = (expr);                                                                       \
    if (_e != hipSuccess)                                                                         \
      throw cc::Error(-5, std::string(#expr) + ": " + hipGetErrorString(_e) + " @" + __FILE__ + ":" + std::to_string(__LINE__)); \
  } while (0)

#define CC_CHECK(cond, msg)                                                                       \
  do {                                                                                            \
    if (!(cond)) throw cc::Error(-22, std::string(msg) + " [" #cond "] @" + __FILE__ + ":" + std::to_string(__LINE__)); \
  } while (0)

// ---- 16-bit storage types -------------------------------------------------------------------
struct bf16_t { uint16_t v; };
typedef _Float16 f16_t;

template <class T> struct TypeTag;
template <> struct TypeTag<float>  { static constexpr int dt = F32; };
template <> struct TypeTag<f16_t>  { static constexpr int dt = F16; };
template <> struct TypeTag<bf16_t> { static constexpr int dt = BF16; };

__host__ __device__ inline float bf16_bits_to_f32(uint16_t b) {
  union { uint32_t u; float f; } x; x.u = (uint32_t)b << 16; return x.f;
}
__host__ __device__ inline uint16_t f32_to_bf16_bits(float f) {   // round-to-nearest-even
  union { uint32_t u; float f; } x; x.f = f;
  uint32_t u = x.u;
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

template <class T> __host__ __device__ inline float to_f32(T v);
template <> __host__ __device__ inline float to_f32<float>(float v) { return v; }
template <> __host__ __device__ inline float to_f32<f16_t>(f16_t v) { return (float)v; }
template <> __host__ __device__ inline float to_f32<bf16_t>(bf16_t v) { return bf16_bits_to_f32(v.v); }

template <class T> __host__ __device__ inline T from_f32(float f);
template <> __host__ __device__ inline float from_f32<float>(float f) { return f; }
template <> __host__ __device__ inline f16_t from_f32<f16_t>(float f) { return (f16_t)f; }
template <> __host__ __device__ inline bf16_t from_f32<bf16_t>(float f) {
  bf16_t r;
#if defined(__HIP_DEVICE_COMPILE__)
  r.v = __builtin_bit_cast(uint16_t, (__bf16)f);      // v_cvt_pk_bf16_f32 (RNE) instead of ~7 integer VALU ops
#else
  r.v = f32_to_bf16_bits(f);
#endif
  return r;
}

// two f32 -> one dword of two storage-dtype values (low half = a): one v_cvt_pk_* on gfx950
typedef float cc_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 cc_bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 cc_f16x2 __attribute__((ext_vector_type(2)));
template <class T> __device__ inline uint32_t pack2(float a, float b);
template <> __device__ inline uint32_t pack2<bf16_t>(float a, float b) {
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(cc_f32x2{a, b}, cc_bf16x2));
}
template <> __device__ inline uint32_t pack2<f16_t>(float a, float b) {
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(cc_f32x2{a, b}, cc_f16x2));
}

// host-side conversion of an f32 array into the storage dtype
void convert_f32_to(int dt, const float* src, void* dst, size_t n);

// ---- tensor views (NHWC activations with channel stride) --------------------------------------
// A Src is one channel range of a conv's input: a channel slice [coff, coff+C) of an NHWC buffer whose
// pixels hold `cstride` channels.  shift=1 reads the buffer as if nearest-upsampled x2 (Upsample,
// detection/yolov9.py:285-292 folded into the consumer's loader).  shift=-1 reads its 2x2 stride-1 AVERAGE (ADown's
// avg_pool2d(2, 1, 0), :45, folded into the stride-2 conv's loader: conv_adown.hip; the conv's Hin x Win are then (H-1) x (W-1)).
struct Src {
  const void* ptr; int H, W; int cstride, coff, C; int shift;
};

// One convolution / linear layer = implicit GEMM  out[m][n] = act(sum_k X[m][k] W[n][k] + bias[n]) (+res)
//   m = (b, ho, wo) output pixel, n = output channel, k = (tap r,s ; channel c) with c over s0 then s1.
struct ConvP {
  Src s0, s1;
  int B, Hin, Win, Cin;          // logical input dims (after shift), Cin = s0.C + s1.C
  int Ho, Wo, Cout;
  int ks, stride, pad;
  int Ktot;                      // ks*ks*Cin (split weights: 2*ks*ks*Cin)
  int Kw;                        // weight row stride in elements (>= Ktot, zero padded to a multiple of 64)
  const void* w;                 // [Cout][Kw], storage dtype
  const float* bias;             // [Cout] or null
  void* out; int out_cstride, out_coff; int out_f32;
  const void* res; int res_cstride, res_coff; int res_f32;   // optional residual (same pixel grid as out)
  int act;                       // 0 none, 1 SiLU, 2 tanh-GELU, 3 PReLU (x > 0 ? x : slope[channel] * x), 4 ReLU applied AFTER the residual add
  const float* slope;            // [Cout] PReLU slopes (act 3), else null
  // Split weights (C-ABI dtype "f16s"): W * 2^e = W_hi + W_lo, two f16 planes per filter tap - the weight row is
  // [tap 0: hi(Cin) | lo(Cin)] [tap 1: hi | lo] ..., i.e. the K walk visits every tap's input channels twice (2 ks^2 "virtual taps")
  // and both products land in the same f32 accumulator: ~22 significant weight bits at twice the MFMA count, activation traffic
  // from HBM unchanged.  Ktot = 2 ks^2 Cin.  oscale = 2^-e undoes the per-layer power-of-two scale that keeps W_lo a NORMAL f16
  // number (exact: the epilogue computes fma(acc, oscale, bias)).  split = 0: oscale is ignored (taken as 1).
  int split; float oscale;
  int variant;                   // kernel choice: 0 auto; tests force 1 direct, 2 generic MFMA, 3 halo-resident 3x3, 4 weights-stationary 3x3, 5 single-barrier schedule with 256x256 tiles, 6 the same with 128x128 tiles, 7 eight-wave two-group 256x256 kernel, 8 wave-autonomous narrow 3x3, 9 few-tile configuration (narrow channel tiles, 3-4 LDS stages), 10 weights-resident streaming 1x1 (conv_stream.hip)
};

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) and the CU count are per DEVICE: a launcher's one-time set-up is keyed by the current
// device ordinal (a process may hold handles on several GPUs), not by a process-wide flag.
struct PerDevice {
  bool done[64] = {}; int cus[64] = {};
  int index() const { int d = 0; if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 64) d = -1; return d; }
  // true once per device (always true for an ordinal outside the table: the set-up is idempotent)
  bool first(int d) { if (d < 0) return true; const bool f = !done[d]; done[d] = true; return f; }
  int cu_count(int d) {
    if (d >= 0 && cus[d]) return cus[d];
    int dev = 0; hipDeviceProp_t pr;
    CC_HIP(hipGetDevice(&dev)); CC_HIP(hipGetDeviceProperties(&pr, dev));
    if (d >= 0) cus[d] = pr.multiProcessorCount;
    return pr.multiProcessorCount;
  }
};

// Plan cache of a model handle: one plan (buffers + captured hipGraph) per input shape, bounded.  A service that sees many
// batch sizes would otherwise keep a multi-GB arena for each of them forever.  At most cap() plans live at once
// (CLEARCAM_MAX_PLANS, default 16); inserting into a full cache drains the handle's stream and drops the least recently
// used plan.  `on_evict` lets the owner forget raw pointers to it.
template <class Key, class PlanT>
struct PlanCache {
  struct Slot { std::unique_ptr<PlanT> plan; unsigned long long used; };
  std::map<Key, Slot> slots;
  unsigned long long tick = 0;
  static size_t cap() {
    static size_t c = 0;
    if (!c) { const char* e = getenv("CLEARCAM_MAX_PLANS"); const long v = e ? atol(e) : 16; c = (size_t)(v < 1 ? 1 : v); }
    return c;
  }
  PlanT* find(const Key& k) {
    auto it = slots.find(k);
    if (it == slots.end()) return nullptr;
    it->second.used = ++tick;
    return it->second.plan.get();
  }
  template <class OnEvict>
  PlanT* insert(const Key& k, std::unique_ptr<PlanT> p, hipStream_t stream, OnEvict on_evict) {
    while (slots.size() >= cap()) {
      auto victim = slots.begin();
      for (auto it = slots.begin(); it != slots.end(); ++it) if (it->second.used < victim->second.used) victim = it;
      CC_HIP(hipStreamSynchronize(stream));              // its graph may still be running
      on_evict(victim->second.plan.get());
      slots.erase(victim);
    }
    PlanT* raw = p.get();
    slots[k] = Slot{std::move(p), ++tick};
    return raw;
  }
  PlanT* insert(const Key& k, std::unique_ptr<PlanT> p, hipStream_t stream) { return insert(k, std::move(p), stream, [](PlanT*) {}); }
  void clear() { slots.clear(); }
  size_t size() const { return slots.size(); }
};

}  // namespace cc

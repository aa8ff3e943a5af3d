// AdaFace IR-50 face embedder behind the C ABI: stands behind `ADAFACE.__call__` (models/adaface.py:61-95), which
// clearcam calls as `object_finder.adaface(Tensor(face_img))` (clearcam.py:674,1236) on a 112x112 aligned face.
//
// Same engine as the detector: NHWC activations in the storage dtype, every convolution through launch_conv (MFMA
// implicit GEMM), one launch list per batch size captured into a hipGraph.  What is specific to this graph:
//  * inference-mode BatchNorm AFTER a convolution (bn0, res_layer1, res_layer2, shortcut_layer1) is folded into that
//    convolution's weights and bias on the host;  BatchNorm BEFORE a zero-padded convolution (res_layer0) cannot be
//    folded (the padding is applied after it), so it is one per-channel affine pass (affine_kernel, HBM-bound);
//  * PReLU is a conv epilogue (act 3, per-channel slope);  the residual add is the second conv's epilogue;
//  * the identity shortcut of a stride-2 block, MaxPool2d(1, 2), is a k=1 s=2 pooling launch;
//  * bn (512) -> flatten -> linear 25088->512 -> bn2: the linear's columns are permuted from the reference's (C,H,W)
//    flatten order to NHWC, bn2 (no affine) is folded into its rows; bn stays an affine pass;  then x / ||x||_2.
#include <cmath>
#include <map>
#include <memory>
#include <string>
#include <vector>
#include <cstring>
#include "net_common.h"
#include "../../include/clearcam_hip.h"

using namespace cc;

namespace {

constexpr float kEps = 1e-5f;
constexpr int kRes = 112;
const int kBlocks[24][3] = {{64, 64, 2}, {64, 64, 1}, {64, 64, 1}, {64, 128, 2}, {128, 128, 1}, {128, 128, 1}, {128, 128, 1}, {128, 256, 2},
                            {256, 256, 1}, {256, 256, 1}, {256, 256, 1}, {256, 256, 1}, {256, 256, 1}, {256, 256, 1}, {256, 256, 1},
                            {256, 256, 1}, {256, 256, 1}, {256, 256, 1}, {256, 256, 1}, {256, 256, 1}, {256, 256, 1},
                            {256, 512, 2}, {512, 512, 1}, {512, 512, 1}};   // models/adaface.py:58

struct Affine { float* scale = nullptr; float* shift = nullptr; int c = 0; };
struct FBlock { Affine pre; PConv c0, c1, sc; int cin, depth, stride; };

struct AffineP { const void* x; void* y; const float* scale; const float* shift; long n; int C; };
struct FacePreP { const void* img; int img_f32; void* out; int B, cpad; };

struct FOp { int kind; ConvP conv; PoolP pool; AffineP af; FacePreP pre; NormP nm; };   // 0 conv, 1 pool, 2 affine, 3 preprocess, 4 l2norm

struct FPlan {
  int B = 0;
  std::vector<void*> allocs;
  std::vector<FOp> ops;
  void* in_dev = nullptr; float* out_dev = nullptr;
  hipGraphExec_t exec = nullptr;
  ~FPlan() { if (exec) hipGraphExecDestroy(exec); for (void* p : allocs) hipFree(p); }
  char* alloc(size_t bytes) { void* p = nullptr; CC_HIP(hipMalloc(&p, bytes + 256)); allocs.push_back(p); return (char*)p; }
};

// y = x * scale[c] + shift[c] over NHWC (inference BatchNorm in front of a zero-padded conv), 16 bytes per thread
template <class T>
__global__ __launch_bounds__(256) void affine_kernel(const AffineP p) {
  constexpr int E = 16 / (int)sizeof(T);
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i * E >= p.n) return;
  const int c0 = (int)((i * E) % p.C);
  uint4 u = reinterpret_cast<const uint4*>(p.x)[i];
  T* t = reinterpret_cast<T*>(&u);
#pragma unroll
  for (int e = 0; e < E; ++e) t[e] = from_f32<T>(to_f32<T>(t[e]) * p.scale[c0 + e] + p.shift[c0 + e]);
  reinterpret_cast<uint4*>(p.y)[i] = u;
}

// ((x[:, :, ::-1] / 255) - 0.5) / 0.5, HWC -> NHWC with the 3 channels zero-padded to cpad (models/adaface.py:81-82)
template <class T>
__global__ __launch_bounds__(256) void face_pre_kernel(const FacePreP p) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)p.B * kRes * kRes) return;
  float v[3];
  for (int c = 0; c < 3; ++c) {
    const float raw = p.img_f32 ? reinterpret_cast<const float*>(p.img)[i * 3 + (2 - c)] : (float)reinterpret_cast<const uint8_t*>(p.img)[i * 3 + (2 - c)];
    v[c] = __fdiv_rn(__fdiv_rn(raw, 255.0f) - 0.5f, 0.5f);
  }
  T* o = reinterpret_cast<T*>(p.out) + i * p.cpad;
  for (int c = 0; c < p.cpad; ++c) o[c] = from_f32<T>(c < 3 ? v[c] : 0.f);
}

}  // namespace

struct cc_face {
  int dtype = BF16, device = 0;
  hipStream_t stream = nullptr;
  std::map<std::string, HostTensor> host;
  std::vector<void*> wallocs;
  bool finalized = false;
  PConv conv0; std::vector<FBlock> blocks; Affine bn_final; PConv linear;
  PlanCache<int, FPlan> plans;
};

namespace {

const HostTensor& need(cc_face* h, const std::string& name) {
  auto it = h->host.find(name);
  CC_CHECK(it != h->host.end(), "missing parameter " + name);
  return it->second;
}
float* upload(cc_face* h, const std::vector<float>& v) { return upload_f32(h->wallocs, v); }
// inference BatchNorm as y = x * s + t
void bn_st(cc_face* h, const std::string& p, int c, bool affine, std::vector<float>& s, std::vector<float>& t) {
  const HostTensor& mean = need(h, p + ".running_mean"); const HostTensor& var = need(h, p + ".running_var");
  CC_CHECK((int)mean.data.size() == c && (int)var.data.size() == c, p + ": BatchNorm size");
  s.resize(c); t.resize(c);
  for (int i = 0; i < c; ++i) {
    const float inv = 1.0f / std::sqrt(var.data[i] + kEps);
    const float w = affine ? need(h, p + ".weight").data[i] : 1.0f, b = affine ? need(h, p + ".bias").data[i] : 0.0f;
    s[i] = inv * w; t[i] = b - mean.data[i] * inv * w;
  }
}
Affine make_affine(cc_face* h, const std::string& p, int c) {
  std::vector<float> s, t; bn_st(h, p, c, true, s, t);
  return Affine{upload(h, s), upload(h, t), c};
}
// OIHW conv (no bias) followed by BatchNorm `bn`: scale folded into the rows, shift becomes the bias; optional PReLU slopes
PConv make_conv(cc_face* h, const std::string& wname, const std::string& bn, const std::string& prelu, int cin_pad = 0) {
  const HostTensor& w = need(h, wname);
  std::vector<float> s, t; bn_st(h, bn, (int)w.shape[0], true, s, t);
  PConv pc = pack_conv(h->dtype, h->wallocs, w, 1, s, t, cin_pad);
  if (!prelu.empty()) { const HostTensor& a = need(h, prelu); CC_CHECK((int)a.data.size() == pc.cout, prelu + " size"); pc.slope = upload(h, a.data); }
  return pc;
}

void run_ops(cc_face* h, FPlan* P, hipStream_t s) {
  for (const FOp& op : P->ops) {
    switch (op.kind) {
      case 0: launch_conv(h->dtype, op.conv, s); break;
      case 1: launch_pool(h->dtype, op.pool, s); break;
      case 2: {
        const int E = h->dtype == F32 ? 4 : 8;
        const unsigned blocks = (unsigned)((op.af.n / E + 255) / 256);
        if (h->dtype == F32) hipLaunchKernelGGL(affine_kernel<float>, dim3(blocks), dim3(256), 0, s, op.af);
        else if (h->dtype == F16) hipLaunchKernelGGL(affine_kernel<f16_t>, dim3(blocks), dim3(256), 0, s, op.af);
        else hipLaunchKernelGGL(affine_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, op.af);
        break;
      }
      case 3: {
        const unsigned blocks = (unsigned)(((long)op.pre.B * kRes * kRes + 255) / 256);
        if (h->dtype == F32) hipLaunchKernelGGL(face_pre_kernel<float>, dim3(blocks), dim3(256), 0, s, op.pre);
        else if (h->dtype == F16) hipLaunchKernelGGL(face_pre_kernel<f16_t>, dim3(blocks), dim3(256), 0, s, op.pre);
        else hipLaunchKernelGGL(face_pre_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, op.pre);
        break;
      }
      default: launch_l2norm(op.nm, s);
    }
  }
  CC_HIP(hipGetLastError());
}

FPlan* get_plan(cc_face* h, int B, int img_f32) {
  const int key = B * 2 + img_f32;
  if (FPlan* hit = h->plans.find(key)) return hit;
  std::unique_ptr<FPlan> P(new FPlan()); P->B = B;
  const size_t es = dtype_size(h->dtype);
  auto act_buf = [&](int H, int W, int C) { return P->alloc((size_t)B * H * W * C * es); };
  auto add_conv = [&](const PConv& pc, const void* x, int H, int W, int stride, void* out, int act, const void* res) {
    FOp op{}; op.kind = 0; op.conv = conv_params(pc, x, B, H, W, stride, out, pc.cout, 0, act, res, pc.cout); P->ops.push_back(op);
  };
  auto add_affine = [&](const Affine& a, const void* x, void* y, long n) {
    FOp op{}; op.kind = 2; op.af = AffineP{x, y, a.scale, a.shift, n, a.c}; P->ops.push_back(op);
  };
  P->in_dev = P->alloc((size_t)B * kRes * kRes * 3 * (img_f32 ? 4 : 1));
  P->out_dev = (float*)P->alloc((size_t)B * 512 * 4);
  const int cp = h->conv0.cin;
  char* x0 = act_buf(kRes, kRes, cp);
  { FOp op{}; op.kind = 3; op.pre = FacePreP{P->in_dev, img_f32, x0, B, cp}; P->ops.push_back(op); }
  int H = kRes;
  char* x = act_buf(H, H, 64);
  add_conv(h->conv0, x0, H, H, 1, x, 3, nullptr);                          // conv0 + bn0 + PReLU
  for (const FBlock& b : h->blocks) {
    const int Ho = (H + 2 - 3) / b.stride + 1;
    const char* sc = x;
    if (b.cin != b.depth) { char* s = act_buf(Ho, Ho, b.depth); add_conv(b.sc, x, H, H, b.stride, s, 0, nullptr); sc = s; }
    else if (b.stride != 1) {                                              // MaxPool2d(1, stride): subsample
      char* s = act_buf(Ho, Ho, b.depth);
      FOp op{}; op.kind = 1;
      op.pool = PoolP{x, b.cin, 0, s, b.depth, 0, B, H, H, b.cin, Ho, Ho, 1, b.stride, 0, 1};
      P->ops.push_back(op); sc = s;
    }
    char* t0 = act_buf(H, H, b.cin);
    add_affine(b.pre, x, t0, (long)B * H * H * b.cin);                     // res_layer0 (BatchNorm before the padded conv)
    char* t1 = act_buf(H, H, b.depth);
    add_conv(b.c0, t0, H, H, 1, t1, 3, nullptr);                           // conv_layer0 + res_layer1 + PReLU
    char* y = act_buf(Ho, Ho, b.depth);
    add_conv(b.c1, t1, H, H, b.stride, y, 0, sc);                          // conv_layer1 + res_layer2, + shortcut
    x = y; H = Ho;
  }
  CC_CHECK(H == 7, "unexpected final feature size");
  char* xb = act_buf(7, 7, 512);
  add_affine(h->bn_final, x, xb, (long)B * 49 * 512);
  {                                                                         // linear (+ bn2 folded) on the NHWC-flattened features
    FOp op{}; op.kind = 0;
    op.conv = gemm_params(xb, 49 * 512, B, 49 * 512, h->linear.w, h->linear.kw, h->linear.bias, 512, P->out_dev, 512, 1, 0, nullptr, 0, 0);
    P->ops.push_back(op);
  }
  { FOp op{}; op.kind = 4; op.nm = NormP{P->out_dev, B, 512, 0.0f}; P->ops.push_back(op); }
  FPlan* pp = P.get();
  P->exec = capture_graph(h->stream, [&]() { run_ops(h, pp, h->stream); });
  return h->plans.insert(key, std::move(P), h->stream);
}

}  // namespace

#define CC_API_BEGIN try {
#define CC_API_END                                                         \
  return 0; }                                                              \
  catch (const cc::Error& e) { cc::set_error(e.what()); return e.code; }   \
  catch (const std::exception& e) { cc::set_error(e.what()); return -1; }

extern "C" {

int cc_face_create(cc_face** h, int dtype, int device) {
  CC_API_BEGIN
  CC_CHECK(h, "null argument");
  CC_CHECK(dtype >= 0 && dtype <= 2, "dtype must be 0 (f32), 1 (f16) or 2 (bf16)");
  int n = 0; CC_HIP(hipGetDeviceCount(&n));
  CC_CHECK(n > 0 && device >= 0 && device < n, "no such HIP device");
  CC_HIP(hipSetDevice(device));
  std::unique_ptr<cc_face> f(new cc_face());
  f->dtype = dtype; f->device = device;
  f->stream = pool_stream_get(device);
  *h = f.release();
  CC_API_END
}

int cc_face_load(cc_face* h, const char* name, const float* data, const int64_t* shape, int ndim) {
  CC_API_BEGIN
  CC_CHECK(h && name && data && shape && ndim >= 0 && ndim <= 4, "bad argument");
  CC_CHECK(!h->finalized, "cc_face_load after cc_face_finalize");
  HostTensor t; size_t n = 1;
  for (int i = 0; i < ndim; ++i) { t.shape.push_back(shape[i]); n *= (size_t)shape[i]; }
  t.data.assign(data, data + n);
  h->host[name] = std::move(t);
  CC_API_END
}

int cc_face_finalize(cc_face* h) {
  CC_API_BEGIN
  CC_CHECK(h && !h->finalized, "bad handle");
  CC_HIP(hipSetDevice(h->device));
  const int E = h->dtype == F32 ? 4 : 8;
  h->conv0 = make_conv(h, "conv0.weight", "bn0", "prelu_weight", E);       // 3 input channels padded to one 16-byte chunk
  for (int i = 0; i < 24; ++i) {
    const std::string p = "body.list." + std::to_string(i) + ".";
    FBlock b; b.cin = kBlocks[i][0]; b.depth = kBlocks[i][1]; b.stride = kBlocks[i][2];
    b.pre = make_affine(h, p + "res_layer0", b.cin);
    b.c0 = make_conv(h, p + "conv_layer0.weight", p + "res_layer1", p + "prelu_weight");
    b.c1 = make_conv(h, p + "conv_layer1.weight", p + "res_layer2", "");
    if (b.cin != b.depth) b.sc = make_conv(h, p + "shortcut_layer0.weight", p + "shortcut_layer1", "");
    CC_CHECK(b.c0.cin == b.cin && b.c0.cout == b.depth && b.c1.cout == b.depth, p + ": conv shapes");
    h->blocks.push_back(b);
  }
  h->bn_final = make_affine(h, "bn", 512);
  {   // linear: reference flattens (C,H,W); activations here are (H,W,C).  bn2 (affine=False) folds into rows and bias.
    const HostTensor& w = need(h, "linear.weight"); const HostTensor& bias = need(h, "linear.bias");
    CC_CHECK(w.shape.size() == 2 && w.shape[0] == 512 && w.shape[1] == 512 * 49, "linear.weight must be (512, 25088)");
    std::vector<float> s, t; bn_st(h, "bn2", 512, false, s, t);
    const size_t K = 512 * 49;
    std::vector<float> packed(512 * K), b2(512);
    for (int n = 0; n < 512; ++n) {
      for (int c = 0; c < 512; ++c)
        for (int px = 0; px < 49; ++px) packed[(size_t)n * K + (size_t)px * 512 + c] = w.data[(size_t)n * K + (size_t)c * 49 + px] * s[n];
      b2[n] = bias.data[n] * s[n] + t[n];
    }
    PConv pc; pc.cin = (int)K; pc.cout = 512; pc.k = 1; pc.kw = (int)K;
    std::vector<char> tmp(packed.size() * dtype_size(h->dtype));
    convert_f32_to(h->dtype, packed.data(), tmp.data(), packed.size());
    CC_HIP(hipMalloc(&pc.w, tmp.size() + 256));
    CC_HIP(hipMemcpy(pc.w, tmp.data(), tmp.size(), hipMemcpyHostToDevice));
    h->wallocs.push_back(pc.w);
    pc.bias = upload(h, b2);
    h->linear = pc;
  }
  h->finalized = true;
  h->host.clear();
  CC_API_END
}

int cc_face_embed(cc_face* h, const void* faces, int B, int img_f32, int faces_on_device, float* out, int out_on_device, void* stream) {
  CC_API_BEGIN
  CC_CHECK(h && faces && out && B > 0, "bad argument");
  CC_CHECK(h->finalized, "cc_face_embed before cc_face_finalize");
  CC_HIP(hipSetDevice(h->device));
  FPlan* P = get_plan(h, B, img_f32 ? 1 : 0);
  hipStream_t s = h->stream;
  if (stream) {
    hipEvent_t e; CC_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    CC_HIP(hipEventRecord(e, (hipStream_t)stream)); CC_HIP(hipStreamWaitEvent(s, e, 0)); CC_HIP(hipEventDestroy(e));
  }
  const size_t nb = (size_t)B * kRes * kRes * 3 * (img_f32 ? 4 : 1);
  CC_HIP(hipMemcpyAsync(P->in_dev, faces, nb, faces_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s));
  CC_HIP(hipGraphLaunch(P->exec, s));
  const size_t ob = (size_t)B * 512 * 4;
  if (out_on_device) {
    CC_HIP(hipMemcpyAsync(out, P->out_dev, ob, hipMemcpyDeviceToDevice, s));
    if (stream) {
      hipEvent_t e; CC_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      CC_HIP(hipEventRecord(e, s)); CC_HIP(hipStreamWaitEvent((hipStream_t)stream, e, 0)); CC_HIP(hipEventDestroy(e));
    }
  } else {
    CC_HIP(hipMemcpyAsync(out, P->out_dev, ob, hipMemcpyDeviceToHost, s));
    CC_HIP(hipStreamSynchronize(s));
  }
  CC_API_END
}

void cc_face_destroy(cc_face* h) {
  if (!h) return;
  hipSetDevice(h->device);
  if (h->stream) hipStreamSynchronize(h->stream);
  h->plans.clear();
  for (void* p : h->wallocs) hipFree(p);
  pool_stream_put(h->device, h->stream);                  // parked, never destroyed (kernels.h)
  delete h;
}

}  // extern "C"

// BlazeFace face detector behind the C ABI: stands behind `BlazeFace.__call__(img)` (models/blazeface.py:165-192), which
// `ObjectFinder.img_to_face` calls on a 640x640 letterboxed crop (models/objects.py:253-255) -> (896,17) detections.
//
// The network is tiny (~60 MFLOP): every layer goes through launch_conv with dense weights - the depthwise 3x3 convs as
// block-diagonal [C][9C] matrices, a few MFLOP each - so the model is one more graph on the engine, not a second engine:
//  * pad(1,2) + 5x5 s2 conv and the stride-2 blocks' pad(0,2) + 3x3 s2 depthwise conv are the same kernels with an explicit
//    output size: the loaders read zeros outside the image, which is exactly asymmetric bottom/right padding;
//  * BlazeBlock = depthwise conv -> 1x1 conv whose epilogue adds the shortcut and then applies ReLU (act 4);  a stride-2
//    block's shortcut is a 2x2 max-pool written into the low channels of a zero-filled buffer (the channel zero-pad);
//  * the four heads write float32; one 1024-thread block then decodes the 896 anchors, orders them by score (bitonic sort
//    keyed on (score, index) = the stable descending order), applies the reference's overlap rule and maps boxes back.
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

constexpr int kIn = 256, kAnchors = 896;
const int kBlocks[31][3] = {{24, 24, 1}, {24, 24, 1}, {24, 24, 1}, {24, 24, 1}, {24, 24, 1}, {24, 24, 1}, {24, 24, 1}, {24, 24, 2},
                            {24, 24, 1}, {24, 24, 1}, {24, 24, 1}, {24, 24, 1}, {24, 24, 1}, {24, 24, 1}, {24, 24, 1}, {24, 48, 2},
                            {48, 48, 1}, {48, 48, 1}, {48, 48, 1}, {48, 48, 1}, {48, 48, 1}, {48, 48, 1}, {48, 48, 1}, {48, 96, 2},
                            {96, 96, 1}, {96, 96, 1}, {96, 96, 1}, {96, 96, 1}, {96, 96, 1}, {96, 96, 1}, {96, 96, 1}};   // blazeface.py:88-119

struct BBlock { PConv dw, pw; int cin, cout, stride; };

struct PostP {                 // decode (:194-226), postprocess (:228-238), back-map (:188-192)
  const float* r8; const float* r16; const float* c8; const float* c16;   // (16,16,32) (8,8,96) (16,16,2) (8,8,6) float32
  const float* anchors;        // (896,4) x_center, y_center, w, h
  float scale; int pad_top, pad_left;
  float* rows;                 // (896,17) scratch: decoded, unsorted
  float* out;                  // (896,17)
};

struct BOp { int kind; ConvP conv; PoolP pool; PreP pre; PostP post; };   // 0 conv, 1 pool, 2 preprocess, 3 decode+postprocess

struct BPlan {
  int H = 0, W = 0, f32 = 0;
  std::vector<void*> allocs;
  std::vector<BOp> ops;
  void* in_dev = nullptr; float* out_dev = nullptr;
  hipGraphExec_t exec = nullptr;
  ~BPlan() { if (exec) hipGraphExecDestroy(exec); for (void* p : allocs) hipFree(p); }
  char* alloc(size_t bytes, bool zero = false) {
    void* p = nullptr; CC_HIP(hipMalloc(&p, bytes + 256)); allocs.push_back(p);
    if (zero) CC_HIP(hipMemset(p, 0, bytes + 256));
    return (char*)p;
  }
};

__global__ __launch_bounds__(1024) void blaze_post_kernel(const PostP p) {
  __shared__ float box[kAnchors][4];                             // sorted boxes for the 896 x 896 overlap pass
  __shared__ float key[1024];
  __shared__ short ord[1024];
  const int i = threadIdx.x;
  if (i < kAnchors) {
    const float* raw = i < 512 ? p.r8 + (size_t)i * 16 : p.r16 + (size_t)(i - 512) * 16;
    const float sraw = i < 512 ? p.c8[i] : p.c16[i - 512];
    const float ax = p.anchors[i * 4], ay = p.anchors[i * 4 + 1], aw = p.anchors[i * 4 + 2], ah = p.anchors[i * 4 + 3];
    const float xc = raw[0] / 256.0f * aw + ax, yc = raw[1] / 256.0f * ah + ay;
    const float w = raw[2] / 256.0f * aw, h = raw[3] / 256.0f * ah;
    float row[17];
    row[0] = yc - h / 2.0f; row[1] = xc - w / 2.0f; row[2] = yc + h / 2.0f; row[3] = xc + w / 2.0f;
#pragma unroll
    for (int k = 0; k < 6; ++k) { row[4 + 2 * k] = raw[4 + 2 * k] / 256.0f * aw + ax; row[5 + 2 * k] = raw[5 + 2 * k] / 256.0f * ah + ay; }
    const float cl = fminf(fmaxf(sraw, -100.0f), 100.0f);
    const float s = 1.0f / (1.0f + expf(-cl));
    row[16] = s;
    const float m = s >= 0.85f ? 1.0f : 0.0f;                    // detections *= mask (:201)
#pragma unroll
    for (int k = 0; k < 17; ++k) p.rows[(size_t)i * 17 + k] = row[k] * m;
    key[i] = row[16] * m;
  } else key[i] = -1.0f;                                         // padding sorts behind every real row (scores are >= 0)
  ord[i] = (short)i;
  __syncthreads();
  // bitonic sort of 1024 (score desc, index asc): the order of a stable descending sort
  for (int k = 2; k <= 1024; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      const int ixj = i ^ j;
      if (ixj > i) {
        const bool up = (i & k) == 0;
        const float ka = key[i], kb = key[ixj]; const short oa = ord[i], ob = ord[ixj];
        const bool a_first = ka > kb || (ka == kb && oa < ob);   // a belongs before b
        if (up ? !a_first : a_first) { key[i] = kb; key[ixj] = ka; ord[i] = ob; ord[ixj] = oa; }
      }
      __syncthreads();
    }
  if (i < kAnchors) {
    const float* r = p.rows + (size_t)ord[i] * 17;
    box[i][0] = r[0]; box[i][1] = r[1]; box[i][2] = r[2]; box[i][3] = r[3];
  }
  __syncthreads();
  if (i < kAnchors) {
    const float x1 = box[i][0], y1 = box[i][1], x2 = box[i][2], y2 = box[i][3];
    const float area = (x2 - x1) * (y2 - y1);
    int hits = 0;
    for (int j = 0; j < i; ++j) {                                // triu(diagonal=1) of the (1,N,N) mask summed over axis 1: better-ranked rows j < i (:232-234)
      const float u1 = box[j][0], v1 = box[j][1], u2 = box[j][2], v2 = box[j][3];
      const float w = fmaxf(0.0f, fminf(x2, u2) - fmaxf(x1, u1)), h = fmaxf(0.0f, fminf(y2, v2) - fmaxf(y1, v1));
      const float inter = w * h;
      const float iou = inter / (area + (u2 - u1) * (v2 - v1) - inter);
      hits += iou > 0.3f ? 1 : 0;                                // NaN (0/0 between zeroed rows) compares false
    }
    const float* r = p.rows + (size_t)ord[i] * 17;
    const float m = (hits == 0 && r[16] >= 0.85f) ? 1.0f : 0.0f;
    float* o = p.out + (size_t)i * 17;
#pragma unroll
    for (int k = 0; k < 17; ++k) {
      float v = r[k] * m * 256.0f;
      if (k == 0 || k == 2) v -= (float)p.pad_top;
      if (k == 1 || k == 3) v -= (float)p.pad_left;
      o[k] = v / p.scale;                                        // every column, as the reference writes it (:192)
    }
  }
}

}  // namespace

struct cc_blaze {
  int dtype = BF16, device = 0;
  hipStream_t stream = nullptr;
  std::map<std::string, HostTensor> host;
  std::vector<void*> wallocs;
  bool finalized = false;
  PConv stem; std::vector<BBlock> blocks; PConv fdw, fpw, cls8, cls16, reg8, reg16;
  float* anchors = nullptr;
  PlanCache<std::vector<int>, BPlan> plans;
};

namespace {

const HostTensor& need(cc_blaze* h, const std::string& name) {
  auto it = h->host.find(name);
  CC_CHECK(it != h->host.end(), "missing parameter " + name);
  return it->second;
}
PConv make_conv(cc_blaze* h, const std::string& name, int groups, int cin_pad = 0) {
  const HostTensor& w = need(h, name + ".weight");
  return pack_conv(h->dtype, h->wallocs, w, groups, {}, need(h, name + ".bias").data, cin_pad);
}

void run_ops(cc_blaze* h, BPlan* P, hipStream_t s) {
  for (const BOp& op : P->ops) {
    if (op.kind == 0) launch_conv(h->dtype, op.conv, s);
    else if (op.kind == 1) launch_pool(h->dtype, op.pool, s);
    else if (op.kind == 2) launch_preprocess(h->dtype, op.pre, s);
    else hipLaunchKernelGGL(blaze_post_kernel, dim3(1), dim3(1024), 0, s, op.post);
  }
  CC_HIP(hipGetLastError());
}

template <class T> T* to_dev(BPlan* P, const std::vector<T>& v) {
  T* d = (T*)P->alloc(v.size() * sizeof(T));
  CC_HIP(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  return d;
}

BPlan* get_plan(cc_blaze* h, int H, int W, int f32) {
  const std::vector<int> key{H, W, f32};
  if (BPlan* hit = h->plans.find(key)) return hit;
  std::unique_ptr<BPlan> P(new BPlan()); P->H = H; P->W = W; P->f32 = f32;
  const size_t es = dtype_size(h->dtype);
  auto act = [&](int hh, int ww, int c, bool zero = false) { return P->alloc((size_t)hh * ww * c * es, zero); };
  auto add_conv = [&](const PConv& pc, const void* x, int hh, int ww, int stride, void* out, int out_f32, int actv, const void* res,
                      int pad = -1, int Ho = 0) {
    BOp op{}; op.kind = 0; op.conv = conv_params(pc, x, 1, hh, ww, stride, out, pc.cout, out_f32, actv, res, pc.cout, pad, Ho, Ho);
    P->ops.push_back(op);
  };
  // ---- preprocess geometry (:166-179)
  const double dscale = std::min(256.0 / (double)W, 256.0 / (double)H);     // Python floats are doubles: int(w0 * scale)
  const int new_w = (int)((double)W * dscale), new_h = (int)((double)H * dscale);
  const float scale = (float)dscale;
  CC_CHECK(new_w > 0 && new_h > 0 && new_w <= kIn && new_h <= kIn, "image too small / extreme aspect");
  const int pad_top = (kIn - new_h) / 2, pad_left = (kIn - new_w) / 2;
  std::vector<int> xlo, xhi, ylo, yhi; std::vector<float> xfr, yfr;
  axis_tables(W, new_w, xlo, xhi, xfr); axis_tables(H, new_h, ylo, yhi, yfr);
  P->in_dev = P->alloc((size_t)H * W * 3 * (f32 ? 4 : 1));
  P->out_dev = (float*)P->alloc((size_t)kAnchors * 17 * 4);
  const int cp = h->stem.cin;
  char* x0 = act(kIn, kIn, cp);
  {
    BOp op{}; op.kind = 2; PreP& q = op.pre;
    q.frames = P->in_dev; q.frame_f32 = f32; q.B = 1; q.H = H; q.W = W; q.nh = new_h; q.nw = new_w; q.pad_y = pad_top; q.pad_x = pad_left;
    q.Hn = kIn; q.Wn = kIn; q.xlo = to_dev(P.get(), xlo); q.xhi = to_dev(P.get(), xhi); q.xfr = to_dev(P.get(), xfr);
    q.ylo = to_dev(P.get(), ylo); q.yhi = to_dev(P.get(), yhi); q.yfr = to_dev(P.get(), yfr);
    q.out = x0; q.out_c = cp; q.flip = 0; q.div = 127.5f; q.sub = 1.0f; q.pad_val = -1.0f;      // x / 127.5 - 1 over the padded image
    P->ops.push_back(op);
  }
  int S = 128;
  char* x = act(S, S, 24);
  add_conv(h->stem, x0, kIn, kIn, 2, x, 0, 4, nullptr, 1, S);               // pad (1,2), 5x5 s2, ReLU
  for (const BBlock& b : h->blocks) {
    const int So = S / b.stride;
    char* t = act(So, So, b.cin);
    if (b.stride == 2) add_conv(b.dw, x, S, S, 2, t, 0, 0, nullptr, 0, So);  // pad (0,2) + depthwise 3x3 s2
    else add_conv(b.dw, x, S, S, 1, t, 0, 0, nullptr);
    const char* sc = x;
    if (b.stride == 2) {                                                    // max_pool2d(2,2) and the channel zero-pad
      char* s2 = act(So, So, b.cout, true);
      BOp op{}; op.kind = 1; op.pool = PoolP{x, b.cin, 0, s2, b.cout, 0, 1, S, S, b.cin, So, So, 2, 2, 0, 1};
      P->ops.push_back(op); sc = s2;
    }
    char* y = act(So, So, b.cout);
    add_conv(b.pw, t, So, So, 1, y, 0, 4, sc);                              // 1x1, + shortcut, ReLU
    x = y; S = So;
  }
  CC_CHECK(S == 16, "unexpected backbone output size");
  char* ft = act(8, 8, 96);
  add_conv(h->fdw, x, 16, 16, 2, ft, 0, 0, nullptr, 0, 8);
  char* hf = act(8, 8, 96);
  add_conv(h->fpw, ft, 8, 8, 1, hf, 0, 4, nullptr);
  float* c8 = (float*)P->alloc(16 * 16 * 2 * 4); float* c16 = (float*)P->alloc(8 * 8 * 6 * 4);
  float* r8 = (float*)P->alloc(16 * 16 * 32 * 4); float* r16 = (float*)P->alloc(8 * 8 * 96 * 4);
  add_conv(h->cls8, x, 16, 16, 1, c8, 1, 0, nullptr);
  add_conv(h->cls16, hf, 8, 8, 1, c16, 1, 0, nullptr);
  add_conv(h->reg8, x, 16, 16, 1, r8, 1, 0, nullptr);
  add_conv(h->reg16, hf, 8, 8, 1, r16, 1, 0, nullptr);
  { BOp op{}; op.kind = 3; op.post = PostP{r8, r16, c8, c16, h->anchors, scale, pad_top, pad_left, (float*)P->alloc((size_t)kAnchors * 17 * 4), P->out_dev}; P->ops.push_back(op); }
  BPlan* pp = P.get();
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

int cc_blaze_create(cc_blaze** h, int dtype, int device) {
  CC_API_BEGIN
  CC_CHECK(h, "null argument");
  CC_CHECK(dtype >= 0 && dtype <= 2, "dtype must be 0 (f32), 1 (f16) or 2 (bf16)");
  int n = 0; CC_HIP(hipGetDeviceCount(&n));
  CC_CHECK(n > 0 && device >= 0 && device < n, "no such HIP device");
  CC_HIP(hipSetDevice(device));
  std::unique_ptr<cc_blaze> b(new cc_blaze());
  b->dtype = dtype; b->device = device;
  b->stream = pool_stream_get(device);
  *h = b.release();
  CC_API_END
}

int cc_blaze_load(cc_blaze* h, const char* name, const float* data, const int64_t* shape, int ndim) {
  CC_API_BEGIN
  CC_CHECK(h && name && data && shape && ndim >= 0 && ndim <= 4, "bad argument");
  CC_CHECK(!h->finalized, "cc_blaze_load after cc_blaze_finalize");
  HostTensor t; size_t n = 1;
  for (int i = 0; i < ndim; ++i) { t.shape.push_back(shape[i]); n *= (size_t)shape[i]; }
  t.data.assign(data, data + n);
  h->host[name] = std::move(t);
  CC_API_END
}

int cc_blaze_finalize(cc_blaze* h) {
  CC_API_BEGIN
  CC_CHECK(h && !h->finalized, "bad handle");
  CC_HIP(hipSetDevice(h->device));
  const int E = h->dtype == F32 ? 4 : 8;
  h->stem = make_conv(h, "conv_tiny", 1, E);
  for (int i = 0; i < 31; ++i) {
    const std::string p = "backbone_tiny.list." + std::to_string(i) + ".";
    BBlock b; b.cin = kBlocks[i][0]; b.cout = kBlocks[i][1]; b.stride = kBlocks[i][2];
    b.dw = make_conv(h, p + "conv0_tiny", b.cin);
    b.pw = make_conv(h, p + "conv1_tiny", 1);
    CC_CHECK(b.dw.cout == b.cin && b.pw.cin == b.cin && b.pw.cout == b.cout, p + ": conv shapes");
    h->blocks.push_back(b);
  }
  h->fdw = make_conv(h, "final.conv0_tiny", 96); h->fpw = make_conv(h, "final.conv1_tiny", 1);
  h->cls8 = make_conv(h, "classifier_8_tiny", 1); h->cls16 = make_conv(h, "classifier_16_tiny", 1);
  h->reg8 = make_conv(h, "regressor_8_tiny", 1); h->reg16 = make_conv(h, "regressor_16_tiny", 1);
  const HostTensor& a = need(h, "anchors");
  CC_CHECK(a.data.size() == (size_t)kAnchors * 4, "anchors must be (896,4)");
  h->anchors = upload_f32(h->wallocs, a.data);
  h->finalized = true;
  h->host.clear();
  CC_API_END
}

int cc_blaze_detect(cc_blaze* h, const void* img, int H, int W, int img_f32, int img_on_device, float* out, int out_on_device, void* stream) {
  CC_API_BEGIN
  CC_CHECK(h && img && out && H > 0 && W > 0, "bad argument");
  CC_CHECK(h->finalized, "cc_blaze_detect before cc_blaze_finalize");
  CC_HIP(hipSetDevice(h->device));
  BPlan* P = get_plan(h, H, W, img_f32 ? 1 : 0);
  hipStream_t s = h->stream;
  if (stream) {
    hipEvent_t e; CC_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    CC_HIP(hipEventRecord(e, (hipStream_t)stream)); CC_HIP(hipStreamWaitEvent(s, e, 0)); CC_HIP(hipEventDestroy(e));
  }
  CC_HIP(hipMemcpyAsync(P->in_dev, img, (size_t)H * W * 3 * (img_f32 ? 4 : 1), img_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s));
  CC_HIP(hipGraphLaunch(P->exec, s));
  const size_t ob = (size_t)kAnchors * 17 * 4;
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

void cc_blaze_destroy(cc_blaze* h) {
  if (!h) return;
  hipSetDevice(h->device);
  if (h->stream) hipStreamSynchronize(h->stream);
  h->plans.clear();
  for (void* p : h->wallocs) hipFree(p);
  pool_stream_put(h->device, h->stream);                  // parked, never destroyed (kernels.h)
  delete h;
}

}  // extern "C"

// CLIP runtime: OpenCLIP ViT image tower + text tower behind the C ABI.
//
// Stands behind OpenCLIP.precompute_embedding (models/objects.py:94-133) and encode_text
// (models/objects.py:145-186).  Per batch size one Plan (buffers + launch list captured into a hipGraph).
// Layout: tokens are rows; the residual stream stays f32 in HBM (B*L x D), every GEMM operand is the storage
// dtype with f32 accumulation; bias / tanh-GELU / residual-add are GEMM epilogues, attention is one fused kernel
// per layer, so a ViT-L/14 layer is 7 launches (ln, qkv, attn, out+res, ln, fc+gelu, proj+res).
#include <map>
#include <memory>
#include <string>
#include <vector>
#include <cstring>
#include "kernels.h"
#include "../../include/clearcam_hip.h"

using namespace cc;

namespace {

struct HostTensor { std::vector<float> data; std::vector<int64_t> shape; };

struct Lin { void* w = nullptr; float* b = nullptr; int n = 0, k = 0, kw = 0; };   // W[N][kw] storage dtype (rows zero padded to 64), bias f32
struct Norm { float* w = nullptr; float* b = nullptr; };
struct Block { Norm ln1, ln2; Lin qkv, out, fc, proj; };

struct COp {
  int kind;   // 0 gemm, 1 layernorm, 2 attention, 3 patchify, 4 assemble, 5 embed, 6 l2norm
  ConvP g; LnP ln; AttnP at; PatchP pa; AssembleP as; EmbedP em; NormP nm;
};

struct CPlan {
  int B = 0;
  std::vector<void*> allocs;
  std::vector<COp> ops;
  float* in_dev = nullptr; int* tok_dev = nullptr; int* eot_dev = nullptr; float* out_dev = nullptr;
  hipGraphExec_t exec = nullptr;
  // image plans: the patch gather is launched OUTSIDE the captured graph, so that it can read the caller's device buffer where it lies
  // (round 4 staged the 153 MB of a 255-image batch through in_dev first: ~130 copy kernels, 4.9 % of the trace)
  bool has_patchify = false; PatchP patchify{};
  ~CPlan() { if (exec) hipGraphExecDestroy(exec); for (void* p : allocs) hipFree(p); }
  template <class T> T* alloc(size_t n) { void* p = nullptr; CC_HIP(hipMalloc(&p, n * sizeof(T) + 256)); allocs.push_back(p); return (T*)p; }
};

}  // namespace

struct cc_clip {
  cc_clip_config cfg{};
  int dtype = BF16, device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  std::map<std::string, HostTensor> host;
  std::vector<void*> wallocs;
  bool finalized = false, has_image = false, has_text = false;
  // image tower
  Lin patch; int patch_kpad = 0; float *cls = nullptr, *pos = nullptr; Norm ln_pre, ln_post; Lin proj;
  std::vector<Block> vblocks;
  // text tower
  float *tok_emb = nullptr, *tpos = nullptr; Norm ln_final; Lin tproj;
  std::vector<Block> tblocks;
  PlanCache<int, CPlan> img_plans, txt_plans;             // key: batch size | slot << 24
  bool timed = false;
  // batches in flight (cc_clip_submit_image / cc_clip_wait), as cc_yolo's: slot i > 0 has its own stream and plans
  std::vector<hipStream_t> slot_stream;
  std::vector<hipEvent_t> slot_done;
  long long submitted = 0;
  hipStream_t stream_of_slot(int i) const { return i == 0 ? stream : slot_stream[i - 1]; }
};

namespace {

const HostTensor& need(cc_clip* h, const std::string& name) {
  auto it = h->host.find(name);
  CC_CHECK(it != h->host.end(), "missing parameter " + name);
  return it->second;
}

float* upload_f32(cc_clip* h, const HostTensor& t) {
  float* d = nullptr;
  CC_HIP(hipMalloc((void**)&d, t.data.size() * 4 + 256));
  CC_HIP(hipMemcpy(d, t.data.data(), t.data.size() * 4, hipMemcpyHostToDevice));
  h->wallocs.push_back(d);
  return d;
}

// W given as [N][K] (transpose=false) or [K][N] (transpose=true); K optionally zero-padded to kpad.
Lin make_lin(cc_clip* h, const HostTensor& w, const HostTensor* b, bool transpose, int kpad = 0) {
  Lin l;
  const size_t numel = w.data.size();
  const int d0 = (int)w.shape[0];
  const int rest = (int)(numel / d0);
  l.n = transpose ? rest : d0;
  const int k = transpose ? d0 : rest;
  l.k = kpad > k ? kpad : k;
  l.kw = (l.k + 63) / 64 * 64;
  std::vector<float> p((size_t)l.n * l.kw, 0.f);
  for (int n = 0; n < l.n; ++n)
    for (int kk = 0; kk < k; ++kk) p[(size_t)n * l.kw + kk] = transpose ? w.data[(size_t)kk * l.n + n] : w.data[(size_t)n * k + kk];
  std::vector<char> tmp(p.size() * dtype_size(h->dtype));
  convert_f32_to(h->dtype, p.data(), tmp.data(), p.size());
  CC_HIP(hipMalloc(&l.w, tmp.size() + 256));
  CC_HIP(hipMemcpy(l.w, tmp.data(), tmp.size(), hipMemcpyHostToDevice));
  h->wallocs.push_back(l.w);
  if (b) { CC_CHECK((int)b->data.size() == l.n, "bias size mismatch"); l.b = upload_f32(h, *b); }
  return l;
}

Norm make_norm(cc_clip* h, const std::string& p) { return Norm{upload_f32(h, need(h, p + ".weight")), upload_f32(h, need(h, p + ".bias"))}; }

Block make_block(cc_clip* h, const std::string& p, const char* out_w, const char* out_b) {
  Block b;
  b.ln1 = make_norm(h, p + "ln_1"); b.ln2 = make_norm(h, p + "ln_2");
  b.qkv = make_lin(h, need(h, p + "in_proj_weight"), &need(h, p + "in_proj_bias"), false);
  b.out = make_lin(h, need(h, p + out_w), &need(h, p + out_b), false);
  b.fc = make_lin(h, need(h, p + "mlp_c_fc.weight"), &need(h, p + "mlp_c_fc.bias"), false);
  b.proj = make_lin(h, need(h, p + "mlp_c_proj.weight"), &need(h, p + "mlp_c_proj.bias"), false);
  return b;
}

void gemm(CPlan* P, const void* A, int M, const Lin& l, void* out, int out_f32, int act, const void* res) {
  COp op{}; op.kind = 0;
  op.g = gemm_params(A, l.k, M, l.k, l.w, l.kw, l.b, l.n, out, l.n, out_f32, act, res, l.n, 1);
  P->ops.push_back(op);
}
void lnorm(CPlan* P, const float* in, long stride, const int* idx, const Norm& n, void* out, int out_f32, int rows, int D) {
  COp op{}; op.kind = 1; op.ln = LnP{in, stride, idx, n.w, n.b, out, out_f32, rows, D}; P->ops.push_back(op);
}

// residual blocks shared by both towers (objects.py:104-127, 151-180)
void add_blocks(cc_clip* h, CPlan* P, const std::vector<Block>& blocks, float* x, int B, int L, int D, int H, int mlp, int causal) {
  const int M = B * L; const size_t es = dtype_size(h->dtype);
  char* hbuf = P->alloc<char>((size_t)M * D * es);
  char* qkv = P->alloc<char>((size_t)M * 3 * D * es);
  char* ctx = P->alloc<char>((size_t)M * D * es);
  char* ff = P->alloc<char>((size_t)M * mlp * es);
  for (const Block& b : blocks) {
    lnorm(P, x, D, nullptr, b.ln1, hbuf, 0, M, D);
    gemm(P, hbuf, M, b.qkv, qkv, 0, 0, nullptr);
    COp at{}; at.kind = 2; at.at = AttnP{qkv, ctx, B, L, H, D, causal, 1.0f / 8.0f}; P->ops.push_back(at);
    gemm(P, ctx, M, b.out, x, 1, 0, x);                 // x = x + attn_out   (f32 residual stream, in place)
    lnorm(P, x, D, nullptr, b.ln2, hbuf, 0, M, D);
    gemm(P, hbuf, M, b.fc, ff, 0, 2, nullptr);          // tanh-GELU epilogue
    gemm(P, ff, M, b.proj, x, 1, 0, x);                 // x = x + mlp
  }
}

void run_ops(cc_clip* h, CPlan* P, hipStream_t s) {
  for (const COp& op : P->ops) {
    if (op.kind == 3 && P->has_patchify) continue;           // launched by the caller of the graph with the batch's own source pointer
    switch (op.kind) {
      case 0: launch_conv(h->dtype, op.g, s); break;
      case 1: launch_layernorm(h->dtype, op.ln, s); break;
      case 2: launch_attention(h->dtype, op.at, s); break;
      case 3: launch_patchify(h->dtype, op.pa, s); break;
      case 4: launch_assemble_ln(h->dtype, op.as, s); break;
      case 5: launch_embed(op.em, s); break;
      default: launch_l2norm(op.nm, s); break;
    }
  }
}

// the patch gather of an image plan, reading `x` (the caller's device buffer, or the plan's staging buffer after a host upload)
void run_patchify(cc_clip* h, CPlan* P, const float* x, hipStream_t s) {
  PatchP pa = P->patchify; pa.x = x;
  launch_patchify(h->dtype, pa, s);
}

void capture(cc_clip* h, CPlan* P) {
  if (P->has_patchify) run_patchify(h, P, P->in_dev, h->stream);
  run_ops(h, P, h->stream);                      // eager warm-up (sets kernel attributes, validates launches)
  CC_HIP(hipStreamSynchronize(h->stream));
  hipGraph_t graph = nullptr;
  CC_HIP(hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
  try { run_ops(h, P, h->stream); } catch (...) { hipStreamEndCapture(h->stream, &graph); if (graph) hipGraphDestroy(graph); throw; }
  CC_HIP(hipStreamEndCapture(h->stream, &graph));
  CC_HIP(hipGraphInstantiate(&P->exec, graph, nullptr, nullptr, 0));
  CC_HIP(hipGraphDestroy(graph));
}

CPlan* image_plan(cc_clip* h, int B, int slot = 0) {
  CC_CHECK(B < (1 << 24), "batch too large");
  const int key = B | slot << 24;
  if (CPlan* hit = h->img_plans.find(key)) return hit;
  const cc_clip_config& c = h->cfg;
  std::unique_ptr<CPlan> P(new CPlan()); P->B = B;
  const int g = c.image_size / c.patch, L = g * g + 1, D = c.v_width; const size_t es = dtype_size(h->dtype);
  P->in_dev = P->alloc<float>((size_t)B * 3 * c.image_size * c.image_size);
  P->out_dev = P->alloc<float>((size_t)B * c.embed);
  char* patches = P->alloc<char>((size_t)B * g * g * h->patch_kpad * es);
  char* pemb = P->alloc<char>((size_t)B * g * g * D * es);
  float* x = P->alloc<float>((size_t)B * L * D);
  char* pooled = P->alloc<char>((size_t)B * D * es);
  COp pa{}; pa.kind = 3; pa.pa = PatchP{P->in_dev, patches, B, c.image_size, c.patch, h->patch_kpad}; P->ops.push_back(pa);
  P->has_patchify = true; P->patchify = pa.pa;
  gemm(P.get(), patches, B * g * g, h->patch, pemb, 0, 0, nullptr);                       // visual_conv1 (no bias)
  COp as{}; as.kind = 4; as.as = AssembleP{pemb, h->cls, h->pos, h->ln_pre.w, h->ln_pre.b, x, B, L, D}; P->ops.push_back(as);
  add_blocks(h, P.get(), h->vblocks, x, B, L, D, c.v_heads, c.v_mlp, 0);
  lnorm(P.get(), x, (long)L * D, nullptr, h->ln_post, pooled, 0, B, D);                   // token 0 of every image
  gemm(P.get(), pooled, B, h->proj, P->out_dev, 1, 0, nullptr);
  COp nm{}; nm.kind = 6; nm.nm = NormP{P->out_dev, B, c.embed, 1e-8f}; P->ops.push_back(nm);
  capture(h, P.get());
  return h->img_plans.insert(key, std::move(P), h->stream, [&](CPlan*) { for (hipStream_t t : h->slot_stream) hipStreamSynchronize(t); });
}

CPlan* text_plan(cc_clip* h, int B) {
  if (CPlan* hit = h->txt_plans.find(B)) return hit;
  const cc_clip_config& c = h->cfg;
  std::unique_ptr<CPlan> P(new CPlan()); P->B = B;
  const int L = c.t_ctx, D = c.t_width; const size_t es = dtype_size(h->dtype);
  P->tok_dev = P->alloc<int>((size_t)B * L);
  P->eot_dev = P->alloc<int>(B);
  CC_HIP(hipMemset(P->tok_dev, 0, (size_t)B * L * 4)); CC_HIP(hipMemset(P->eot_dev, 0, B * 4));
  P->out_dev = P->alloc<float>((size_t)B * c.embed);
  float* x = P->alloc<float>((size_t)B * L * D);
  char* pooled = P->alloc<char>((size_t)B * D * es);
  COp em{}; em.kind = 5; em.em = EmbedP{P->tok_dev, h->tok_emb, h->tpos, x, B, L, D}; P->ops.push_back(em);
  add_blocks(h, P.get(), h->tblocks, x, B, L, D, c.t_heads, c.t_mlp, 1);
  lnorm(P.get(), x, D, P->eot_dev, h->ln_final, pooled, 0, B, D);                         // row b*L + argmax(tokens[b])
  gemm(P.get(), pooled, B, h->tproj, P->out_dev, 1, 0, nullptr);
  COp nm{}; nm.kind = 6; nm.nm = NormP{P->out_dev, B, c.embed, 0.f}; P->ops.push_back(nm);
  capture(h, P.get());
  return h->txt_plans.insert(B, std::move(P), h->stream);
}

}  // namespace

#define CC_API_BEGIN try {
#define CC_API_END                                                         \
  return 0; }                                                              \
  catch (const cc::Error& e) { cc::set_error(e.what()); return e.code; }   \
  catch (const std::exception& e) { cc::set_error(e.what()); return -1; }

extern "C" {

int cc_clip_create(cc_clip** h, const cc_clip_config* cfg, int dtype, int device) {
  CC_API_BEGIN
  CC_CHECK(h && cfg, "null argument");
  CC_CHECK(dtype >= 0 && dtype <= 2, "dtype must be 0 (f32), 1 (f16) or 2 (bf16)");
  CC_CHECK(cfg->v_width == cfg->v_heads * 64 && cfg->t_width == cfg->t_heads * 64, "head dim must be 64");
  CC_CHECK(cfg->image_size % cfg->patch == 0 && cfg->v_width <= 1024 && cfg->t_width <= 1024, "unsupported geometry");
  const int g = cfg->image_size / cfg->patch;
  CC_CHECK(g * g + 1 <= 288 && cfg->t_ctx <= 288, "at most 288 tokens per sequence");
  int n = 0; CC_HIP(hipGetDeviceCount(&n));
  CC_CHECK(n > 0 && device >= 0 && device < n, "no such HIP device");
  CC_HIP(hipSetDevice(device));
  std::unique_ptr<cc_clip> c(new cc_clip());
  c->cfg = *cfg; c->dtype = dtype; c->device = device;
  c->stream = pool_stream_get(device);
  CC_HIP(hipEventCreate(&c->ev0)); CC_HIP(hipEventCreate(&c->ev1));
  *h = c.release();
  CC_API_END
}

int cc_clip_load(cc_clip* h, const char* name, const float* data, const int64_t* shape, int ndim) {
  CC_API_BEGIN
  CC_CHECK(h && name && data && shape && ndim >= 0 && ndim <= 4, "bad argument");
  CC_CHECK(!h->finalized, "cc_clip_load after cc_clip_finalize");
  HostTensor t; size_t n = 1;
  for (int i = 0; i < ndim; ++i) { t.shape.push_back(shape[i]); n *= (size_t)shape[i]; }
  t.data.assign(data, data + n);
  h->host[name] = std::move(t);
  CC_API_END
}

int cc_clip_finalize(cc_clip* h) {
  CC_API_BEGIN
  CC_CHECK(h && !h->finalized, "bad handle");
  CC_HIP(hipSetDevice(h->device));
  const cc_clip_config& c = h->cfg;
  // a tower is built when its first parameter is present; a missing parameter inside a present tower is an error
  if (h->host.count("visual_conv1.weight")) {
    const int k = 3 * c.patch * c.patch;
    h->patch_kpad = (k + 7) / 8 * 8;
    h->patch = make_lin(h, need(h, "visual_conv1.weight"), nullptr, false, h->patch_kpad);
    CC_CHECK(h->patch.n == c.v_width, "visual_conv1.weight shape");
    h->cls = upload_f32(h, need(h, "class_embedding"));
    h->pos = upload_f32(h, need(h, "positional_embedding"));
    h->ln_pre = make_norm(h, "ln_pre"); h->ln_post = make_norm(h, "ln_post");
    h->proj = make_lin(h, need(h, "proj"), nullptr, true);
    for (int i = 0; i < c.v_layers; ++i) h->vblocks.push_back(make_block(h, "resblocks_img." + std::to_string(i) + ".", "out_proj_weight", "out_proj_bias"));
    h->has_image = true;
  }
  if (h->host.count("token_embedding.weight")) {
    h->tok_emb = upload_f32(h, need(h, "token_embedding.weight"));
    h->tpos = upload_f32(h, need(h, "positional_embedding_text"));
    h->ln_final = make_norm(h, "ln_final");
    h->tproj = make_lin(h, need(h, "text_projection"), nullptr, true);
    for (int i = 0; i < c.t_layers; ++i) h->tblocks.push_back(make_block(h, "resblocks." + std::to_string(i) + ".", "attn_out_proj_weight", "attn_out_proj_bias"));
    h->has_text = true;
  }
  CC_CHECK(h->has_image || h->has_text, "no tower parameters loaded");
  h->finalized = true;
  h->host.clear();
  CC_API_END
}

static void chain_in(cc_clip* h, void* stream) {
  if (!stream) return;
  hipEvent_t e; CC_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  CC_HIP(hipEventRecord(e, (hipStream_t)stream)); CC_HIP(hipStreamWaitEvent(h->stream, e, 0)); CC_HIP(hipEventDestroy(e));
}
static void finish(cc_clip* h, CPlan* P, float* out, int out_on_device, void* stream, size_t bytes) {
  if (out_on_device) {
    CC_HIP(hipMemcpyAsync(out, P->out_dev, bytes, hipMemcpyDeviceToDevice, h->stream));
    if (stream) {
      hipEvent_t e; CC_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      CC_HIP(hipEventRecord(e, h->stream)); CC_HIP(hipStreamWaitEvent((hipStream_t)stream, e, 0)); CC_HIP(hipEventDestroy(e));
    }
  } else {
    CC_HIP(hipMemcpyAsync(out, P->out_dev, bytes, hipMemcpyDeviceToHost, h->stream));
    CC_HIP(hipStreamSynchronize(h->stream));
  }
}

int cc_clip_encode_image(cc_clip* h, const float* x, int B, int x_on_device, float* out, int out_on_device, void* stream) {
  CC_API_BEGIN
  CC_CHECK(h && x && out && B > 0, "bad argument");
  CC_CHECK(h->finalized && h->has_image, "image tower not loaded");
  CC_HIP(hipSetDevice(h->device));
  CPlan* P = image_plan(h, B);
  chain_in(h, stream);
  const size_t nb = (size_t)B * 3 * h->cfg.image_size * h->cfg.image_size * 4;
  // device input: read in place
  static const bool stage_always = [] { const char* e = getenv("CLEARCAM_CLIP_STAGE_INPUT"); return e && atoi(e) != 0; }();   // A/B: round 4's copy into the plan's buffer
  const bool in_place = x_on_device && ((uintptr_t)x & 3) == 0 && !stage_always;
  if (!in_place) CC_HIP(hipMemcpyAsync(P->in_dev, x, nb, x_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, h->stream));
  CC_HIP(hipEventRecord(h->ev0, h->stream));
  run_patchify(h, P, in_place ? x : P->in_dev, h->stream);
  CC_HIP(hipGraphLaunch(P->exec, h->stream));
  CC_HIP(hipEventRecord(h->ev1, h->stream));
  h->timed = true;
  finish(h, P, out, out_on_device, stream, (size_t)B * h->cfg.embed * 4);
  CC_API_END
}

// Batches in flight for the image tower, the contract of cc_yolo_set_in_flight / cc_yolo_submit / cc_yolo_wait (include/clearcam_hip.h).
// What it buys depends on the batch: a 255-image batch fills the chip by itself (its GEMMs have > 2000 tiles: 48 ms with one, two or
// three in flight), small batches do not (the reference encodes one crop per call, models/objects.py:356-363).
int cc_clip_set_in_flight(cc_clip* h, int n) {
  CC_API_BEGIN
  CC_CHECK(h && n >= 1 && n <= 8, "in-flight depth must be 1..8");
  CC_HIP(hipSetDevice(h->device));
  CC_HIP(hipStreamSynchronize(h->stream));
  for (hipStream_t t : h->slot_stream) CC_HIP(hipStreamSynchronize(t));
  if (n == (int)h->slot_stream.size() + 1) return 0;   // same depth: slots and tickets stay valid
  while ((int)h->slot_stream.size() > n - 1) { pool_stream_put(h->device, h->slot_stream.back()); h->slot_stream.pop_back(); }
  grow_slot_streams(h->device, h->stream, h->slot_stream, n - 1);
  while ((int)h->slot_done.size() > n) { hipEventDestroy(h->slot_done.back()); h->slot_done.pop_back(); }
  while ((int)h->slot_done.size() < n) { hipEvent_t e; CC_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming)); h->slot_done.push_back(e); }
  h->submitted = 0;
  CC_API_END
}

int cc_clip_submit_image(cc_clip* h, const float* x, int B, int x_on_device, float* out, int out_on_device, void* stream, long long* ticket) {
  CC_API_BEGIN
  CC_CHECK(h && x && out && ticket && B > 0, "bad argument");
  CC_CHECK(h->finalized && h->has_image, "image tower not loaded");
  CC_HIP(hipSetDevice(h->device));
  if (h->slot_done.empty()) { hipEvent_t e; CC_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming)); h->slot_done.push_back(e); }
  const int depth = (int)h->slot_stream.size() + 1, slot = (int)(h->submitted % depth);
  CPlan* P = image_plan(h, B, slot);
  hipStream_t s = h->stream_of_slot(slot);
  if (stream) {   // x is ready on the caller's stream; that stream does not wait for the result (cc_clip_wait does)
    hipEvent_t e; CC_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    CC_HIP(hipEventRecord(e, (hipStream_t)stream)); CC_HIP(hipStreamWaitEvent(s, e, 0)); CC_HIP(hipEventDestroy(e));
  }
  const size_t nb = (size_t)B * 3 * h->cfg.image_size * h->cfg.image_size * 4;
  const bool in_place = x_on_device && ((uintptr_t)x & 3) == 0;    // `x` must then stay valid until the submission has completed
  if (!in_place) CC_HIP(hipMemcpyAsync(P->in_dev, x, nb, x_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s));
  run_patchify(h, P, in_place ? x : P->in_dev, s);
  CC_HIP(hipGraphLaunch(P->exec, s));
  CC_HIP(hipMemcpyAsync(out, P->out_dev, (size_t)B * h->cfg.embed * 4, out_on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, s));
  CC_HIP(hipEventRecord(h->slot_done[slot], s));
  *ticket = h->submitted++;
  CC_API_END
}

int cc_clip_wait(cc_clip* h, long long ticket, void* stream) {
  CC_API_BEGIN
  CC_CHECK(h, "null handle");
  CC_CHECK(ticket >= 0 && ticket < h->submitted, "no such submission");
  CC_HIP(hipSetDevice(h->device));
  hipEvent_t e = h->slot_done[(int)(ticket % ((long long)h->slot_stream.size() + 1))];   // a later submission of the slot finishes after this one
  if (stream) CC_HIP(hipStreamWaitEvent((hipStream_t)stream, e, 0));
  else CC_HIP(hipEventSynchronize(e));
  CC_API_END
}

int cc_clip_encode_text(cc_clip* h, const int32_t* tokens, int B, float* out, int out_on_device, void* stream) {
  CC_API_BEGIN
  CC_CHECK(h && tokens && out && B > 0, "bad argument");
  CC_CHECK(h->finalized && h->has_text, "text tower not loaded");
  CC_HIP(hipSetDevice(h->device));
  const int L = h->cfg.t_ctx;
  std::vector<int> eot(B);
  for (int b = 0; b < B; ++b) {
    int best = 0;
    for (int t = 0; t < L; ++t) {
      const int v = tokens[(size_t)b * L + t];
      CC_CHECK(v >= 0 && v < h->cfg.t_vocab, "token id out of range");
      if (v > tokens[(size_t)b * L + best]) best = t;          // first occurrence of the maximum = Tensor.argmax
    }
    eot[b] = b * L + best;
  }
  CPlan* P = text_plan(h, B);
  chain_in(h, stream);
  CC_HIP(hipMemcpyAsync(P->tok_dev, tokens, (size_t)B * L * 4, hipMemcpyHostToDevice, h->stream));
  CC_HIP(hipMemcpyAsync(P->eot_dev, eot.data(), (size_t)B * 4, hipMemcpyHostToDevice, h->stream));
  CC_HIP(hipStreamSynchronize(h->stream));       // `eot` is a stack buffer
  CC_HIP(hipEventRecord(h->ev0, h->stream));
  CC_HIP(hipGraphLaunch(P->exec, h->stream));
  CC_HIP(hipEventRecord(h->ev1, h->stream));
  h->timed = true;
  finish(h, P, out, out_on_device, stream, (size_t)B * h->cfg.embed * 4);
  CC_API_END
}

int cc_clip_last_gpu_ms(cc_clip* h, float* ms) {
  CC_API_BEGIN
  CC_CHECK(h && ms && h->timed, "no encode call yet");
  CC_HIP(hipEventSynchronize(h->ev1));
  CC_HIP(hipEventElapsedTime(ms, h->ev0, h->ev1));
  CC_API_END
}

void cc_clip_destroy(cc_clip* h) {
  if (!h) return;
  hipSetDevice(h->device);
  if (h->stream) hipStreamSynchronize(h->stream);
  for (hipStream_t t : h->slot_stream) hipStreamSynchronize(t);
  h->img_plans.clear(); h->txt_plans.clear();
  for (hipStream_t t : h->slot_stream) pool_stream_put(h->device, t);       // parked, never destroyed (kernels.h)
  for (hipEvent_t e : h->slot_done) hipEventDestroy(e);
  for (void* p : h->wallocs) hipFree(p);
  if (h->ev0) hipEventDestroy(h->ev0);
  if (h->ev1) hipEventDestroy(h->ev1);
  pool_stream_put(h->device, h->stream);
  delete h;
}

}  // extern "C"

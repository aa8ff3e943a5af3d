// Host-side helpers shared by the small conv nets of the face path (face.hip: AdaFace IR-50, blaze.hip: BlazeFace):
// parameter staging, OIHW -> [Cout][kh][kw][Cin] packing in the storage dtype, ConvP construction, hipGraph capture.
#pragma once
#include <functional>
#include <string>
#include <vector>
#include "kernels.h"

namespace cc {

struct HostTensor { std::vector<float> data; std::vector<int64_t> shape; };
struct PConv { void* w = nullptr; float* bias = nullptr; float* slope = nullptr; int cin = 0, cout = 0, k = 0, kw = 0; };

inline float* upload_f32(std::vector<void*>& owner, const std::vector<float>& v) {
  float* d = nullptr;
  CC_HIP(hipMalloc((void**)&d, v.size() * 4 + 256));
  CC_HIP(hipMemcpy(d, v.data(), v.size() * 4, hipMemcpyHostToDevice));
  owner.push_back(d);
  return d;
}

// OIHW weights (O, I/groups, k, k) -> dense [O][k][k][cin_pad] rows zero-padded to 64 elements, each output row scaled by
// row_scale[o] (BatchNorm folding; empty = 1).  groups > 1 (depthwise included) become block-diagonal dense weights: these
// layers are a few MFLOP, one kernel for everything beats a second code path.
inline PConv pack_conv(int dtype, std::vector<void*>& owner, const HostTensor& w, int groups, const std::vector<float>& row_scale,
                       const std::vector<float>& bias, int cin_pad = 0) {
  CC_CHECK(w.shape.size() == 4 && w.shape[2] == w.shape[3], "conv weight must be OIHW");
  const int co = (int)w.shape[0], cig = (int)w.shape[1], k = (int)w.shape[2], ci = cig * groups, cog = co / groups;
  const int cp = cin_pad > ci ? cin_pad : ci;
  const size_t kreal = (size_t)k * k * cp, kw = (kreal + 63) / 64 * 64;
  std::vector<float> packed((size_t)co * kw, 0.f);
  for (int n = 0; n < co; ++n) {
    const float s = row_scale.empty() ? 1.0f : row_scale[n];
    const int g = n / cog;
    for (int c = 0; c < cig; ++c)
      for (int r = 0; r < k; ++r)
        for (int q = 0; q < k; ++q)
          packed[(size_t)n * kw + (size_t)(r * k + q) * cp + g * cig + c] = w.data[(((size_t)n * cig + c) * k + r) * k + q] * s;
  }
  PConv pc; pc.cin = cp; pc.cout = co; pc.k = k; pc.kw = (int)kw;
  std::vector<char> tmp(packed.size() * dtype_size(dtype));
  convert_f32_to(dtype, packed.data(), tmp.data(), packed.size());
  CC_HIP(hipMalloc(&pc.w, tmp.size() + 256));
  CC_HIP(hipMemcpy(pc.w, tmp.data(), tmp.size(), hipMemcpyHostToDevice));
  owner.push_back(pc.w);
  CC_CHECK(bias.empty() || (int)bias.size() == co, "conv bias size");
  if (!bias.empty()) pc.bias = upload_f32(owner, bias);
  return pc;
}

// NHWC conv over one dense source.  pad < 0: k/2 on every side.  Ho/Wo <= 0: the usual (H + 2 pad - k) / stride + 1;
// larger values describe asymmetric padding (extra zero rows/columns at the bottom/right: the loaders read zeros there).
inline ConvP conv_params(const PConv& pc, const void* x, int B, int H, int W, int stride, void* out, int out_cstride, int out_f32, int act,
                         const void* res, int res_cstride, int pad = -1, int Ho = 0, int Wo = 0) {
  ConvP c{};
  c.s0 = Src{x, H, W, pc.cin, 0, pc.cin, 0};
  c.s1 = Src{x, 1, 1, 0, 0, 0, 0};
  c.B = B; c.Hin = H; c.Win = W; c.Cin = pc.cin;
  c.ks = pc.k; c.stride = stride; c.pad = pad < 0 ? pc.k / 2 : pad;
  c.Ho = Ho > 0 ? Ho : (H + 2 * c.pad - pc.k) / stride + 1; c.Wo = Wo > 0 ? Wo : (W + 2 * c.pad - pc.k) / stride + 1;
  c.Cout = pc.cout; c.Ktot = pc.k * pc.k * pc.cin; c.Kw = pc.kw; c.w = pc.w; c.bias = pc.bias;
  c.out = out; c.out_cstride = out_cstride; c.out_coff = 0; c.out_f32 = out_f32;
  c.res = res; c.res_cstride = res_cstride; c.res_coff = 0; c.res_f32 = 0;
  c.act = act; c.slope = pc.slope;
  return c;
}

// run once eagerly (kernel attributes, launch validation), then capture the same launch list into an executable graph
inline hipGraphExec_t capture_graph(hipStream_t s, const std::function<void()>& run) {
  run();
  CC_HIP(hipStreamSynchronize(s));
  hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr;
  CC_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  try { run(); } catch (...) { hipStreamEndCapture(s, &graph); if (graph) hipGraphDestroy(graph); throw; }
  CC_HIP(hipStreamEndCapture(s, &graph));
  CC_HIP(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
  CC_HIP(hipGraphDestroy(graph));
  return exec;
}

}  // namespace cc

// Kernel launchers (one translation unit per family).  All launches are asynchronous on `stream`.
#pragma once
#include "common.h"

namespace cc {

// ---- convolution / GEMM (conv_mfma.hip, conv_direct.hip) --------------------------------------
// True when the MFMA implicit-GEMM kernel can run this problem (16-byte channel alignment etc.).
bool conv_mfma_supported(int dt, const ConvP& p);
void launch_conv_mfma(int dt, const ConvP& p, hipStream_t stream);
// Generic direct convolution: any channel counts; slow path for shapes the MFMA kernel rejects.
void launch_conv_direct(int dt, const ConvP& p, hipStream_t stream);
// Picks the MFMA kernel when supported, else the direct kernel.
void launch_conv(int dt, const ConvP& p, hipStream_t stream);

// ---- pooling (pool.hip) ------------------------------------------------------------------------
struct PoolP {
  const void* in; int in_cstride, in_coff;
  void* out; int out_cstride, out_coff;
  int B, H, W, C;        // input dims / channels processed
  int Ho, Wo;
  int k, stride, pad;
  int mode;              // 0 = avg (count_include_pad, pad must be 0), 1 = max (-inf padding)
};
void launch_pool(int dt, const PoolP& p, hipStream_t stream);

// ---- detector pre/post (detect.hip) ------------------------------------------------------------
struct PreP {                 // letterbox: detection/yolov9.py:376-379,390-404
  const void* frames; int frame_f32;      // (B,H,W,3) BGR u8 or f32
  int B, H, W;                            // source dims
  int nh, nw, pad_y, pad_x, Hn, Wn;       // resized dims, padding, network dims
  const int* xlo; const int* xhi; const float* xfr;   // per-axis interpolation tables (device)
  const int* ylo; const int* yhi; const float* yfr;
  void* out; int out_c;                   // (B,Hn,Wn,out_c) storage dtype, RGB in ch 0..2, rest 0
};
void launch_preprocess(int dt, const PreP& p, hipStream_t stream);

struct DecodeP {              // DDetect decode + class max: detection/yolov9.py:209-220,440-448
  const float* raw[3]; int H[3], W[3];    // per level (B,H,W,144) f32
  int B, A;                               // A = sum H*W
  const float* dfl_w;                     // 16 weights
  float conf;                             // 0.25
  float* det;                             // (B,A,6) x1,y1,x2,y2,score(thresholded),cls
};
void launch_decode(const DecodeP& p, hipStream_t stream);

struct NmsP {                 // top-300 + mask NMS + scale_boxes: detection/yolov9.py:406-458
  const float* det; int B, A;
  float iou_thr;
  float pad_x, pad_y, gain; float src_w, src_h;
  float* out;                             // (B,300,6)
};
void launch_topk_nms(const NmsP& p, hipStream_t stream);

}  // namespace cc

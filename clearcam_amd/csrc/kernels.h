// Kernel launchers (one translation unit per family).  All launches are asynchronous on `stream`.
#pragma once
#include <vector>
#include "common.h"

namespace cc {

// Per-launch note for the profiler's table (cc_yolo_profile -> CLEARCAM_PROFILE_CSV): which kernel a conv took, how many tiles of work it
// had and how many tiles the chip takes per round (resident blocks x CUs; the grid for persistent kernels).  Filled only while
// g_note_launches is set - the occupancy query is not free.
struct LaunchNote { const char* kernel; long tiles; long slots; };
extern thread_local LaunchNote g_launch_note;   // per thread: cc_yolo_profile sets the switch and reads the note on the thread that launches
extern thread_local bool g_note_launches;
template <class K> inline void note_launch(const char* name, K kernel, long tiles, int threads, size_t lds, long persistent_grid = 0) {
  if (!g_note_launches) return;
  long slots = persistent_grid;
  if (!slots) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(kernel), threads, lds) != hipSuccess) nb = 1;
    static PerDevice pd;                                   // CU count cached per device (hipGetDeviceProperties is not cheap)
    slots = (long)nb * pd.cu_count(pd.index());
  }
  g_launch_note = LaunchNote{name, tiles, slots > 0 ? slots : 1};
}

// ---- convolution / GEMM (conv_mfma.hip, conv_direct.hip) --------------------------------------
// True when the MFMA implicit-GEMM kernel can run this problem (16-byte channel alignment etc.).
bool conv_mfma_supported(int dt, const ConvP& p);
void launch_conv_mfma(int dt, const ConvP& p, hipStream_t stream);
// Generic direct convolution: any channel counts; slow path for shapes the MFMA kernel rejects.
void launch_conv_direct(int dt, const ConvP& p, hipStream_t stream);
// Picks the MFMA kernel when supported, else the direct kernel.
void launch_conv(int dt, const ConvP& p, hipStream_t stream);
// 3x3 stride-2 conv over the 2x2 stride-1 average of its source (ConvP::s0.shift == -1; Hin x Win = the averaged map): conv_adown.hip
bool conv_adown_supported(int dt, const ConvP& p);
void launch_conv_adown(int dt, const ConvP& p, hipStream_t stream);

// ---- fused RepNCSP block (csp_fused.hip) ----------------------------------------------------------
// detection/yolov9.py:92-105 with n = 1: cv1 | cv2 (1x1), RepConvN 3x3, 3x3 + shortcut, cv3 (1x1) in one launch; 16-bit
// storage, hidden width 32 or 64.  Weights are the packed [Cout][Kw] matrices of the four convs it replaces.
struct CspP {
  const void* x; int x_cstride, x_coff;          // input view: 2*hid channels of a (B,H,W,x_cstride) tensor
  void* out; int out_cstride, out_coff;          // output view: 2*hid channels
  const void* w12; int kw12; const float* b12;   // cv1 | cv2: [2*hid][kw12]
  const void* wr; int kwr; const float* br;      // m.0.cv1 (RepConvN): [hid][kwr], k = tap*hid + c
  const void* wb; int kwb; const float* bb;      // m.0.cv2: [hid][kwb]
  const void* w3; int kw3; const float* b3;      // cv3: [2*hid][kw3], k = [m-branch | cv2]
  int B, H, W, hid;
  int split; float os12, osr, osb, os3;          // split weights (ConvP::split): 0 none, 1 all four convs, 2 the 1x1 convs (cv1 | cv2, cv3) only; rows [tap: hi | lo], the four exact 2^-e output scales
  int tx, tiles; float inv_tiles, inv_tx;        // filled by the launcher: 8x16 tiles per row / per image
  int dbg;                                       // development: 1..3 = stop after that stage and write its intermediate to `out`
  int stream;                                    // 1: take the weight-streaming one-tile-per-block variant even where the weights fit in LDS
};
bool csp_fused_supported(int dt, int hid, int split = 0);
void launch_csp_fused(int dt, const CspP& p, hipStream_t stream);

// ---- calibration-aware rounding of the 1x1 convs (calibrate.hip, C-ABI dtype "f16c") -----------------------------------------------
// S pixel rows (indices rows_dev) of a 1x1 stride-1 conv's input view over f32 activations -> dense (S, Cin) f32
void launch_sample_rows(const ConvP& p, const int* rows_dev, int S, float* out_dev, hipStream_t stream);
// H (ci x ci) = X^T X / rows in double
void second_moments(const float* X, int rows, int ci, std::vector<double>& H);
// GPTQ column walk: w (co x ci) f32 -> f16-representable f32 values; 0 ok, < 0: H not positive definite (leave the conv to controlled rounding)
int gptq_round_f16(const float* w, int co, int ci, const double* H, double damp, float* out);

// ---- pooling (pool.hip) ------------------------------------------------------------------------
struct PoolP {
  const void* in; int in_cstride, in_coff;
  void* out; int out_cstride, out_coff;
  int B, H, W, C;        // input dims / channels processed
  int Ho, Wo;
  int k, stride, pad;
  int mode;              // 0 = avg (count_include_pad, pad must be 0), 1 = max (-inf padding),
                         // 2 = avg 2x2 s1 p0 followed by max k s p (k,stride,pad describe the max stage; H,W are the raw input)
};
void launch_pool(int dt, const PoolP& p, hipStream_t stream);

// CBFuse (detection/yolov9.py:230-245): out = sum_k nearest_upsample(in_k, 2^shift_k) (+ the last, unscaled input)
struct FuseP {
  int n;                                  // inputs (<= 6)
  const void* in[6]; int H[6], W[6], cstride[6], coff[6], shift[6];
  void* out; int out_cstride, out_coff;
  int B, Ho, Wo, C;
};
void launch_fuse(int dt, const FuseP& p, hipStream_t stream);

// ---- detector pre/post (detect.hip) ------------------------------------------------------------
struct PreP {                 // letterbox: detection/yolov9.py:376-379,390-404
  const void* frames; int frame_f32;      // (B,H,W,3) BGR u8 or f32
  int B, H, W;                            // source dims
  int nh, nw, pad_y, pad_x, Hn, Wn;       // resized dims, padding, network dims
  const int* xlo; const int* xhi; const float* xfr;   // per-axis interpolation tables (device)
  const int* ylo; const int* yhi; const float* yfr;
  void* out; int out_c;                   // (B,Hn,Wn,out_c) storage dtype, colour in ch 0..2, rest 0
  int flip;                               // 1: BGR -> RGB (detector), 0: keep the channel order (BlazeFace)
  float div, sub, pad_val;                // inside the image: v / div - sub; in the padding: pad_val
};
void launch_preprocess(int dt, const PreP& p, hipStream_t stream);

// Letterbox fused into the detector's first conv (3x3 stride 2, 3 -> Cout, bias + SiLU): the network-input tensor
// (B,Hn,Wn,8) is never written or read; 16-bit storage types only (detect.hip).
struct StemP {
  PreP pre;                               // pre.out / pre.out_c unused
  const void* w; const float* bias;       // [Cout][32] storage dtype from stem_pack_weights (k = r*9 + s*3 + c, zero for k >= 27), bias f32
  const void* w_lo; float oscale;         // split weights (ConvP::split): the low plane in the same layout and the exact 2^-e output scale; null / ignored otherwise
  int Cout;                               // 16, 32 or 64
  void* out; int out_cstride, out_coff;   // (B,Ho,Wo,out_cstride) storage dtype
  int Ho, Wo;                             // Hn/2, Wn/2
  int abl;                                // timing ablation bits (development only; 0 in production)
};
bool stem_fused_supported(int dt, int Cout);
// reorder the generic conv's packed weights [Cout][w_row] (k = (r*3+s)*tap_stride + plane_off + c; plain: tap_stride = cin_pad, plane_off = 0;
// split: tap_stride = 2*cin_pad, plane_off = 0 / cin_pad for the high / low plane) into the fused kernel's [Cout][32] rows
void stem_pack_weights(int dt, const void* w_packed, int w_row, int tap_stride, int plane_off, int Cout, void* out, hipStream_t stream);
void launch_stem_fused(int dt, const StemP& p, hipStream_t stream);
// tinygrad `interpolate(mode='linear', align_corners=False)` index tables for one axis, evaluated in float32 (yolo.hip)
void axis_tables(int n_in, int n_out, std::vector<int>& lo, std::vector<int>& hi, std::vector<float>& fr);

struct DecodeP {              // DDetect decode + class max: detection/yolov9.py:209-220,440-448
  const float* raw[3]; int H[3], W[3];    // per level (B,H,W,144) f32
  int B, A;                               // A = sum H*W
  const float* dfl_w;                     // 16 weights
  float conf;                             // 0.25
  float* det;                             // (B,A,6) x1,y1,x2,y2,score(thresholded),cls
  int* nonfinite;                         // device counter: anchors whose logits were not finite (f16 activations past 65504 -> inf -> NaN); may be null
};
void launch_decode(const DecodeP& p, hipStream_t stream);

// DDetect's last 1x1 convs of both branches + decode in one launch (detection/yolov9.py:202-220,247-282): per level the box branch
// cv2[l][2] (1x1, 64 -> 64, four groups densified) and the class branch cv3[l][2] (1x1, ch -> 80) over their 16-bit inputs, then DFL
// softmax-expectation, dist2bbox, sigmoid, class max / argmax, threshold -> (B, A, 6) rows.  The 144 f32 logits per anchor stay on chip.
// Same MFMA accumulation order as the conv kernels and decode_kernel's own arithmetic: bit-identical rows.
struct HeadTailP {
  const void* bx[3]; const void* cl[3];   // (B,H,W,64) and (B,H,W,ch) storage dtype, dense
  const void* w2[3]; const void* w3[3];   // packed [64][kw2] / [80][kw3]
  const float* b2[3]; const float* b3[3];
  int H[3], W[3];
  int kw2, kw3, ch;
  int split; float os2[3], os3[3];        // split weights: rows are [hi(Cin) | lo(Cin)], os2 / os3 the exact 2^-e output scales per level
  int B, A;
  const float* dfl_w; float conf;
  float* det;                             // (B,A,6)
  int* nonfinite;                         // as DecodeP::nonfinite
};
bool head_tail_supported(int dt, int ch, int split = 0);
void launch_head_tail(int dt, const HeadTailP& p, hipStream_t stream);
// Streams of handles come from a process-wide pool per device and go back to it when a handle dies - they are NEVER destroyed.
// Destroying streams makes the runtime re-deal its few hardware queues, and a handle created afterwards has been seen replaying
// its captured graph 3x slower for its whole life ("Uploads"); parked streams keep the queue assignment stable.
hipStream_t pool_stream_get(int device);                 // a non-blocking stream of `device` (the device must be current)
void pool_stream_put(int device, hipStream_t s);         // idle it and park it for the next handle
// batches in flight (yolo.hip): streams for the extra slots of a handle, probed until kernels on them overlap with `base` and each other
bool streams_overlap(hipStream_t a, hipStream_t b);
void grow_slot_streams(int device, hipStream_t base, std::vector<hipStream_t>& slots, int n_extra);

struct NmsP {                 // top-300 + mask NMS + scale_boxes: detection/yolov9.py:406-458
  const float* det; int B, A;
  float iou_thr;
  float pad_x, pad_y, gain; float src_w, src_h;
  float* out;                             // (B,300,6)
};
void launch_topk_nms(const NmsP& p, hipStream_t stream);

// ---- CLIP towers (clip_kernels.hip) -------------------------------------------------------------
// Plain GEMM through the conv kernel:  out[M][N] = act(A[M][K] W[N][K]^T + bias) (+res).
ConvP gemm_params(const void* A, int lda, int M, int K, const void* W, int ldw, const float* bias, int N, void* out, int ldc,
                  int out_f32, int act, const void* res, int ldres, int res_f32);

struct LnP {                  // LayerNorm(eps 1e-5, biased var, affine): models/objects.py:105,123,129,153,176,182
  const float* in; long in_row_stride; const int* row_index;   // row r reads in + (row_index ? row_index[r] : r) * in_row_stride
  const float* w; const float* b;
  void* out; int out_f32;     // [rows][D]
  int rows, D;
};
void launch_layernorm(int dt, const LnP& p, hipStream_t stream);

struct PatchP { const float* x; void* out; int B, S, patch, Kpad; };   // (B,3,S,S) f32 -> (B*g*g, Kpad), k = c*p*p + kh*p + kw
void launch_patchify(int dt, const PatchP& p, hipStream_t stream);

struct AssembleP {            // cat(class_embedding, patches) + positional_embedding -> ln_pre   (objects.py:98-102)
  const void* patches;        // (B*(L-1), D) storage dtype
  const float* cls; const float* pos; const float* w; const float* b;
  float* out;                 // (B*L, D) f32 residual stream
  int B, L, D;
};
void launch_assemble_ln(int dt, const AssembleP& p, hipStream_t stream);

struct EmbedP { const int* tokens; const float* table; const float* pos; float* out; int B, L, D; };   // objects.py:148-149
void launch_embed(const EmbedP& p, hipStream_t stream);

struct AttnP {                // softmax(q k^T / sqrt(dh)) v per head, dh = 64   (objects.py:110-119,160-171)
  const void* qkv;            // (B*L, 3*D) storage dtype: [q | k | v]
  void* ctx;                  // (B*L, D)
  int B, L, H, D; int causal; float scale;
  int abl;                    // development (cc_attn_bench): 1 = stage K / V only, 2 = no staging (tiles over whatever LDS holds), 32 = empty blocks,
                              // 64 = print the occupancy, 128 = one query tile per wave round instead of two; 0 in production
};
void launch_attention(int dt, const AttnP& p, hipStream_t stream);

struct NormP { float* x; int rows, D; float eps; };   // x / (||x||_2 + eps)   (objects.py:132,186)
void launch_l2norm(const NormP& p, hipStream_t stream);

}  // namespace cc

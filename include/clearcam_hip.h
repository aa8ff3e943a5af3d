/* libclearcam_hip — C ABI of the MI355X-native detect / CLIP-encode / search path for clearcam.
 *
 * The reference (roryclear/clearcam) has no FFI for this path: its boundary is three duck-typed Python
 * call surfaces that sit directly on tinygrad.  Each entry point below names the reference interface
 * it stands behind; the Python shim in clearcam_amd/ keeps those surfaces byte-compatible and
 * INTEGRATION.md shows the ctypes binding a clearcam maintainer would add.
 *
 * Conventions: every function returns 0 on success or a negative errno-style code, with a thread-local
 * message available from cc_last_error().  The caller owns every buffer it passes.  A handle belongs to
 * one GPU and one submitting thread at a time (the reference runs all tensor work on its main thread,
 * clearcam.py:1214-1226).  `stream` is a hipStream_t (NULL = the handle's own stream); when results
 * go to host memory the call returns after they have landed, otherwise it only enqueues work.
 */
#ifndef CLEARCAM_HIP_H
#define CLEARCAM_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define CC_DTYPE_F32 0   /* parity mode: f32 storage, exact-f32 MFMA (157 TFLOP/s class)                                             */
#define CC_DTYPE_F16 1   /* speed mode:  f16 storage (activations AND weights rounded to 11 bits), f32 accumulate                    */
#define CC_DTYPE_BF16 2  /* speed mode:  bf16 storage (8 bits), f32 accumulate                                                       */
#define CC_DTYPE_F16S 3  /* tolerance mode at MFMA rate (detector only): f16 activations, every conv weight carried as */
                         /* TWO f16 planes W = W_hi + W_lo (~22 bits) multiplied into the same f32 accumulator - detections stay      */
                         /* within the reference tolerance for any float32 checkpoint at twice the MFMA issue of CC_DTYPE_F16        */
#define CC_DTYPE_F16H 4  /* CC_DTYPE_F16S where it is needed: two planes for the 1x1 convs of the detector's backbone (blocks up to the     */
                         /* SPPELAN) and the stem conv, one controlled-rounded f16 plane for every other conv (controlled rounding balances  */
                         /* a 3x3 filter's taps; a 1x1 has nothing to balance).  Measured as close to f32 as CC_DTYPE_F16S on un-rounded      */
                         /* checkpoints at 1.15x instead of 2x the MFMA issue of CC_DTYPE_F16 (bench default)                              */

#define CC_DTYPE_F16C 5  /* calibrated mode (detector only): one f16 plane per conv (two in the stem conv), the 1x1 convs' weights rounded with    */
                         /* the errors steered by their inputs' second moments on calibration frames (cc_yolo_calibrate): CC_DTYPE_F16's rate     */

#define CC_MAX_DET 300   /* rows per frame of the detector output (detection/yolov9.py:439) */

const char* cc_last_error(void);
int cc_version(void);
int cc_device_count(int* n);

/* ---------------------------------------------------------------------------------------------
 * Detector — stands behind `YOLOv9(size, res)` and `YOLOv9.__call__(frame)`
 * (detection/yolov9.py:298-388), as called from clearcam.py:583 and test/run_mot.py:34.
 * ------------------------------------------------------------------------------------------- */
typedef struct cc_yolo cc_yolo;

/* YOLOv9.__init__ (yolov9.py:298-371): size in {"t","s","m","c","e"}, res = letterbox target. */
int cc_yolo_create(cc_yolo** h, const char* size, int res, int dtype, int device);
/* load_state_dict (yolov9.py:372-373): one call per state-dict entry, reference key names
 * (SURVEY.md Appendix C), host float32 data, OIHW weights / (Cout,) biases. */
int cc_yolo_load(cc_yolo* h, const char* name, const float* data, const int64_t* shape, int ndim);
/* Packs the loaded tensors into the device layout ([Cout][kh][kw][Cin], storage dtype). Fails if
 * any parameter of the graph is missing. */
int cc_yolo_finalize(cc_yolo* h);
/* dtype 5 ("f16c"): f16 activations and ONE f16 weight plane per conv (two in the stem conv), the 1x1 convs' float32 weights rounded to
 * f16 by a calibration-aware recursion (GPTQ: rounding errors steered by the second moments of each conv's own input) that cc_yolo_finalize
 * runs on an internal float32 pass over calibration frames: plain f16's frame rate, "f16h"'s tolerance on inputs like the calibration
 * frames (INTEGRATION.md says what was measured with mismatched ones).  cc_yolo_calibrate hands over the frames ((B,H,W,3) BGR uint8 or
 * float32 HOST memory, copied) between cc_yolo_load and cc_yolo_finalize; without it four frames of seeded white noise are used.
 * cc_yolo_calibration_info: packed 1x1 convs the recursion rounded / left to the plain controlled rounding (singular second moments). */
int cc_yolo_calibrate(cc_yolo* h, const void* frames, int B, int H, int W, int frame_f32);
int cc_yolo_calibration_info(cc_yolo* h, int* n_calibrated, int* n_fallback);
/* Host-only helper (no GPU needed; tests): the recursion itself - w (cout,cin) float32, H (cin,cin) float64 second moments of the input,
 * damp (ridge on H as a share of its mean diagonal; 0.03 in the library) -> out (cout,cin) float32 values exactly representable in f16. */
int cc_gptq_round_f16(const float* w, int64_t cout, int64_t cin, const double* H, double damp, float* out);
/* YOLOv9.__call__ (yolov9.py:375-388) over a batch: frames (B,H,W,3) BGR, uint8 (frame_f32=0, the
 * production path clearcam.py:582) or float32 (frame_f32=1, the MOT path run_mot.py:33);
 * out (B,300,6) float32 [x1,y1,x2,y2,conf,cls] in source-frame pixels, suppressed rows zero.
 * B=1 is exactly the reference call.  frames_on_device/out_on_device select host or device pointers. */
int cc_yolo_detect(cc_yolo* h, const void* frames, int B, int H, int W, int frame_f32,
                   int frames_on_device, float* out, int out_on_device, void* stream);
/* Range guard of the 16-bit modes: f16 activations saturate at 65504, and a checkpoint that drives them past it yields inf -> NaN logits,
 * which every threshold silently turns into "no detection".  The decode stage counts anchors whose logits were not finite: a
 * cc_yolo_detect call that returns its rows to HOST memory fails with -34 (ERANGE) and a message naming the remedy (dtype bf16 / f32) when
 * the count is non-zero; for device-output and cc_yolo_submit calls cc_yolo_nonfinite waits for the handle's streams and returns (and
 * clears) the count accumulated since the last query. */
int cc_yolo_nonfinite(cc_yolo* h, int* count);
/* Batches in flight - throughput mode for a GPU that serves many cameras (no counterpart in the reference, whose loop is one
 * synchronous batch-1 call per frame, clearcam.py:583).  cc_yolo_set_in_flight(h, n) gives the handle n slots (1..8, default 1), each
 * with its own stream, tensor arena and captured graph (the streams are probed until kernels on them really run side by side);
 * cc_yolo_submit runs one batch on the next slot (round robin), ordered after `stream` (where the caller produced device frames;
 * may be NULL) but WITHOUT making `stream` wait for the result, so the next submission's first layers - and its upload - overlap
 * this one's last; cc_yolo_wait(h, ticket, stream) makes `stream` (NULL: the calling thread) wait until that submission's
 * (B,300,6) rows are in `out`.  frames / out may be host pointers (frames_on_device / out_on_device = 0; pinned memory, or the copies
 * are not asynchronous): upload, detect and download then form one in-order chain on the slot's stream.  Results are bit-identical
 * to cc_yolo_detect.  `frames` must stay valid until the submission has completed; `out` buffers of submissions in flight must be
 * distinct.  Changing the depth drops the cached plans. */
int cc_yolo_set_in_flight(cc_yolo* h, int n);
int cc_yolo_submit(cc_yolo* h, const void* frames, int B, int H, int W, int frame_f32, int frames_on_device, float* out,
                   int out_on_device, void* stream, long long* ticket);
int cc_yolo_wait(cc_yolo* h, long long ticket, void* stream);
/* Parity taps: copy a named intermediate of the LAST detect call to host float32.
 * names: "input" (B,Hn,Wn,3) | "p3","p4","p5" (B,H,W,C) | "raw0","raw1","raw2" (B,H,W,144) |
 * "decoded" (B,A,6).  shape receives up to 4 dims; pass out=NULL to query the shape only.
 * In the 16-bit modes the letterbox is fused into the first conv and "input" is rebuilt on demand, and cc_yolo_profile
 * re-runs that kernel: for both, device frames handed to the last cc_yolo_detect call must still be alive (host frames are
 * staged in a buffer the handle owns). */
int cc_yolo_get_tensor(cc_yolo* h, const char* name, float* out, int64_t* shape, int* ndim);
/* GPU milliseconds of the last detect call's kernels (hipEvents on the launch stream). */
int cc_yolo_last_gpu_ms(cc_yolo* h, float* ms);
/* Eager replay of the last plan's launch list, `iters` times, with a hipEvent pair around every launch
 * on the launch stream.  ms[5] = average per-step milliseconds of {conv/GEMM, pooling, decode, top-k+NMS, fused letterbox +
 * first conv (0 when that fusion is off: f32 mode or CLEARCAM_FUSE_STEM=0; its time is then in the conv slot and in the
 * separate letterbox launch)}; alg_macs_per_step / n_conv_launches cover the launches timed in the conv slot. */
int cc_yolo_profile(cc_yolo* h, int iters, float* ms, double* alg_macs_per_step, int* n_conv_launches);
/* A subset of the last plan's launch list captured into a hipGraph of its own and replayed `iters` times between ONE hipEvent
 * pair on the launch stream (no per-launch event records): average milliseconds per replay.
 * which = 0: the conv / GEMM launches (the ones cc_yolo_profile counts in its conv slot), 1: every other launch (pooling, decode,
 * top-k + NMS, CBFuse, and the fused letterbox + first conv), 2: the whole step. */
int cc_yolo_profile_graph(cc_yolo* h, int iters, int which, float* ms_per_replay);
void cc_yolo_destroy(cc_yolo* h);

/* Single-layer entry used by the parity tests: NHWC conv + bias + optional SiLU on device buffers.
 * x (B,H,W,Cin) and out (B,Ho,Wo,Cout) in storage dtype `dtype`; w OIHW float32 host, bias host.
 * groups>1 is densified to block-diagonal weights exactly as the detector does for the head convs.
 * force_direct selects the kernel: 0 = what the detector would pick, 1 = direct (non-MFMA) fallback, 2 = generic MFMA
 * implicit GEMM, 3 = halo-resident 3x3, 4 = weights-stationary 3x3, ... 10 = weights-resident streaming 1x1 (ConvP::variant in
 * csrc/common.h lists them all; the specialised ones fail if the shape is not eligible). */
int cc_conv2d_nhwc(int dtype, const void* x_dev, int B, int H, int W, int Cin, const float* w_oihw,
                   const float* bias, int Cout, int k, int stride, int groups, int act, void* out_dev,
                   int force_direct, void* stream);
/* Diagnostic (kernel tuning, tools/dev/stream_ab.py, variant_sweep.py): average device milliseconds of one launch of the conv above on random
 * 16-bit data resident in HBM, weights packed once, `iters` launches between two events.  Not part of the drop-in surface. */
int cc_conv_bench(int dtype, int B, int H, int W, int Cin, int Cout, int k, int stride, int act, int variant, int iters, float* ms);
/* Diagnostic (kernel tuning): average device milliseconds of one attention launch (B images, L tokens, H heads of 64)
 * on random 16-bit data; abl: 0 the kernel, 1 K / V staging only, 2 tiles without staging.  Not part of the drop-in surface. */
int cc_attn_bench(int dtype, int B, int L, int H, int causal, int abl, int iters, float* ms);
/* Host-only helper (no GPU needed; tests / tooling): the CONTROLLED rounding cc_yolo_finalize applies to a conv's OIHW float32 weights
 * (cout, cin, k, k) for the plain 16-bit storage types (dtype 1 / 2): every weight goes to one of its two neighbours in the storage
 * type, chosen so that each output channel's total, the tap-sums of each input channel's filter, the channel-sums at each tap and the
 * first moments over the taps stay within about an ulp of their float32 values.  out[i] = the value the storage type holds, as float32. */
int cc_round_weights(int dtype, const float* w, int64_t cout, int64_t cin, int64_t k, float* out);
/* Diagnostic (kernel tuning): set a process-wide tuning switch at run time so that one process can A/B kernel variants.
 * key "phase_flags": the CLEARCAM_PHASE_FLAGS bit set of the eight-wave kernel (-1 = back to the environment / default);
 * key "stream": the weights-resident streaming 1x1 kernel off (0) / on (1) / as CLEARCAM_STREAM says (-1);
 * key "tile64_w": tile geometry of the 3x3 64 -> 64 tile kernel, 32 (8 x 32 pixels) / 16 (16 x 16) / -1 (its own rule);
 * keys "stream_abl", "stream_flags", "tile64_abl": timing ablations / flag bits of development builds (no effect otherwise).
 * Plans already built keep the launches they were built with.  Not part of the drop-in surface. */
int cc_dev_set(const char* key, int value);

/* ---------------------------------------------------------------------------------------------
 * CLIP — stands behind `OpenCLIP.precompute_embedding(x)` (models/objects.py:94-133) and
 * `encode_text(model, tokens)` (models/objects.py:145-186); tokenisation stays on the host
 * (utils/clip_tokenizer.py).
 * ------------------------------------------------------------------------------------------- */
typedef struct cc_clip cc_clip;
typedef struct {
  int image_size, patch, v_width, v_layers, v_heads, v_mlp;
  int t_ctx, t_vocab, t_width, t_layers, t_heads, t_mlp, embed;
} cc_clip_config;   /* ViT-L/14: {224,14,1024,24,16,4096, 77,49408,768,12,12,3072, 768} */

int cc_clip_create(cc_clip** h, const cc_clip_config* cfg, int dtype, int device);
int cc_clip_load(cc_clip* h, const char* name, const float* data, const int64_t* shape, int ndim);
int cc_clip_finalize(cc_clip* h);
/* precompute_embedding: x (B,3,S,S) float32 normalised pixels -> out (B,embed) float32 unit-norm. */
int cc_clip_encode_image(cc_clip* h, const float* x, int B, int x_on_device, float* out, int out_on_device, void* stream);
/* encode_text: tokens (B,t_ctx) int32 (SOT ... EOT, zero padded) -> out (B,embed) float32 unit-norm.
 * The reference only ever passes B=1 and picks row argmax(tokens) (= EOT) of batch 0. */
int cc_clip_encode_text(cc_clip* h, const int32_t* tokens, int B, float* out, int out_on_device, void* stream);
/* Batches in flight for the image tower: the contract of cc_yolo_set_in_flight / cc_yolo_submit / cc_yolo_wait above.  Pays for
 * small batches (the reference encodes one crop per call, models/objects.py:356-363); a 255-image batch fills the GPU by itself. */
int cc_clip_set_in_flight(cc_clip* h, int n);
int cc_clip_submit_image(cc_clip* h, const float* x, int B, int x_on_device, float* out, int out_on_device, void* stream,
                         long long* ticket);
int cc_clip_wait(cc_clip* h, long long ticket, void* stream);
int cc_clip_last_gpu_ms(cc_clip* h, float* ms);
void cc_clip_destroy(cc_clip* h);

/* `ObjectFinder.preprocess(img)` (models/objects.py:237-242) for a batch of crops: cv2.resize(img,(S,S),INTER_CUBIC)
 * (OpenCV 4.10 8-bit fixed-point cubic, requirements.txt:3) -> float32/255 -> (x-0.5)/0.5 -> HWC->CHW.
 * pixels: packed uint8 HWC crops (3 channels, channel order untouched), crop b = heights[b] x widths[b] starting at
 * byte offsets[b]; the three tables are host arrays.  out_dev: device float32 (B,3,S,S), ready for
 * cc_clip_encode_image(x_on_device=1).  Synchronises `stream` before returning. */
int cc_crop_preprocess(const uint8_t* pixels, const int64_t* offsets, const int32_t* heights, const int32_t* widths,
                       int B, int pixels_on_device, int out_size, float* out_dev, int device, void* stream);

/* ---------------------------------------------------------------------------------------------
 * AdaFace IR-50 face embedder — stands behind `ADAFACE.__call__(x)` (models/adaface.py:79-95), called as
 * `object_finder.adaface(Tensor(face_img))` (clearcam.py:674,1236) on a 112x112x3 aligned face.
 * Parameters by the reference's state-dict names (conv0.weight, bn0.*, prelu_weight, body.list.<i>.*, bn.*,
 * linear.*, bn2.running_*).  faces: (B,112,112,3) uint8 or float32 in the order the reference receives them
 * (it flips [:,:,::-1] itself) -> out (B,512) float32, each row x / ||x||_2.
 * ------------------------------------------------------------------------------------------- */
typedef struct cc_face cc_face;
int cc_face_create(cc_face** h, int dtype, int device);
int cc_face_load(cc_face* h, const char* name, const float* data, const int64_t* shape, int ndim);
int cc_face_finalize(cc_face* h);
int cc_face_embed(cc_face* h, const void* faces, int B, int img_f32, int faces_on_device, float* out, int out_on_device, void* stream);
void cc_face_destroy(cc_face* h);

/* The two OpenCV calls of `ObjectFinder.img_to_face` (models/objects.py:247,318,332) for uint8 HWC 3-channel host images:
 * cv2.resize(src, (dw,dh)) with the default INTER_LINEAR, and cv2.warpAffine(src, M, (dw,dh)) with the defaults
 * INTER_LINEAR / BORDER_CONSTANT 0 (M = the forward 2x3 matrix, row-major doubles, exactly what cv2 is given). */
int cc_cv_resize_linear_u8(const uint8_t* src, int H, int W, uint8_t* dst, int dh, int dw, int device);
int cc_cv_warp_affine_u8(const uint8_t* src, int H, int W, const double* M, uint8_t* dst, int dh, int dw, int device);

/* ---------------------------------------------------------------------------------------------
 * BlazeFace face detector — stands behind `BlazeFace.__call__(img)` (models/blazeface.py:165-192), called by
 * `ObjectFinder.img_to_face` (models/objects.py:253-255).  Parameters by the reference's state-dict names
 * (conv_tiny.*, backbone_tiny.list.<i>.conv{0,1}_tiny.*, final.*, classifier_{8,16}_tiny.*, regressor_{8,16}_tiny.*, anchors).
 * img: (H,W,3) uint8 or float32, channel order untouched -> out (896,17) float32 rows
 * [ymin,xmin,ymax,xmax, 6 x (kx,ky), score] in source pixels, score-descending, suppressed rows zeroed before the back-map.
 * ------------------------------------------------------------------------------------------- */
typedef struct cc_blaze cc_blaze;
int cc_blaze_create(cc_blaze** h, int dtype, int device);
int cc_blaze_load(cc_blaze* h, const char* name, const float* data, const int64_t* shape, int ndim);
int cc_blaze_finalize(cc_blaze* h);
int cc_blaze_detect(cc_blaze* h, const void* img, int H, int W, int img_f32, int img_on_device, float* out, int out_on_device, void* stream);
void cc_blaze_destroy(cc_blaze* h);

/* ---------------------------------------------------------------------------------------------
 * OC-SORT tracker (host code, no GPU) — stands behind `OCSort(...)` / `OCSort.update(preds, thresh)`
 * (ocsort_tracker/ocsort.py:164-308), the consumer of the detector output (clearcam.py:239,585).
 * dets: n rows [x1,y1,x2,y2,score,class] float32 exactly as cc_yolo_detect writes them (zero rows allowed).
 * out: up to cap rows of 9 doubles [tlx,tly,w,h,track_id,tracklet_len,class_id,score,speed] = the STrack fields
 * (ocsort_tracker/STrack.py:5-17), newest track first; *n_out = rows produced.  One handle per camera.
 * ------------------------------------------------------------------------------------------- */
typedef struct cc_ocsort cc_ocsort;
int cc_ocsort_create(cc_ocsort** h, int max_age, int min_hits, double iou_threshold, int delta_t, double inertia, int use_byte);
int cc_ocsort_update(cc_ocsort* h, const float* dets, int n, double det_thresh, double* out, int cap, int* n_out);
/* One frame for each of `count` cameras: camera c reads rows_per rows at dets + c*rows_per*6 (e.g. the (B,300,6) detector
 * output as is) and writes n_out[c] <= cap_per rows at out + c*cap_per*9; cameras run on up to n_threads host threads. */
int cc_ocsort_update_many(cc_ocsort* const* hs, int count, const float* dets, int rows_per, double det_thresh, double* out,
                          int cap_per, int* n_out, int n_threads);
int cc_ocsort_num_tracks(cc_ocsort* h, int* n);
void cc_ocsort_destroy(cc_ocsort* h);

/* ---------------------------------------------------------------------------------------------
 * Embedding index — stands behind the scoring loop of `ObjectFinder.search`
 * (models/objects.py:365-376: sim = img_emb @ text_emb.T per item) over a device-resident matrix.
 * Path filtering / best-per-track-id / sorting of the reference stay in the Python shim.
 * ------------------------------------------------------------------------------------------- */
typedef struct cc_index cc_index;
int cc_index_create(cc_index** h, int dim, int64_t capacity, int device);        /* f32 rows */
/* storage 0: f32 rows, exact f32 dot products.  storage 2: rows rounded to bf16 (half the bytes per scan; scores within
 * ~1e-3 of the f32 index for unit vectors; dim % 256 == 0).  `capacity` is the initial allocation: the matrix grows
 * geometrically (device-to-device copy) when cc_index_add* runs past it. */
int cc_index_create_ex(cc_index** h, int dim, int64_t capacity, int device, int storage);
int cc_index_add(cc_index* h, const float* emb, int64_t n, int on_device);       /* append n rows (n,dim) f32, group 0 */
/* the same with one group id (0 <= id < 2^24) per row on the HOST: the caller's (camera, day folder) bucket, the unit
 * `ObjectFinder.search` filters by (models/objects.py:368-371: `/cameras/<cam>/`, `/objects/<day>/` substrings) */
int cc_index_add_grouped(cc_index* h, const float* emb, int64_t n, int on_device, const int32_t* groups);
int cc_index_size(cc_index* h, int64_t* n);
int cc_index_info(cc_index* h, int64_t* capacity, int* storage, int* dim);       /* any pointer may be NULL */
int cc_index_scores(cc_index* h, const float* q, int Q, float* scores, int on_device, void* stream); /* (Q,N) */
/* top-k per query by score (ties: lower row id first): idx (Q,k) int32, score (Q,k) f32; rows past N: -1/-inf */
int cc_index_search(cc_index* h, const float* q, int Q, int k, int32_t* idx, float* score, int on_device, void* stream);
/* top-k restricted to rows whose group is allowed: `allowed` = n_groups bytes on the HOST (non-zero keeps the group),
 * NULL = no filter; fewer than k allowed rows -> -1/-inf padding */
int cc_index_search_groups(cc_index* h, const float* q, int Q, int k, const uint8_t* allowed, int n_groups, int32_t* idx, float* score,
                           int on_device, void* stream);
void cc_index_destroy(cc_index* h);

#ifdef __cplusplus
}
#endif
#endif

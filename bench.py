#!/usr/bin/env python3
"""Headline benchmark: YOLOv9-C 640x640 frames/s on MI355X (BASELINE.json configs[1]).

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the detect hot path (letterbox -> 144 convs -> decode -> top-300 + mask NMS)
over one batch of 64 synthetic 640x640 BGR uint8 frames that are already resident in HBM.  The headline storage mode is
"f16h": f16 activations; the stem conv's and the backbone's 1x1 convs' weights (blocks 0-9: where weight rounding is amplified by everything
downstream, and the filters controlled rounding cannot balance) carried as two f16 planes (W_hi + W_lo, f32 accumulation), every other
conv's as one f16 plane with controlled rounding - detections stay inside the reference tolerance with UN-ROUNDED float32 weights on three independently calibrated checkpoints (measured in
this run: `parity`).  "f16s" (two planes everywhere: weights exact to f32 for any checkpoint), plain f16 / bf16 (weights rounded to
11 / 8 bits, speed modes) and f32 (exact arithmetic) are reported beside it.  The K timed steps are
submitted round robin to `--in-flight` (default 3) slots of one handle (cc_yolo_submit: own stream, arena and graph
per slot, so consecutive batches overlap on the GPU; bit-identical rows) and all complete inside the timed region;
the same K steps as back-to-back cc_yolo_detect calls are reported beside it (`one_batch_in_flight`).  With N>1
(launched by torch.distributed.run, one rank per GPU) every rank runs its own 64 cameras' frames —
cameras shard one-per-GPU, there is no data-path collective for detection — and
value = all ranks' frames / max-over-ranks time ("weak" scaling).

Environment (tests only): CLEARCAM_BENCH_BACKEND=gloo runs the N>1 plumbing on CPU tensors with whatever model classes
the caller has put in place (tests/test_bench_multi.py mocks them); the default is "nccl" (RCCL) on cuda:LOCAL_RANK.

Rank 0 prints ONE JSON line with the contract fields plus
  dtype         the arithmetic type of the matrix work ("f16": f16 operands, f32 accumulation); `storage_mode` names the mode ("f16h", ...)
  roofline      dominant kernel family (the conv kernels): algorithmic FLOPs per step / their summed duration per
                step with ONE batch in flight, measured LIVE in this run (whole-step hipGraph minus non-conv hipGraph,
                hipEvents on the launch stream); the committed rocprofv3 trace of the same kernels (digest-, device- and
                dtype-checked) is the referee beside it (frac_rocprofv3).  Split weights issue two MFMAs per algorithmic
                multiply-add: `frac` counts the algorithmic FLOPs, `frac_mfma_issued` the matrix-pipe work
  cpu_baseline  the PyTorch-CPU fp32 oracle (restatement of the reference; tinygrad's CPU path cannot
                run offline) timed on this host's cores on a bounded sample, batch 1 as the reference runs it
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

PEAK_TFLOPS = {"bf16": 2500.0, "f16": 2500.0, "f16s": 2500.0, "f16h": 2500.0, "f16c": 2500.0, "f32": 157.3}   # dense MFMA peaks, MI355X_MICROARCH.md
MFMA_PER_MAC = {"f16s": 2.0}                                    # split weights: two matrix instructions per algorithmic multiply-add
                                                                # ("f16h": only the stem and the backbone's 1x1 convs - read off the per-launch table's weight_planes)
HBM_PEAK = 8.0e12
FLOP_PER_FRAME_C640 = 2 * 51.068e9


def _cpu_threads():
    # threads actually used: the host's usable cores, capped (oversubscribing a 256-thread box with one
    # batch-1 conv stream is slower than 16 threads; override with CLEARCAM_CPU_THREADS)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    return int(os.environ.get("CLEARCAM_CPU_THREADS", min(avail, 16))), avail


def _timed_loop(fn, seconds: float):
    """fn() once untimed (warm-up, also bounds the sample on a very slow host), then repeated for ~`seconds`."""
    t0 = time.time(); fn(); warm = time.time() - t0
    n, t0 = 0, time.time()
    while n < 1 or (time.time() - t0 < seconds and (time.time() - t0) + warm < 2 * seconds):
        fn(); n += 1
    return n, time.time() - t0


def cpu_baseline(size: str, res: int, seconds: float = 10.0, with_clip: bool = True) -> dict:
    """The CPU restatement of the reference (oracle/, PyTorch-CPU fp32 + numpy: tinygrad's CPU path cannot run offline) timed on
    this host's cores beside every GPU figure of the line: the detector at batch 1 (the reference's call) and
    batch 8, the ViT-L/14 image tower at batch 1 and 8, and the reference's Python search loop at 10 k / 100 k crops."""
    import torch
    from clearcam_amd.weights import synthetic_yolov9_state_dict
    from oracle.yolov9_oracle import YOLOv9Oracle
    cores, avail = _cpu_threads()
    torch.set_num_threads(cores)
    o = YOLOv9Oracle(size, res, synthetic_yolov9_state_dict(size, 1234))
    frames = np.random.default_rng(1).integers(0, 256, (8, res, res, 3), dtype=np.uint8)
    n1, t1 = _timed_loop(lambda: o(frames[0]), seconds)
    n8, t8 = _timed_loop(lambda: o.detect_batch(frames), seconds / 2)
    # the thread count is a choice, so it is measured: the batch-1 detector at 8 / 16 / 32 / 64 threads (~2 s each); `value` keeps the
    # default (CLEARCAM_CPU_THREADS or min(usable, 16)).  One batch-1 conv stream does not scale: on the 256-thread GPU box 16 threads
    # gave 12.8 frames/s, 64 threads 3.0 and all 256 threads 0.01 (one frame in 100 s; profiles/r04e_bench_line.json) - which is why
    # the sweep stops at 64 (CLEARCAM_CPU_SWEEP="16,64,256" overrides)
    sweep = {}
    want = [int(t) for t in os.environ.get("CLEARCAM_CPU_SWEEP", "8,16,32,64").split(",") if t.strip()]
    for th in sorted({min(avail, t) for t in want}):
        torch.set_num_threads(th)
        ns, ts = _timed_loop(lambda: o(frames[0]), 2.0)
        sweep[str(th)] = round(ns / ts, 3)
    torch.set_num_threads(cores)
    out = {"value": round(n1 / t1, 3), "unit": "frames/s", "cores": torch.get_num_threads(), "host_cores": os.cpu_count(),
           "frames_per_sec_by_threads": sweep,
           "host_cores_usable": avail, "kind": "port",
           "sample": f"{n1} frames of {res}x{res}, batch 1 (reference semantics), PyTorch-CPU fp32 restatement "
                     f"of detection/yolov9.py (tinygrad CPU path not runnable offline)",
           "yolo_batch8_frames_per_sec": round(8 * n8 / t8, 3), "yolo_batch8_sample": f"{n8} batches of 8"}
    if with_clip:
        from clearcam_amd.arch import CLIP_L14
        from clearcam_amd.weights import synthetic_clip_state_dict
        from oracle.clip_oracle import OpenCLIPOracle, search_reference
        oc = OpenCLIPOracle(synthetic_clip_state_dict(CLIP_L14, 4321), CLIP_L14)
        x = (np.random.default_rng(2).random((8, 3, 224, 224), dtype=np.float32) * 2 - 1).astype(np.float32)
        c1, d1 = _timed_loop(lambda: oc.precompute_embedding(x[:1]), seconds / 3)
        c8, d8 = _timed_loop(lambda: oc.precompute_embedding(x), seconds / 3)
        out["clip_l14_image_embeds_per_sec"] = {"batch1": round(c1 / d1, 3), "batch8": round(8 * c8 / d8, 3),
                                                "sample": f"{c1} x batch 1, {c8} x batch 8, oracle/clip_oracle.py (models/objects.py:94-133)"}
        # the reference's search: a Python loop over a dict of (1,768) arrays, one .item() per crop (models/objects.py:365-390)
        rng = np.random.default_rng(3)
        srch = {}
        for N in (10_000, 100_000):
            e = rng.standard_normal((N, 768), dtype=np.float32)
            e /= np.linalg.norm(e, axis=1, keepdims=True)
            table = {f"data/cameras/cam{i % 8}/objects/2026-01-0{1 + i % 7}/{1700000000 + i}_{i % 5000}_{i % 80}.jpg": e[i:i + 1] for i in range(N)}
            q = e[7:8].copy()
            t0 = time.time(); reps = 0
            while reps < 1 or time.time() - t0 < 1.0:
                search_reference(table, q, top_k=100); reps += 1
            srch[f"rows_{N}_ms_per_query"] = round((time.time() - t0) / reps * 1e3, 2)
        srch["what"] = "oracle.clip_oracle.search_reference (the reference's loop semantics), one query, k=100, single Python thread"
        out["search_reference_loop"] = srch
    return out


def measured_parity(device_index: int, n_cond: int = 16, n_chaotic: int = 4) -> dict:
    """Parity numbers computed IN THIS RUN (not quoted from a test log): every storage mode against the f32 CPU oracle on the same
    seeded frames.  f32 mode on the chaotic checkpoint (~260 detections per frame); f16 / bf16 on the well-conditioned checkpoint,
    once with 16-bit-exact weights (activation rounding only) and once with the weights un-rounded (the mode's own weight rounding
    inside the comparison).  A detection counts as matched only with the same class, IoU >= 0.9 AND all four coordinates within
    1e-3 * max(H, W) px of the oracle's (oracle.yolov9_oracle.match_detections_strict)."""
    import torch
    from clearcam_amd.weights import conditioned_yolov9_state_dict, synthetic_yolov9_state_dict
    from clearcam_amd.yolov9 import YOLOv9
    from oracle.yolov9_oracle import YOLOv9Oracle, decoded_rows, parity_summary, tolerance_bars
    cores, _ = _cpu_threads()
    torch.set_num_threads(cores)
    tol = 1e-3 * 640
    keys = ("n_ref", "n_got", "n_strict", "match_frac", "match_frac_clear_of_threshold", "match_frac_iou_only", "box_err_px_p50", "box_err_px_p99", "box_err_px_max_strict",
            "score_err_max", "anchor_box_err_px_p50", "anchor_box_err_px_p99", "anchor_box_err_px_p999", "anchor_box_err_px_max", "anchors_over_tol",
            "anchors_both_over_thr", "anchor_score_err_max", "box_tol_px")

    def oracle(sd, frames):
        o = YOLOv9Oracle("c", 640, sd)
        det, dec = [], []
        with torch.no_grad():
            for i in range(0, len(frames), 4):
                x = o.network_input(frames[i:i + 4])
                y = o.decode(o.head_raw(o.features(x)))
                dec.append(decoded_rows(y))
                det.append(o.scale_boxes((640, 640), o.postprocess(y), (640, 640)).numpy())
        return np.concatenate(det), np.concatenate(dec)

    def hip(sd, frames, dtype):
        m = YOLOv9("c", 640, state_dict=sd, dtype=dtype, device=device_index)
        got = m.detect_batch(frames)
        dec = m.get_tensor("decoded")
        m.close()
        return got, dec

    def summary(ref, got):
        s = parity_summary(ref[0], got[0], tol, ref[1], got[1])
        return {k: (round(s[k], 5) if isinstance(s[k], float) else s[k]) for k in keys}

    out = {"how": "measured in this run against the f32 CPU oracle on the same seeded frames; matched = same class, IoU >= 0.9 and all four "
                  "coordinates within 1e-3*max(H,W) = 0.64 px; anchor_* = the same anchor's decoded box / score wherever both sides score it over 0.25",
           "box_tol_px": tol, "match_frac_clear_of_threshold": "unmatched rows whose own score is within 2e-3 of the 0.25 threshold left out of the denominator"}
    fr = np.random.default_rng(1).integers(0, 256, (n_chaotic, 640, 640, 3), dtype=np.uint8)
    sd = synthetic_yolov9_state_dict("c", 1234)
    out["f32_chaotic_checkpoint"] = dict(summary(oracle(sd, fr), hip(sd, fr, "f32")), frames=n_chaotic)
    fr = np.random.default_rng(1).integers(0, 256, (n_cond, 640, 640, 3), dtype=np.uint8)
    modes = ("f16h", "f16s", "f16", "bf16")
    # three independently calibrated conditioned checkpoints (clearcam_amd/assets/synth_cond_c*.npz: another seed's base filters, its own
    # data-dependent gains and biases); the 16-bit-exact variant (activation rounding only) on the first
    for label, seed, exact in (("weights_16bit_exact", 1234, True), ("weights_unrounded", 1234, False), ("weights_unrounded_checkpoint_b", 7, False),
                               ("weights_unrounded_checkpoint_c", 99, False)):
        sd = conditioned_yolov9_state_dict("c", seed, exact=exact)
        ref = oracle(sd, fr)
        out[label] = {dt: summary(ref, hip(sd, fr, dt)) for dt in modes}
        if not exact:                                        # the calibrated mode (round 5): the library's default calibration (seeded noise inside cc_yolo_finalize)
            out[label]["f16c"] = summary(ref, hip(sd, fr, "f16c"))
        out[label]["frames"] = n_cond
    # stress variants of the conditioned checkpoint (clearcam_amd.weights.COND_STRESS: f32 perturbation gain ~3-4 and ~8-14 from the input to
    # P3..P5 instead of ~1.5): how far the tolerance claim reaches.  Reported, not part of `holds_tolerance_*`
    for label, stress in (("weights_unrounded_gain3", "g3"), ("weights_unrounded_gain10", "g10")):
        try:
            sd = conditioned_yolov9_state_dict("c", 1234, exact=False, stress=stress)
            ref = oracle(sd, fr)
            out[label] = {dt: summary(ref, hip(sd, fr, dt)) for dt in ("f16h", "f16s", "f16c", "f16")}
            out[label]["frames"] = n_cond
        except Exception as exc:                 # noqa: BLE001  a side measurement
            out[label] = {"error": f"{type(exc).__name__}: {exc}"}
    out["conditioning_limit_note"] = ("measured f32 perturbation gains in clearcam_amd/assets/synth_cond_report.json (c, c_s7, c_s99: 1.4-1.9; c_g3: 2.8-3.9; c_g10: 8-13; "
                                      "the chaotic seeded checkpoint: 30-60).  At gain ~3 the f16-activation modes keep the median anchor within 0.03 px and 99 % "
                                      "within ~1 px, with every anchor beyond the tolerance in ONE frame of 128; at gain ~10 no 16-bit mode holds anything "
                                      "(exact-weight f16s included: activation rounding) - profiles/r05p_stress_128.txt")
    # the bars a mode has to hold WITH UN-ROUNDED WEIGHTS (a trained checkpoint is float32: detection/yolov9.py:372-373) to carry the
    # north star's "within 1e-3" - oracle.yolov9_oracle.tolerance_bars: >= 98.5 % strict matches clear of the 0.25 threshold (>= 97.5 % with
    # every row counted), scores within 2e-3, 99.9 % of the anchors within 1e-3 * max(H, W) and none beyond 1.5x that - on EVERY one of the
    # three checkpoints
    def holds(s):
        return bool(tolerance_bars(s)["all"])
    unrounded = ("weights_unrounded", "weights_unrounded_checkpoint_b", "weights_unrounded_checkpoint_c")
    modes = modes + ("f16c",)
    out["holds_tolerance_with_unrounded_weights"] = {dt: all(holds(out[label][dt]) for label in unrounded) for dt in modes}
    out["worst_anchor_box_err_px_with_unrounded_weights"] = {dt: max(out[label][dt]["anchor_box_err_px_max"] for label in unrounded) for dt in modes}
    # the stricter form the bars had until round 4 - EVERY anchor within the tolerance - reported, not part of `holds`: the worst anchor of a
    # frame set is a heavy-tailed draw of the f16 activation rounding ("The tail")
    out["every_anchor_within_tolerance_with_unrounded_weights"] = {dt: all(out[label][dt]["anchors_over_tol"] == 0 for label in unrounded) for dt in modes}
    out["holds_tolerance_note"] = ("99.9 % of the anchors within 0.64 px and none beyond 0.96 px, scores within 2e-3, >= 98.5 % strict matches clear of the threshold, against the "
                                   "f32 oracle on the conditioned checkpoint with its float32 weights NOT pre-rounded; f16 / bf16 round their weights with controlled rounding "
                                   "(yolo.hip round_controlled: filter sums preserved), f16s carries them as two f16 planes, f16h as two planes in the stem conv and the backbone's "
                                   "1x1 convs (blocks 0-9) and one controlled-rounded plane elsewhere, f16c as one plane everywhere but the stem with the 1x1 convs' weights "
                                   "rounded by the calibration-aware recursion on the library's default calibration frames (seeded noise, i.e. the distribution of "
                                   "these test frames: the matched case - tests/test_gpu_yolo.py::test_calibrated_mode_* covers mismatched calibration); three "
                                   "independently calibrated checkpoints, all must hold.  These are 16-frame samples of a heavy-tailed statistic: "
                                   "profiles/r05o_tail_256.txt has 256 frames per checkpoint (on checkpoint 1234 one frame in ~256 moves boxes by tens of "
                                   "pixels in EVERY f16-activation mode, exact-weight f16s included)")
    return out


def clip_side_metrics(device_index: int, dev) -> dict:
    """Second half of BASELINE.json's metric: CLIP ViT-L/14 image embeds/s (configs[2]) and the search scan
    (configs[4], one GPU's shard).  Same rules: seeded synthetic weights, inputs resident in HBM, hipEvent/sync timing."""
    import torch
    from clearcam_amd.arch import CLIP_L14
    from clearcam_amd.objects import EmbeddingIndex, OpenCLIP
    from clearcam_amd.weights import synthetic_clip_state_dict
    out = {}
    m = OpenCLIP(state_dict=synthetic_clip_state_dict(CLIP_L14, 4321), arch=CLIP_L14, dtype="bf16", device=device_index)

    def rate(B, reps):
        x = torch.rand(B, 3, 224, 224, device=dev) * 2 - 1
        emb = torch.empty(B, 768, device=dev)
        for _ in range(2):
            m.precompute_embedding_device(x, emb)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            m.precompute_embedding_device(x, emb)
        torch.cuda.synchronize()
        return B / ((time.perf_counter() - t0) / reps)
    B = 255                                      # 255 images x 257 tokens = 65535 rows: whole rounds of 256-row GEMM tiles on 256 CUs
    r = rate(B, 4)
    out["model"] = "ViT-L/14 (the model the reference runs, models/objects.py:29-89)"
    out["image_embeds_per_sec"] = round(r, 1)
    out["image_batch"] = B
    out["image_tflops"] = round(r * 162.03e9 / 1e12, 1)
    out["image_frac_of_mfma_peak"] = round(r * 162.03e9 / 1e12 / PEAK_TFLOPS["bf16"], 4)
    out["seconds_per_10k_crops"] = round(10000 / r, 3)
    # the reference's own speed test sweeps the batch size (test/test_clip_speed.py:8-15: bs = 1, 2, 4, ... 128)
    out["image_embeds_per_sec_by_batch"] = {str(b): round(rate(b, 8 if b <= 16 else 4), 1) for b in (1, 2, 4, 8, 16, 32, 64, 128)}
    # ... and with three small batches in flight (cc_clip_submit_image: slots with their own streams; the reference encodes one crop
    # per call, so many cameras' crops arrive as many small batches)
    try:
        m.set_in_flight(3)
        def rate3(B, reps):
            x = torch.rand(B, 3, 224, 224, device=dev) * 2 - 1
            embs = [torch.empty(B, 768, device=dev) for _ in range(3)]
            for k in range(3):
                m.submit_image(x, embs[k])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(reps):
                m.submit_image(x, embs[k % 3])
            torch.cuda.synchronize()
            return B / ((time.perf_counter() - t0) / reps)
        out["image_embeds_per_sec_by_batch_3_in_flight"] = {str(b): round(rate3(b, 24 if b <= 16 else 9), 1) for b in (1, 2, 4, 8, 16, 32, 64, 128, 255)}
        m.set_in_flight(1)
    except Exception as exc:                     # noqa: BLE001  a side metric
        out["image_embeds_per_sec_by_batch_3_in_flight"] = f"error: {type(exc).__name__}: {exc}"
    toks = np.zeros((64, 77), np.int32)
    toks[:, 0] = 49406
    toks[:, 1:5] = [9606, 325, 275, 271]
    toks[:, 5] = 49407
    m.encode_tokens(toks)
    t0 = time.perf_counter()
    m.encode_tokens(toks)
    out["text_embeds_per_sec"] = round(64 / (time.perf_counter() - t0), 1)
    m.close()
    # ViT-B/32 (named in BASELINE.json's north star; NOT the model the reference runs and not the 10k-crops target): same engine
    from clearcam_amd.arch import CLIP_B32
    mb = OpenCLIP(state_dict=synthetic_clip_state_dict(CLIP_B32, 99), arch=CLIP_B32, dtype="bf16", device=device_index)
    Bb = 1024
    xb = torch.rand(Bb, 3, 224, 224, device=dev) * 2 - 1
    eb = torch.empty(Bb, 512, device=dev)
    for _ in range(2):
        mb.precompute_embedding_device(xb, eb)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(4):
        mb.precompute_embedding_device(xb, eb)
    torch.cuda.synchronize()
    dtb = (time.perf_counter() - t0) / 4
    out["vit_b32_image_embeds_per_sec"] = round(Bb / dtb, 1)
    out["vit_b32_seconds_per_10k_crops"] = round(10000 / (Bb / dtb), 3)
    mb.close()
    del xb, eb
    # face path: AdaFace IR-50 embeddings and BlazeFace detections, seeded weights
    from clearcam_amd.adaface import ADAFACE
    from clearcam_amd.blazeface import BlazeFace
    from clearcam_amd.weights import synthetic_adaface_state_dict, synthetic_blazeface_state_dict
    fa = ADAFACE(state_dict=synthetic_adaface_state_dict(777), dtype="bf16", device=device_index)
    faces = np.random.default_rng(5).integers(0, 256, (256, 112, 112, 3), dtype=np.uint8)
    for _ in range(2):
        fa.embed_batch(faces)
    t0 = time.perf_counter()
    for _ in range(3):
        fa.embed_batch(faces)
    out["adaface_faces_per_sec"] = round(3 * 256 / (time.perf_counter() - t0), 1)       # host in, host out (PCIe included)
    fa.close()
    bz = BlazeFace(state_dict=synthetic_blazeface_state_dict(555), dtype="bf16", device=device_index)
    img = np.random.default_rng(6).integers(0, 256, (640, 640, 3), dtype=np.uint8)
    for _ in range(3):
        bz(img)
    t0 = time.perf_counter()
    for _ in range(20):
        bz(img)
    out["blazeface_ms_per_image"] = round((time.perf_counter() - t0) / 20 * 1e3, 3)     # batch 1, as the reference calls it
    bz.close()
    # search (BASELINE.json configs[4]): k=100, 1 and 64 queries, host-observed call latency (query upload + scan + top-k +
    # result read-back).  125 k rows = one GPU's shard of 1 M vectors over 8 GPUs; 1 M rows = the whole index on ONE GPU
    # (3.07 GB f32 / 1.54 GB bf16: HBM-bound scans).  bf16 rows change scores by <= 1e-3 (tests/test_gpu_clip.py).
    def search_case(N, storage):
        ix = EmbeddingIndex(768, N, device=device_index, storage=storage)
        g = torch.Generator(device=dev).manual_seed(3)
        for i in range(0, N, 125_000):
            e = torch.randn(min(125_000, N - i), 768, device=dev, generator=g)
            e /= e.norm(dim=1, keepdim=True)
            ix.add(e)
        res = {"rows": N, "storage": storage, "k": 100}
        for Q, reps in ((1, 200), (64, 20)):
            q = torch.randn(Q, 768, generator=torch.Generator().manual_seed(7))
            q = (q / q.norm(dim=1, keepdim=True)).numpy()
            for _ in range(3):
                ix.search(q, 100)
            lat = []
            for _ in range(reps):
                t0 = time.perf_counter()
                ix.search(q, 100)
                lat.append(time.perf_counter() - t0)
            lat.sort()
            res[f"q{Q}_p50_ms"] = round(lat[len(lat) // 2] * 1e3, 3)
            res[f"q{Q}_p99_ms"] = round(lat[min(len(lat) - 1, int(len(lat) * 0.99))] * 1e3, 3)
        res["q1_end_to_end_GBps"] = round(N * 768 * (4 if storage == "f32" else 2) / (res["q1_p50_ms"] * 1e-3) / 1e9, 1)
        ix.close()
        return res
    out["search"] = {"shard_125k_f32": search_case(125_000, "f32"), "one_gpu_1M_f32": search_case(1_000_000, "f32"),
                     "one_gpu_1M_bf16": search_case(1_000_000, "bf16")}
    return out


def sharded_search_metrics(device_index: int, dev, world: int, rank: int) -> dict:
    """BASELINE.json configs[4]: cross-camera search over a row-sharded index, 125 k x 768 f32 rows per GPU (1 M rows at
    8 GPUs).  Every rank scans its shard (HBM), the per-rank top-k lists meet in ONE all-gather over RCCL/xGMI and are
    merged on every rank (clearcam_amd/dist.py).  Latency = max over ranks of each rank's median, host-observed."""
    import torch
    import torch.distributed as dist
    from clearcam_amd.dist import ShardedIndex
    from clearcam_amd.objects import EmbeddingIndex
    N = int(os.environ.get("CLEARCAM_BENCH_SHARD_ROWS", 125_000))
    ix, err = None, ""
    try:                                         # local set-up may fail on one rank (memory): agree before any collective
        ix = EmbeddingIndex(768, N, device=device_index)
        g = torch.Generator(device=dev).manual_seed(100 + rank)
        e = torch.randn(N, 768, device=dev, generator=g)
        e /= e.norm(dim=1, keepdim=True)
        ix.add(e)
    except Exception as exc:                     # noqa: BLE001
        err = f"{type(exc).__name__}: {exc}"
    ok = torch.tensor([0.0 if err else 1.0], dtype=torch.float64, device=dev)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if float(ok.item()) < 1.0:
        if ix is not None:
            ix.close()
        return {"error": err or "another rank failed to build its shard"}
    sh = ShardedIndex(ix, device=dev)
    q = torch.randn(1, 768, generator=torch.Generator().manual_seed(7))
    q /= q.norm()
    qn = q.numpy()
    for _ in range(3):
        ids, sc = sh.search(qn, 100)
    lat = []
    for _ in range(20):                          # the all-gather inside every search keeps the ranks in step
        t0 = time.perf_counter()
        ids, sc = sh.search(qn, 100)
        lat.append(time.perf_counter() - t0)
    lat.sort()
    t = torch.tensor([lat[len(lat) // 2], lat[-1]], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ok = bool((np.diff(sc[0]) <= 0).all() and (ids[0] >= 0).all() and ids[0].max() < sh.total)
    ix.close()
    return {"rows_total": int(sh.total), "rows_per_gpu": N, "k": 100, "p50_ms": round(float(t[0]) * 1e3, 3),
            "max_ms": round(float(t[1]) * 1e3, 3), "exchange_bytes_per_rank": 100 * 16, "result_sorted_and_in_range": ok,
            "exchange": "device-resident: scan + top-k, one all-gather of (1,2,100) int64 per rank, merge on the GPU, one copy out"}


def stream_side_metrics(device_index: int, size: str, res: int, dtype: str) -> dict:
    """BASELINE.json configs[3]: synthetic 1080p cameras -> letterbox -> detect -> OC-SORT on one GPU, end to end
    INCLUDING the PCIe upload of every frame (pinned rings, async copies, two batches in flight; clearcam_amd/streams.py).
    Class biases are shifted so that the seeded detector reports a realistic ~25 objects per frame to the tracker
    (the FLOPs are unchanged); the crowded run keeps all ~260 noise detections per frame as a tracker worst case."""
    from clearcam_amd.streams import StreamPipeline, make_cameras
    from clearcam_amd.weights import shift_class_bias, synthetic_yolov9_state_dict
    from clearcam_amd.yolov9 import YOLOv9
    out = {}
    sd = synthetic_yolov9_state_dict(size, 1234)
    banks = {64: make_cameras(64), 8: make_cameras(8)}          # one pinned bank per camera count: a tick goes up as ONE copy
    # (the 8-camera leg - detector slots - runs FIRST and the plain pipelines after it: the order that replayed graphs slowly in round 3;
    # tests/test_gpu_streams.py::test_pipeline_after_slotted_pipeline_replays_at_full_speed holds it)
    for key, n, shift in (("cams8", 8, -20.0), ("cams64", 64, -20.0), ("cams64_crowded", 64, 0.0)):
        m = YOLOv9(size, res, state_dict=shift_class_bias(sd, shift), dtype=dtype, device=device_index)
        pipe = StreamPipeline(m, n)
        cams = banks[n]
        if hasattr(cams, "t"):
            cams.t = 0
        st = pipe.run(cams, 24 if n > 8 else 60)
        out[key] = {k: (round(v, 3) if isinstance(v, float) else v) for k, v in st.items()}
        pipe.close(); m.close()
    return out


def kernel_source_digest() -> str:
    """sha256[:16] over clearcam_amd/csrc/*: ties a PMC measurement under profiles/ to the kernels it was taken from."""
    import hashlib
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "clearcam_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(root)):
        if f.endswith((".hip", ".h", ".cpp")):
            h.update(f.encode()); h.update(open(os.path.join(root, f), "rb").read())
    return h.hexdigest()[:16]


def measured_traffic(default_cfg: bool, dtype: str = "f16h"):
    """HBM bytes per step of the conv kernels from this round's rocprofv3 PMC passes (profiles/pmc_traffic.json, written
    by tools/pmc_traffic.py on the GPU box).  A file taken from other kernels than the ones in the tree is NOT quoted."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_traffic.json")
    if not default_cfg:
        return None, "PMC traffic is recorded for the headline configuration only"
    if not os.path.exists(path):
        return None, "profiles/pmc_traffic.json missing: run tools/pmc_traffic.py on the GPU box"
    rec = json.load(open(path))
    if rec.get("dtype", "f16") != dtype:
        return None, f"profiles/pmc_traffic.json was recorded for dtype {rec.get('dtype', 'f16')}"
    if rec.get("kernel_source_digest") != kernel_source_digest():
        print(f"bench: profiles/pmc_traffic.json was measured on other kernels ({rec.get('kernel_source_digest')} != "
              f"{kernel_source_digest()}); roofline.traffic left null — re-run tools/pmc_traffic.py", file=sys.stderr)
        return None, "stale: kernels changed since the PMC passes in profiles/pmc_traffic.json; re-run tools/pmc_traffic.py"
    return float(rec["conv_bytes_per_step"]), rec.get("note", "")


def traced_kernel_ms(default_cfg: bool, dtype: str, device_name: str):
    """Conv-family kernel time per step from this round's rocprofv3 --kernel-trace --stats run (profiles/kernel_trace.json, written
    by tools/measure_round.sh on the GPU box): the REFEREE beside the live measurement, quoted only for the kernel sources (digest),
    storage mode and device model it was taken on."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "kernel_trace.json")
    if not default_cfg or not os.path.exists(path):
        return None
    rec = json.load(open(path))
    ok = rec.get("kernel_source_digest") == kernel_source_digest() and rec.get("dtype") == dtype and rec.get("device_name") in (None, device_name)
    return rec if ok else None


MAX_LINE_BYTES = 4096                                           # the driver keeps an 8 KB tail of stdout: the final line must fit with room to spare


def compact_line(d: dict) -> dict:
    """The ONE stdout line (contract fields + roofline + cpu_baseline + a parity verdict, < 4 KB).  Everything else this run
    measured stays in the detail record (bench_detail.json + stderr): VERDICT r5 item 1 - the r05 line had grown to 25 KB and the
    driver's 8 KB tail could not parse it."""
    def pick(src, keys):
        return {k: src[k] for k in keys if isinstance(src, dict) and k in src and src[k] is not None}
    out = pick(d, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling"))
    out["vs_baseline"] = d.get("vs_baseline")
    out.update(pick(d, ("dtype", "storage_mode", "data")))
    cfg = d.get("config") or {}
    out["config"] = pick(cfg, ("workload", "batch_per_gpu", "batches_in_flight", "parallelism"))
    r = d.get("roofline") or {}
    roof = pick(r, ("bound", "achieved", "peak", "unit", "frac", "frac_rocprofv3", "kernel_ms_per_step", "kernel_ms_per_step_rocprofv3",
                    "traffic", "launches_per_step", "frac_mfma_issued"))
    roof["kernel"] = "every conv launch of the step (implicit-GEMM MFMA kernels), one batch in flight"
    if isinstance(r.get("per_launch_roof"), dict):
        roof["per_launch_roof_frac"] = r["per_launch_roof"].get("frac")
        tq = r["per_launch_roof"].get("tile_quantisation")
        if isinstance(tq, dict):
            roof["idle_slot_ms"] = tq.get("idle_slot_ms")
    roof.setdefault("traffic", None)
    roof["traffic_unit"] = "HBM bytes per step, rocprofv3 PMC (profiles/pmc_traffic.json), conv launches"
    out["roofline"] = roof
    c = d.get("cpu_baseline")
    if isinstance(c, dict):
        cb = pick(c, ("value", "unit", "cores", "kind", "host_cores"))
        cb["sample"] = str(c.get("sample", ""))[:160]
        out["cpu_baseline"] = cb
    p = d.get("parity")
    if isinstance(p, dict):
        if "error" in p:
            out["parity"] = {"error": str(p["error"])[:200]}
        else:
            out["parity"] = {"holds": p.get("holds_tolerance_with_unrounded_weights"),
                             "worst_anchor_px": p.get("worst_anchor_box_err_px_with_unrounded_weights"),
                             "every_anchor": p.get("every_anchor_within_tolerance_with_unrounded_weights"),
                             "f32_gate": pick(p.get("f32_chaotic_checkpoint") or {}, ("match_frac", "box_err_px_max_strict", "score_err_max", "anchor_box_err_px_max"))}
    one = d.get("one_batch_in_flight")
    if isinstance(one, dict):
        out["one_batch_in_flight_frames_per_sec"] = one.get("frames_per_sec")
    if isinstance(d.get("frames_per_sec_by_storage_dtype"), dict):
        out["frames_per_sec_by_storage_dtype"] = d["frames_per_sec_by_storage_dtype"]
    if isinstance(d.get("single_frame"), dict):
        out["single_frame_ms_p50"] = d["single_frame"].get("ms_p50")
    cl = d.get("clip")
    if isinstance(cl, dict):
        out["clip"] = pick(cl, ("image_embeds_per_sec", "image_batch", "image_frac_of_mfma_peak", "seconds_per_10k_crops"))
    st = d.get("streams")
    if isinstance(st, dict):
        out["streams"] = {k: (pick(v, ("frames_per_sec", "fps_per_camera")) if isinstance(v, dict) else v) for k, v in st.items() if not isinstance(v, str)}
    out.update(pick(d, ("ranks_seen", "ms_per_step_by_rank", "storage_mode_by_rank", "streams_multi_gpu")))
    sh = d.get("search_sharded")
    if isinstance(sh, dict):
        out["search_sharded"] = {k: v for k, v in sh.items() if not isinstance(v, (str, dict, list)) or k == "error"}
    out["detail"] = "bench_detail.json"
    return out


def emit(detail: dict) -> None:
    """Write the full record to bench_detail.json (repo root, and gpurun_out/ when it exists) and to stderr; print the compact line
    as the LAST line of stdout.  Optional blocks are dropped, least important first, should the line ever exceed MAX_LINE_BYTES."""
    root = os.path.dirname(os.path.abspath(__file__))
    blob = json.dumps(detail, indent=1)
    for path in (os.path.join(root, "bench_detail.json"), os.path.join(root, "gpurun_out", "bench_detail.json")):
        try:
            if os.path.isdir(os.path.dirname(path)):
                with open(path, "w") as f:
                    f.write(blob)
        except OSError as exc:                   # a read-only tree must not cost the line
            print(f"bench: could not write {path}: {exc}", file=sys.stderr)
    print("bench detail: " + json.dumps(detail), file=sys.stderr, flush=True)
    line = compact_line(detail)
    for drop in ("streams", "search_sharded", "streams_multi_gpu", "clip", "frames_per_sec_by_storage_dtype", "parity"):
        if len(json.dumps(line)) < MAX_LINE_BYTES:
            break
        line.pop(drop, None)
    text = json.dumps(line)
    assert len(text) < MAX_LINE_BYTES, len(text)
    sys.stdout.flush()
    print(text, flush=True)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--size", default="c")
    ap.add_argument("--res", type=int, default=640)
    ap.add_argument("--height", type=int, default=0, help="source frame height (default: res)")
    ap.add_argument("--width", type=int, default=0, help="source frame width (default: res)")
    ap.add_argument("--dtype", default="f16h", choices=["f16h", "f16s", "f16c", "bf16", "f16", "f32"],
                    help="storage mode.  f16h (default): f16 activations, the stem's and the backbone's 1x1 convs' weights as two f16 planes, one controlled-rounded plane elsewhere - "
                         "holds the parity yardstick with un-rounded float32 weights (DESIGN.md section 5); f16s: two planes in every conv.  f16 / bf16: speed modes (weights rounded to 11 / 8 bits); f32: exact")
    ap.add_argument("--in-flight", type=int, default=3,
                    help="batches in flight (cc_yolo_submit on that many slots of one handle: the last layers of one batch overlap the first "
                         "layers of the next); 1 = back-to-back cc_yolo_detect calls.  Both are measured; `value` is this mode")
    ap.add_argument("--no-parity", action="store_true", help="skip the in-run parity measurement against the CPU oracle")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-clip", action="store_true", help="skip the CLIP / search side metrics")
    ap.add_argument("--no-streams", action="store_true", help="skip the camera-pipeline side metrics (detect -> OC-SORT incl. PCIe)")
    ap.add_argument("--no-precisions", action="store_true", help="skip the f16 / f32 throughput legs")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from clearcam_amd.weights import synthetic_yolov9_state_dict
    from clearcam_amd.yolov9 import YOLOv9

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("CLEARCAM_BENCH_BACKEND", "nccl")
    on_gpu = backend == "nccl"
    dev = torch.device("cuda", local) if on_gpu else torch.device("cpu")
    sync = torch.cuda.synchronize if on_gpu else (lambda: None)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if on_gpu:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    if args.gpus != world and rank == 0 and world > 1:
        print(f"warning: --gpus {args.gpus} != WORLD_SIZE {world}", file=sys.stderr)
    if on_gpu:
        torch.cuda.set_device(local)

    sd = synthetic_yolov9_state_dict(args.size, 1234)
    B = args.batch
    fh, fw = args.height or args.res, args.width or args.res
    frames = torch.from_numpy(np.random.default_rng(1 + rank).integers(0, 256, (B, fh, fw, 3), dtype=np.uint8)).to(dev)
    out = torch.empty((B, 300, 6), dtype=torch.float32, device=dev)

    def barrier():
        if world > 1:
            dist.barrier()

    depth = max(1, args.in_flight)
    outs = [out] + [torch.empty_like(out) for _ in range(depth - 1)]      # one result buffer per batch in flight

    def runner(m, d):
        """one step of the timed loop: step k goes to slot k mod d (d = 1: the synchronous-order cc_yolo_detect call)"""
        if d > 1:
            return lambda k: m.submit(frames, outs[k % d])
        return lambda k: m.detect_batch_device(frames, out)

    def timed(m, steps, warmup, d=1):
        """W untimed steps, then exactly K steps between barrier + synchronize on both sides; max over ranks."""
        run = runner(m, d)
        for k in range(d if d > 1 else 0):
            run(k)                                   # every slot builds its plan (arena, graph) before the warm-up steps
        for k in range(warmup):
            run(k)
        sync(); barrier(); sync()
        t0 = time.perf_counter()
        for k in range(steps):
            run(k)
        sync(); barrier(); sync()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    def in_flight_model(dt_name):
        m = YOLOv9(args.size, args.res, state_dict=sd, dtype=dt_name, device=local)
        if depth > 1:
            m.set_in_flight(depth)
        return m

    # the headline: K batches through `depth` slots of one handle; next to it the same K batches as back-to-back cc_yolo_detect calls.
    # (cc_yolo_set_in_flight probes its slots' streams until they really run side by side: the runtime maps streams onto a few
    # hardware queues and two slots on one queue would not overlap)
    model_p = in_flight_model(args.dtype)
    elapsed = timed(model_p, args.steps, args.warmup, depth)
    model = YOLOv9(args.size, args.res, state_dict=sd, dtype=args.dtype, device=local) if depth > 1 else model_p
    elapsed_one = timed(model, args.steps, args.warmup, 1) if depth > 1 else elapsed
    # What the collective backend actually saw: an all-reduce of ones (= the number of ranks that took part) and every rank's own
    # K-step time, so that the driver's scaling record can check "N ranks over RCCL" against the line instead of trusting --gpus.
    ranks_seen, per_rank_ms, per_rank_mode = 1, None, None
    if world > 1:
        ones = torch.ones(1, dtype=torch.float64, device=dev)
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)
        ranks_seen = int(round(float(ones.item())))
        run = runner(model_p, depth)
        sync(); t0 = time.perf_counter()
        for k in range(args.steps):
            run(k)
        sync()
        mine = torch.zeros(world, dtype=torch.float64, device=dev)
        mine[rank] = (time.perf_counter() - t0) / args.steps * 1e3
        dist.all_reduce(mine, op=dist.ReduceOp.SUM)
        per_rank_ms = [round(float(v), 3) for v in mine.tolist()]
        from clearcam_amd.yolov9 import DTYPES                     # the storage mode every rank really ran (C-ABI dtype code), checkable in a SCALE record
        codes = torch.zeros(world, dtype=torch.float64, device=dev)
        codes[rank] = float(DTYPES[args.dtype])
        dist.all_reduce(codes, op=dist.ReduceOp.SUM)
        names = {v: k for k, v in DTYPES.items() if k in ("f32", "f16", "bf16", "f16s", "f16h", "f16c")}
        per_rank_mode = [names.get(int(round(float(v))), "?") for v in codes.tolist()]
    # what the batches in flight cost: submit-to-result latency of one batch in the steady state (host clock, result waited for on the host)
    latency = None
    if world == 1 and depth > 1 and on_gpu:
        try:
            run = runner(model_p, depth)
            tk, t_sub, lat = [], [], []
            for k in range(4 * depth + 12):
                if len(tk) == depth:
                    model_p.wait(tk.pop(0), host=True); lat.append(time.perf_counter() - t_sub.pop(0))
                t_sub.append(time.perf_counter()); tk.append(run(k))
            for t in tk:
                model_p.wait(t, host=True)
            lat = sorted(lat[depth:])
            latency = {"p50": round(lat[len(lat) // 2] * 1e3, 3), "max": round(lat[-1] * 1e3, 3), "batches_in_flight": depth,
                       "one_in_flight": round(elapsed_one / args.steps * 1e3, 3)}
        except Exception as exc:                 # noqa: BLE001  a side metric
            latency = {"error": f"{type(exc).__name__}: {exc}"}
    if model_p is not model:
        model_p.close()
    n_det = int((out[..., 4] > 0).sum().item())

    # >= 100 timed iterations, median.  Per-step wall times on this rank (each step synchronised), next to
    # the contract's K-step total above.
    median_100 = None
    if world == 1:
        per = []
        for _ in range(100):
            sync(); t1 = time.perf_counter()
            model.detect_batch_device(frames, out)
            sync(); per.append(time.perf_counter() - t1)
        per.sort()
        median_100 = {"steps": 100, "ms_median": round(per[50] * 1e3, 3), "ms_p10": round(per[10] * 1e3, 3), "ms_p90": round(per[90] * 1e3, 3),
                      "frames_per_sec_at_median": round(B / per[50], 1)}

    # the other storage modes on the same workload (weights re-packed, same frames): the f32 parity mode is the one whose
    # detections match the reference path to the north star's 1e-3; bf16 / f16 are held to the 16-bit bars (tests/)
    precisions = None
    if not args.no_precisions and world == 1:
        precisions = {args.dtype: round(B * args.steps / elapsed, 1)}
        for dt_name, steps in (("f16s", args.steps), ("f16h", args.steps), ("f16c", args.steps), ("f16", args.steps), ("bf16", args.steps), ("f32", max(3, args.steps // 4))):
            if dt_name in precisions:
                continue
            try:
                m2 = in_flight_model(dt_name)
                precisions[dt_name] = round(B * steps / timed(m2, steps, 2, depth), 1)
                m2.close()
            except Exception as exc:             # noqa: BLE001  a side metric
                precisions[dt_name] = f"error: {type(exc).__name__}: {exc}"

    streams_multi = None
    if world > 1 and not args.no_streams:
        # BASELINE.json configs[3] on the whole node: every rank drives its own cameras (one camera -> one GPU, no collective
        # in the data path); the job-wide rate is the sum over ranks, the slowest camera anywhere bounds the per-stream FPS.
        # Local work is exception-safe and the only collective (one all-reduce) is reached by every rank whatever happened.
        stats = torch.zeros(2, 4, dtype=torch.float64)          # per config: frames/s, fps per camera, H2D GB/s, ok flag
        err = ""
        for ci, n_cams in enumerate((64, 8)):
            try:
                from clearcam_amd.streams import StreamPipeline, make_cameras
                from clearcam_amd.weights import shift_class_bias
                m2 = YOLOv9(args.size, args.res, state_dict=shift_class_bias(sd, -20.0), dtype=args.dtype, device=local)
                pipe = StreamPipeline(m2, n_cams, n_threads=max(1, min(32, (os.cpu_count() or 8) // world)))
                cams = make_cameras(n_cams, seed=100 + rank)
                st = pipe.run(cams, 60 if n_cams == 8 else 24)
                stats[ci] = torch.tensor([st["frames_per_sec"], st["fps_per_camera"], st["h2d_GBps"], 1.0], dtype=torch.float64)
                pipe.close(); m2.close(); del cams
            except Exception as exc:             # noqa: BLE001
                err = f"{type(exc).__name__}: {exc}"
        tot = stats.clone().to(dev); dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        neg = (-stats[:, 1]).clone().to(dev); dist.all_reduce(neg, op=dist.ReduceOp.MAX)      # min over ranks of fps per camera
        streams_multi = {}
        for ci, n_cams in enumerate((64, 8)):
            streams_multi[f"cams{n_cams}_per_gpu"] = {"cameras_total": n_cams * world, "ranks_ok": int(tot[ci, 3].item()),
                                                      "frames_per_sec_total": round(float(tot[ci, 0]), 1),
                                                      "min_fps_per_camera": round(-float(neg[ci]), 2),
                                                      "h2d_GBps_total": round(float(tot[ci, 2]), 1)}
        if err:
            streams_multi["error_rank0"] = err
    sharded = None
    if (world > 1 or os.environ.get("CLEARCAM_BENCH_FORCE_SHARDED")) and not args.no_clip:
        try:                                     # side metric: must never take the headline line down with it
            if world == 1 and not dist.is_initialized():
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
                if on_gpu:
                    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
                else:
                    dist.init_process_group(backend, rank=0, world_size=1)
            sharded = sharded_search_metrics(local, dev, world, rank)
        except Exception as exc:                 # noqa: BLE001
            sharded = {"error": f"{type(exc).__name__}: {exc}"}
        # every rank reaches this point whatever happened above; agree on success so that no rank waits in a collective alone
        if world > 1:
            ok = torch.tensor([0.0 if "error" in sharded else 1.0], dtype=torch.float64, device=dev)
            dist.all_reduce(ok, op=dist.ReduceOp.SUM)
            sharded["ranks_ok"] = int(ok.item())
    if rank == 0:
        fps = world * B * args.steps / elapsed
        import csv
        import tempfile
        tmp_csv = os.path.join(tempfile.gettempdir(), f"clearcam_per_launch_{os.getpid()}.csv")
        os.environ["CLEARCAM_PROFILE_CSV"] = tmp_csv      # cc_yolo_profile also writes its per-launch table there
        prof = model.profile(iters=3)
        os.environ.pop("CLEARCAM_PROFILE_CSV", None)
        heaviest = None
        mfma_per_mac = MFMA_PER_MAC.get(args.dtype, 1.0)
        stem_flops = 0.0
        per_launch_roof = None
        try:
            table = list(csv.DictReader(open(tmp_csv)))
            stem_flops = sum(2e9 * float(r["alg_gmac"]) for r in table if r["kind"] == "stem_fused")
            # per-launch roof: every launch of the plan against max(its algorithmic FLOPs / MFMA peak, its minimum bytes / 8 TB/s)
            ideal = sum(max(2e9 * float(r["alg_gmac"]) / (PEAK_TFLOPS[args.dtype] * 1e12), float(r["gbytes_min"]) * 1e9 / HBM_PEAK) for r in table) * 1e3
            spent = sum(float(r["ms"]) for r in table)
            per_launch_roof = {"ideal_ms": round(ideal, 3), "measured_ms": round(spent, 3), "frac": round(ideal / spent, 4),
                               "how": "sum over the plan's launches of max(algorithmic FLOPs / dense MFMA peak, minimum bytes / 8 TB/s) over the sum of their "
                                      "hipEvent-timed durations (eager replay, ~0.3 ms of event overhead per step inside the denominator)"}
            # two-line group summary: the launches whose roof is the matrix pipes, and those whose roof is HBM
            grp = {}
            for bound in ("mfma", "hbm"):
                rs = [r for r in table if r.get("bound") == bound and float(r["ms"]) > 0]
                ms_b = sum(float(r["ms"]) for r in rs)
                if rs and ms_b > 0:
                    grp[bound] = {"launches": len(rs), "ms": round(ms_b, 3), "roof_ms": round(sum(float(r["roof_ms"]) for r in rs), 3),
                                  "TFLOP/s": round(sum(2e9 * float(r["alg_gmac"]) for r in rs) / (ms_b * 1e-3) / 1e12, 1),
                                  "min_bytes_TB/s": round(sum(float(r["gbytes_min"]) for r in rs) * 1e9 / (ms_b * 1e-3) / 1e12, 3)}
            per_launch_roof["by_bound"] = grp
            # tile quantisation: share of every launch's tile slots (resident blocks x CUs per round) left empty in its last round, weighted by its time
            q = [(float(r["ms"]), int(r["tiles"]), int(r["slots"]), int(r["rounds"])) for r in table if r.get("tiles") and int(r["tiles"]) > 0 and int(r["rounds"]) > 0]
            if q:
                per_launch_roof["tile_quantisation"] = {"launches": len(q), "ms": round(sum(m for m, *_ in q), 3),
                                                        "idle_slot_ms": round(sum(m * (1.0 - t / (rd * sl)) for m, t, sl, rd in q), 3),
                                                        "how": "sum over launches of ms x (1 - tiles / (rounds x slots)); slots = resident blocks x CUs (the grid of a persistent "
                                                               "kernel); with three batches in flight other batches' launches fill these slots (value vs one_batch_in_flight)"}
            conv_rows = [r for r in table if r["kind"] in ("conv", "conv_avg", "csp_fused") and r.get("weight_planes")]
            if conv_rows:
                mfma_per_mac = sum(float(r["alg_gmac"]) * int(r["weight_planes"]) for r in conv_rows) / sum(float(r["alg_gmac"]) for r in conv_rows)
            rows = [r for r in table if r["kind"] == "conv"]
            top = max(rows, key=lambda r: float(r["ms"]))
            heaviest = {"layer": f"{top['ks']}x{top['ks']} s{top['stride']} {top['Cin']}->{top['Cout']}, {int(float(top['M']))} output pixels",
                        "ms": round(float(top["ms"]), 4), "TFLOP/s": float(top["tflops"]), "min_GB/s": float(top["gbs"]),
                        "frac_of_peak": round(float(top["tflops"]) / PEAK_TFLOPS[args.dtype], 4)}
            os.remove(tmp_csv)
        except Exception:                         # noqa: BLE001  the table is a convenience, never fatal
            pass
        alg_flops = 2.0 * prof["alg_macs_per_step"]
        parity = None
        if on_gpu and world == 1 and not args.no_parity:      # N = 1 only, like the CPU baseline: the other ranks would wait at the final barrier
            try:
                parity = measured_parity(local)
            except Exception as exc:             # noqa: BLE001  a side measurement must not take the headline down
                parity = {"error": f"{type(exc).__name__}: {exc}"}
        # Conv time per step as the launches run in production (inside the plan's hipGraph), with ONE hipEvent pair per measurement
        # instead of an event record between every two launches (2-3 us each, ~0.3 ms per step of measurement overhead in the
        # per-launch table above):  in-plan conv time = whole step replayed - every non-conv launch replayed.  The conv launches
        # replayed ALONE run ~7 % faster than inside the plan (warmer caches, no pools / stem in between), so that number is only
        # reported, not used; rocprofv3's kernel durations (profiles/) are the referee.
        g_all = g_other = g_conv = None
        try:
            g_all, g_other, g_conv = model.profile_graph(2, 10), model.profile_graph(1, 10), model.profile_graph(0, 10)
        except Exception:                         # noqa: BLE001  older library / mocked model
            pass
        default_cfg = (args.size, args.res, B, fh, fw) == ("c", 640, 64, 640, 640)
        device_name = torch.cuda.get_device_name(local) if on_gpu else "cpu"
        traffic, traffic_note = measured_traffic(default_cfg, args.dtype)
        traced = traced_kernel_ms(default_cfg, args.dtype, device_name)
        peak = PEAK_TFLOPS[args.dtype]
        # `achieved` / `frac` are measured IN THIS RUN: the conv launches' time inside the plan = whole step replayed as a hipGraph minus
        # every non-conv launch replayed as a hipGraph (one hipEvent pair around ten replays each, on the launch stream); the eager
        # per-launch event sum is the fallback.  The committed rocprofv3 trace of the same kernel sources / storage mode / device model
        # is the referee, reported beside it (frac_rocprofv3) and never substituted for the live number.
        if g_all and g_other:
            conv_s, frac_source = (g_all - g_other) * 1e-3, "live in this run: whole-step hipGraph minus non-conv-launch hipGraph, hipEvents on the launch stream"
        else:
            conv_s, frac_source = prof["conv_ms"] * 1e-3, "live in this run: sum of per-launch hipEvent pairs, eager replay (graph profile unavailable)"
        live_s = conv_s
        achieved = alg_flops / conv_s / 1e12
        line = {
            "metric": f"yolov9{args.size}_{args.res}x{args.res}_frames_per_sec" if (fh, fw) == (args.res, args.res)
                      else f"yolov9{args.size}_{fh}x{fw}_letterbox{args.res}_frames_per_sec",
            "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": {"f16h": "f16", "f16s": "f16", "f16c": "f16"}.get(args.dtype, args.dtype), "storage_mode": args.dtype, "data": "synthetic",
            "config": {"workload": f"YOLOv9-{args.size.upper()} {args.dtype} batch={B} {fh}x{fw} frames (letterbox {args.res}) per GPU, "
                                   f"uint8 BGR frames resident in HBM, seeded synthetic weights, full detect path "
                                   f"(letterbox+convs+decode+top300+NMS)",
                       "batch_per_gpu": B, "parallelism": f"one camera batch per GPU x{world}, no collective",
                       "batches_in_flight": depth,
                       "batches_in_flight_note": (f"the K timed steps go round robin to {depth} slots of one handle (cc_yolo_submit: own stream, tensor arena and "
                                                  "captured graph per slot), so the last layers of a batch run beside the first layers of the next; same kernels, "
                                                  "bit-identical detections; the K steps start and complete inside the timed region") if depth > 1 else None,
                       "detections_last_batch": n_det},
            "one_batch_in_flight": {"ms_per_step": round(elapsed_one / args.steps * 1e3, 3), "frames_per_sec": round(world * B * args.steps / elapsed_one, 2),
                                    "how": "the same K steps as back-to-back cc_yolo_detect calls on one stream (the reference's call order; rounds 1-2 measured this)"},
            "parity": parity,
            "frames_per_sec_by_storage_dtype": precisions,
            "ranks_seen": ranks_seen, "ms_per_step_by_rank": per_rank_ms, "storage_mode_by_rank": per_rank_mode,
            "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                         "frac": round(achieved / peak, 4), "frac_source": frac_source,
                         # split weights issue TWO matrix instructions per algorithmic multiply-add (W_hi and W_lo against the same activations):
                         # `achieved` / `frac` count the ALGORITHMIC FLOPs (what the reference computes), these two the matrix pipe's actual work
                         "mfma_issued_tflops": round(achieved * mfma_per_mac, 2),
                         "frac_mfma_issued": round(achieved * mfma_per_mac / peak, 4),
                         "mfma_per_algorithmic_mac": round(mfma_per_mac, 4),
                         "per_launch_roof": per_launch_roof,
                         # the bandwidth-side fraction, unfused activation traffic (380.6 MB/frame bf16 at 640x640) over 8 TB/s
                         "hbm_side_frac": round(fps / world * 380.6e6 / 8e12, 4) if (fh, fw, args.res, args.size) == (640, 640, 640, "c") else None,
                         "traffic": traffic, "traffic_note": traffic_note,
                         "kernel": "conv kernels (every conv launch of the plan incl. the fused RepNCSP launches; the fused letterbox + first conv is reported under other_ms_per_step)",
                         "kernel_timing_note": "kernel durations are taken with ONE batch in flight and every launch on one chain, so that no launch shares "
                                               "the chip with another; with batches in flight the same launches overlap and the step gets shorter than "
                                               "their sum (whole_step_tflops)",
                         "whole_step_tflops": round((alg_flops + stem_flops) * args.steps / elapsed / 1e12, 1),
                         "alg_gflop_per_step": round(alg_flops / 1e9, 2), "kernel_ms_per_step": round(conv_s * 1e3, 3),
                         "kernel_ms_per_step_how": "whole step replayed as a hipGraph minus the non-conv launches replayed as a hipGraph, one hipEvent pair around "
                                                   "10 replays each (the conv launches' time inside the plan, inter-kernel gaps included)"
                                                   if g_all and g_other else "sum of per-launch hipEvent pairs (eager replay)",
                         "graph_ms": {"whole_step": round(g_all, 3), "non_conv_launches": round(g_other, 3), "conv_launches_alone": round(g_conv, 3)} if g_all else None,
                         "kernel_ms_per_step_eager_events": round(prof["conv_ms"], 3),
                         # the referee: rocprofv3's own kernel durations for the same launches (under the profiler the chip clocks ~2 % lower)
                         "kernel_ms_per_step_rocprofv3": round(traced["conv_ms_per_step"], 3) if traced else None,
                         "frac_rocprofv3": round(alg_flops / (traced["conv_ms_per_step"] * 1e-3) / 1e12 / peak, 4) if traced else None,
                         "rocprofv3_record": ({k: traced.get(k) for k in ("tag", "dtype", "device_name", "kernel_source_digest")} if traced else
                                              "no committed trace of these kernel sources / storage mode / device (profiles/kernel_trace.json)"),
                         "launches_per_step": prof["conv_launches"], "heaviest_launch": heaviest,
                         "other_ms_per_step": {k: round(prof[k], 3) for k in ("pool_ms", "decode_ms", "nms_ms", "stem_ms")},
                         "note": "stem_ms = stem_fused_kernel (letterbox + the 3->64 first conv straight from the uint8 frames, a byte/VALU-bound "
                                 "kernel): its time and its 0.35 % of the FLOPs are outside achieved/frac" if prof.get("stem_ms") else None},
            "gflop_per_frame": round((alg_flops + stem_flops) / B / 1e9, 2),
        }
        if median_100:
            line["median_100_steps"] = median_100
        if latency is not None:
            line["config"]["latency_ms_per_batch"] = latency
        # BASELINE.json configs[0] shape of call: one frame per call (the reference's batch-1 semantics), device-resident frame,
        # call-to-result latency including the (300,6) read-back
        try:
            one, lat = frames[:1].contiguous(), []
            for _ in range(5):
                model.detect_batch(one)
            for _ in range(50):
                t1 = time.perf_counter()
                model.detect_batch(one)
                lat.append(time.perf_counter() - t1)
            lat.sort()
            line["single_frame"] = {"ms_p50": round(lat[len(lat) // 2] * 1e3, 3), "ms_p99": round(lat[-1] * 1e3, 3),
                                    "frames_per_sec": round(1.0 / lat[len(lat) // 2], 1), "gpu_ms": round(model.last_gpu_ms(), 3)}
        except Exception as exc:                 # noqa: BLE001  a side metric
            line["single_frame"] = {"error": f"{type(exc).__name__}: {exc}"}
        model.close()
        if not args.no_streams and world == 1:
            line["streams"] = stream_side_metrics(local, args.size, args.res, args.dtype)
        if not args.no_clip and on_gpu:
            line["clip"] = clip_side_metrics(local, dev)
            # the other metrics of BASELINE.json against their own roofs, inside `roofline` so every figure of the line has its fraction
            try:
                c = line["clip"]
                q1 = c["search"]["one_gpu_1M_f32"]
                line["roofline"]["other_metrics"] = {
                    "clip_vit_l14_image_encode": {"bound": "mfma", "achieved": c["image_tflops"], "peak": PEAK_TFLOPS["bf16"], "unit": "TFLOP/s",
                                                  "frac": c["image_frac_of_mfma_peak"], "what": f"{c['image_embeds_per_sec']} img/s at batch {c['image_batch']} x 162.03 GFLOP, bf16, whole tower"},
                    "search_scan_1M_f32": {"bound": "hbm", "achieved": q1["q1_end_to_end_GBps"], "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                                           "frac": round(q1["q1_end_to_end_GBps"] * 1e9 / HBM_PEAK, 4),
                                           "what": "1 M x 768 f32 rows (3.07 GB) / host-observed p50 latency of one query (upload + scan + top-k + read-back)"}}
            except Exception:                     # noqa: BLE001
                pass
        if streams_multi is not None:
            line["streams_multi_gpu"] = streams_multi
        if sharded is not None:
            line["search_sharded"] = sharded
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(args.size, args.res, with_clip=not args.no_clip)
        emit(line)
    if world > 1:
        dist.barrier()
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

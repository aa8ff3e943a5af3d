"""dev tool: A/B of the eight-wave kernel's forms in ONE process (cc_dev_set("phase_flags", ...)):
    base     32          one 256x256 tile per block, schedule 1, block-wide LDS-staged epilogue (conv_phase_kernel)
    persist  512         persistent tile loop, wave-private staged epilogue (conv_persist_kernel<T, 0>)
    Measured and removed (records under profiles/): + 2048 drain the previous tile's stores before the next K loop (no difference,
    r03d); + 1024 v_mfma_f32_32x32x16 (10-15 % slower, r03b; still built, opt-in); block start stagger (no difference, r03f); the
    last, partial round re-cut into shorter tiles (no gain: a 144-of-256-CU round already runs ~1.3x faster per tile, r03i)
    (p32     512 + 1024  persist on v_mfma_f32_32x32x16, conv_persist_kernel<T, 1>: measured 10-15 % slower, r03b)
1. single layers through cc_conv_bench (random data, device time per launch), 2. the YOLOv9-C B=64 detect step, 3. CLIP ViT-L/14.

    python tools/dev/persist_ab.py [layers|yolo|clip|all] [dtype]
"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from clearcam_amd import _lib  # noqa: E402

L = _lib.lib()
FORMS = [("base", 32), ("persist", 512)]
what = sys.argv[1] if len(sys.argv) > 1 else "all"
dtype = sys.argv[2] if len(sys.argv) > 2 else "f16"
DT = {"f16": 1, "bf16": 2}[dtype]

# (name, B, H, W, Cin, Cout, k, stride, act)
SHAPES = [
    ("gemm 65535x1024x1024 (CLIP out-proj)", 255, 257, 1, 1024, 1024, 1, 1, 0),
    ("gemm 65535x3072x1024 (CLIP qkv)", 255, 257, 1, 1024, 3072, 1, 1, 0),
    ("gemm 65535x4096x1024 (CLIP fc, gelu)", 255, 257, 1, 1024, 4096, 1, 1, 2),
    ("gemm 65535x1024x4096 (CLIP proj)", 255, 257, 1, 4096, 1024, 1, 1, 0),
    ("3x3 256->256 @80x80 B64 (head)", 64, 80, 80, 256, 256, 3, 1, 1),
    ("3x3 256->256 @40x40 B64", 64, 40, 40, 256, 256, 3, 1, 1),
    ("3x3 512->256 @40x40 B64", 64, 40, 40, 512, 256, 3, 1, 1),
    ("1x1 1024->512 @40x40 B64", 64, 40, 40, 1024, 512, 1, 1, 1),
    ("1x1 1024->256 @80x80 B64", 64, 80, 80, 1024, 256, 1, 1, 1),
    ("1x1 512->512 @80x80 B64", 64, 80, 80, 512, 512, 1, 1, 1),
    ("1x1 256->256 @160x160 B64", 64, 160, 160, 256, 256, 1, 1, 1),
    ("1x1 256->256 @80x80 B64", 64, 80, 80, 256, 256, 1, 1, 1),
    ("3x3 s2 256->256 @80->40 B64", 64, 80, 80, 256, 256, 3, 2, 1),
    ("1x1 768->512 @40x40 B64", 64, 40, 40, 768, 512, 1, 1, 1),
    ("1x1 512->512 @40x40 B64", 64, 40, 40, 512, 512, 1, 1, 1),
    ("1x1 512->256 @80x80 B64", 64, 80, 80, 512, 256, 1, 1, 1),
    ("1x1 256->256 @40x40 B64 (400 tiles)", 64, 40, 40, 256, 256, 1, 1, 1),
    ("1x1 512->256 @40x40 B64 (400 tiles)", 64, 40, 40, 512, 256, 1, 1, 1),
]


def set_flags(v):
    _lib.check(L.cc_dev_set(b"phase_flags", v))


if what in ("layers", "all"):
    for rnd in range(2):
        for name, B, H, W, Cin, Cout, k, stride, act in SHAPES:
            Ho, Wo = (H + 2 * (k // 2) - k) // stride + 1, (W + 2 * (k // 2) - k) // stride + 1
            gf = 2.0 * B * Ho * Wo * Cout * Cin * k * k / 1e9
            row = []
            for form, flags in FORMS:
                set_flags(flags)
                ms = C.c_float()
                rc = L.cc_conv_bench(DT, B, H, W, Cin, Cout, k, stride, act, 7, 20, C.byref(ms))
                row.append(f"{form}: {ms.value * 1e3:7.1f} us {gf / ms.value:6.0f} TF" if rc == 0 else f"{form}: error {L.cc_last_error().decode()[:60]}")
            print(f"[{rnd}] {dtype} {name:40} {gf:8.1f} GF  " + "   ".join(row), flush=True)
    set_flags(-1)

if what in ("yolo", "all"):
    import torch
    from clearcam_amd.weights import synthetic_yolov9_state_dict
    from clearcam_amd.yolov9 import YOLOv9
    sd = synthetic_yolov9_state_dict("c", 1234)
    B = 64
    f = torch.from_numpy(np.random.default_rng(1).integers(0, 256, (B, 640, 640, 3), dtype=np.uint8)).cuda()
    o = torch.empty(B, 300, 6, device="cuda")
    models, outs = [], {}
    for form, flags in FORMS:
        set_flags(flags)
        m = YOLOv9("c", 640, state_dict=sd, dtype=dtype)
        for _ in range(3):
            m.detect_batch_device(f, o)
        torch.cuda.synchronize()
        outs[form] = o.clone()
        models.append((form, m))
    set_flags(-1)
    times = {form: [] for form, _ in models}
    for _ in range(7):
        for form, m in models:
            torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(10):
                m.detect_batch_device(f, o)
            torch.cuda.synchronize()
            times[form].append((time.perf_counter() - t) / 10 * 1e3)
    for form, m in models:
        ts = sorted(times[form])
        prof = m.profile(iters=3)
        same = bool(torch.equal(outs[form], outs["base"]))
        nd = int((outs[form][..., 4] > 0).sum())
        print(f"yolo {dtype} B=64 {form:8} median {ts[len(ts) // 2]:.3f} ms/step  min {ts[0]:.3f}  conv {prof['conv_ms']:.3f} ms (eager events)  "
              f"detections {nd}  bit-identical to base: {same}", flush=True)
        m.close()

if what in ("clip", "all"):
    import torch
    from clearcam_amd.arch import CLIP_L14
    from clearcam_amd.objects import OpenCLIP
    from clearcam_amd.weights import synthetic_clip_state_dict
    dev = torch.device("cuda", 0)
    sd = synthetic_clip_state_dict(CLIP_L14, 4321)
    x = torch.rand(255, 3, 224, 224, device=dev) * 2 - 1
    embs, models = {}, []
    for form, flags in FORMS:
        set_flags(flags)
        m = OpenCLIP(state_dict=sd, arch=CLIP_L14, dtype="bf16", device=0)
        emb = torch.empty(255, 768, device=dev)
        for _ in range(2):
            m.precompute_embedding_device(x, emb)
        torch.cuda.synchronize()
        embs[form] = emb.clone()
        models.append((form, m))
    set_flags(-1)
    rates = {form: [] for form, _ in models}
    emb = torch.empty(255, 768, device=dev)
    for _ in range(5):
        for form, m in models:
            torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(3):
                m.precompute_embedding_device(x, emb)
            torch.cuda.synchronize()
            rates[form].append(255 / ((time.perf_counter() - t) / 3))
    for form, m in models:
        r = sorted(rates[form])[len(rates[form]) // 2]
        cos = float((embs[form] * embs["base"]).sum(1).min())
        print(f"clip L/14 bf16 B=255 {form:8} {r:8.1f} img/s = {r * 162.03e9 / 1e12:6.1f} TF   min cos vs base {cos:.7f}", flush=True)
        m.close()

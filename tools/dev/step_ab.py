"""dev tool: interleaved A/B of the YOLOv9-C detect step under per-plan environment switches.

    python tools/dev/step_ab.py BATCH [DTYPE] NAME:ENV=V[,ENV=V...] NAME:...
One model per configuration (the switches are applied while its plan is built), rounds of 10 steps interleaved between the
configurations, median ms/step per configuration; the per-launch table of each goes to gpurun_out/ab_<NAME>_b<BATCH>.csv.
Only switches that are read per plan work here (CLEARCAM_FUSE_CSP, CLEARCAM_FUSE_STEM is cached: use separate processes).
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from clearcam_amd.weights import synthetic_yolov9_state_dict  # noqa: E402
from clearcam_amd.yolov9 import YOLOv9  # noqa: E402


def main():
    B = int(sys.argv[1])
    rest = sys.argv[2:]
    dtype = "bf16"
    if rest and ":" not in rest[0]:
        dtype, rest = rest[0], rest[1:]
    cfgs = []
    for spec in rest:
        name, _, envs = spec.partition(":")
        cfgs.append((name, dict(kv.split("=", 1) for kv in envs.split(",") if kv)))
    sd = synthetic_yolov9_state_dict("c", 1234)
    f = torch.from_numpy(np.random.default_rng(1).integers(0, 256, (B, 640, 640, 3), dtype=np.uint8)).cuda()
    o = torch.empty(B, 300, 6, device="cuda")
    models = []
    for name, env in cfgs:
        for k, v in env.items():
            os.environ[k] = v
        m = YOLOv9("c", 640, state_dict=sd, dtype=dtype)
        for _ in range(3):
            m.detect_batch_device(f, o)
        torch.cuda.synchronize()
        for k in env:
            os.environ.pop(k, None)
        models.append((name, m))
    times = {name: [] for name, _ in models}
    steps = 10 if B >= 16 else 50
    for _ in range(7):
        for name, m in models:
            torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(steps):
                m.detect_batch_device(f, o)
            torch.cuda.synchronize()
            times[name].append((time.perf_counter() - t) / steps * 1e3)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    for name, m in models:
        m.detect_batch_device(f, o)
        os.environ["CLEARCAM_PROFILE_CSV"] = os.path.join(ROOT, "gpurun_out", f"ab_{name}_b{B}.csv")
        prof = m.profile(iters=5)
        os.environ.pop("CLEARCAM_PROFILE_CSV", None)
        ts = sorted(times[name])
        print(f"{name:>12} B={B} {dtype}: median {ts[len(ts) // 2]:.3f} ms/step (min {ts[0]:.3f}), {B / ts[len(ts) // 2] * 1e3:.0f} frames/s; "
              f"conv {prof['conv_ms']:.3f} ms in {prof['conv_launches']} launches, pool {prof['pool_ms']:.3f}, stem {prof['stem_ms']:.3f}", flush=True)


if __name__ == "__main__":
    main()

"""Development aid for csp_fused.hip: fused RepNCSP against the four launches it replaces, stage by stage.

Runs each configuration in its own process (the switches are read when a plan is built):
    unfused  CLEARCAM_FUSE_CSP=0  taps csp{k}_ab ([u | b] after the block), csp{k}_t, csp{k}_u
    fused    CLEARCAM_FUSE_CSP=1  tap  csp{k}_u;  CLEARCAM_CSP_DBG=1..3 makes the kernel write an intermediate there instead
and prints, per block and stage, how many values differ and where (channel group, position in the 8x16 tile, image border).
    python tools/dev/csp_debug.py [size] [res] [dtype] [H] [W] [B]
"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SCRIPT = r"""
import sys, numpy as np
from clearcam_amd.weights import conditioned_yolov9_state_dict, synthetic_yolov9_state_dict
from clearcam_amd.yolov9 import YOLOv9
size, res, dtype, H, W, B, out = sys.argv[1], int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]), sys.argv[7]
frames = np.random.default_rng(5).integers(0, 256, (B, H, W, 3), dtype=np.uint8)
sd = conditioned_yolov9_state_dict(size, 1234) if size == "c" else synthetic_yolov9_state_dict(size, 1234)
m = YOLOv9(size, res, state_dict=sd, dtype=dtype, device=0)
det = m.detect_batch(frames)
d = {"det": det}
for k in range(16):
    for suf in ("_ab", "_t", "_u"):
        try: d[f"csp{k}{suf}"] = m.get_tensor(f"csp{k}{suf}")
        except Exception: pass
for n in ("p3", "p4", "p5"): d[n] = m.get_tensor(n)
np.savez(out, **d)
"""


def run(tag, env_extra, args, tmp):
    path = os.path.join(tmp, f"csp_{tag}.npz")
    env = dict(os.environ, CLEARCAM_TAP_CSP="1", PYTHONPATH=ROOT, **env_extra)
    subprocess.run([sys.executable, "-c", SCRIPT] + args + [path], check=True, env=env)
    return np.load(path)


def report(name, ref, got):
    if ref.shape != got.shape:
        print(f"  {name}: SHAPE {ref.shape} vs {got.shape}"); return False
    bad = ref != got
    if not bad.any():
        print(f"  {name}: identical ({ref.size} values, max |x| {np.abs(ref).max():.3g})"); return True
    B, H, W, C = ref.shape
    d = np.abs(ref.astype(np.float64) - got)
    print(f"  {name}: {bad.mean() * 100:.3f} % differ, max |d| {d.max():.4g} (max |ref| {np.abs(ref).max():.3g}), nan {np.isnan(got).sum()}")
    g = C // 4 if C >= 4 else 1
    print("    by channel quarter:", [f"{bad[..., i * g:(i + 1) * g].mean() * 100:.2f}" for i in range(C // g)])
    oy = (np.arange(H) % 8)[None, :, None, None]; ox = (np.arange(W) % 16)[None, None, :, None]
    print("    by tile row      :", [f"{bad[np.broadcast_to(oy == r, bad.shape)].mean() * 100:.2f}" for r in range(8)])
    print("    by tile column   :", [f"{bad[np.broadcast_to(ox == c, bad.shape)].mean() * 100:.2f}" for c in range(16)])
    border = np.zeros((H, W), bool); border[:2] = border[-2:] = True; border[:, :2] = border[:, -2:] = True
    print(f"    image border (2 px): {bad[:, border].mean() * 100:.2f} %   interior: {bad[:, ~border].mean() * 100:.2f} %")
    return False


def main():
    size = sys.argv[1] if len(sys.argv) > 1 else "c"
    res = sys.argv[2] if len(sys.argv) > 2 else "640"
    dtype = sys.argv[3] if len(sys.argv) > 3 else "bf16"
    H = sys.argv[4] if len(sys.argv) > 4 else res
    W = sys.argv[5] if len(sys.argv) > 5 else res
    B = sys.argv[6] if len(sys.argv) > 6 else "2"
    args = [size, res, dtype, H, W, B]
    tmp = os.environ.get("TMPDIR", "/tmp")
    un = run("unfused", {"CLEARCAM_FUSE_CSP": "0"}, args, tmp)
    level = os.environ.get("CSP_FUSE_LEVEL", "2")                      # 2: hidden width 32 and 64, 1: the default selection
    fu = run("fused", {"CLEARCAM_FUSE_CSP": level}, args, tmp)
    ok = True
    fused_blocks = [k for k in range(16) if f"csp{k}_u" in fu.files and f"csp{k}_ab" not in fu.files]
    print("fused blocks:", fused_blocks)
    seen = set()
    for k in fused_blocks:
        ab, t, u = un[f"csp{k}_ab"], un[f"csp{k}_t"], un[f"csp{k}_u"]
        hid = t.shape[-1]
        print(f"block csp{k}: {u.shape}, hidden {hid}")
        if hid not in seen or os.environ.get("CSP_DEBUG_ALL"):
            # with ONLY this block fused its input is the unfused network's, so every stage can be compared exactly
            seen.add(hid)
            only = {"CLEARCAM_DEV": "1", "CLEARCAM_FUSE_CSP": level, "CLEARCAM_CSP_ONLY": str(k)}
            st = {s: run(f"b{k}dbg{s}", dict(only, CLEARCAM_CSP_DBG=str(s)), args, tmp) for s in (1, 2, 3)}
            st[4] = run(f"b{k}full", only, args, tmp)
            ok &= report("stage 1 b   ", ab[..., hid:], st[1][f"csp{k}_u"][..., hid:])
            ok &= report("stage 2 t   ", t, st[2][f"csp{k}_u"][..., :hid])
            ok &= report("stage 3 u|b ", ab, st[3][f"csp{k}_u"])
            ok &= report("stage 4 out ", u, st[4][f"csp{k}_u"])
        ok &= report("all fused   ", u, fu[f"csp{k}_u"])
    for n in ("p3", "p4", "p5"):
        ok &= report(n, un[n], fu[n])
    print("detections equal:", np.array_equal(un["det"], fu["det"]))
    print("ALL IDENTICAL" if ok else "DIFFERENCES FOUND")


if __name__ == "__main__":
    main()

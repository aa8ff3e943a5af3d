# dev tool (GPU; --selftest runs on the CPU): the round-5 lead of DESIGN.md section 8 on the real kernels, without touching the library.
#   1. an f32 handle with the development taps (CLEARCAM_TAP_BLOCKS / CLEARCAM_TAP_CSP, RepNCSP unfused) runs a few calibration frames; the
#      inputs of the 1x1 convs are read back through cc_yolo_get_tensor (conv_inputs() below maps parameter names to taps)
#   2. every such conv's weights are rounded to f16 column by column with the error fed forward through the inverse of H = E[x x^T] (GPTQ)
#   3. the pre-rounded float32 state dict (the library's controlled rounding leaves exactly representable weights alone) is loaded with
#      dtype "f16h" and CLEARCAM_SPLIT_1X1_LAST=-1 (low plane in the stem conv only) and compared with the f32 oracle on test frames, next to
#      "f16h" as shipped and plain "f16"; then the step time of each.
# argv: [test frames] [checkpoint seed];  env CALIB = noise | smooth (calibration frames).   --selftest: the name -> tap mapping against the
# CPU oracle's own conv inputs (no GPU needed).
import os, sys, time, numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from clearcam_amd.arch import YOLO_ARCH

ELANS = (2, 4, 6, 8, 12, 15, 18, 21)                    # RepNCSPELAN4 blocks of the t/s/m/c graph in build order: cat0 .. cat7, csp0 .. csp15
DOWNS = (3, 5, 7, 16, 19)
up = lambda x: x.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)   # noqa: E731


def conv_inputs(get, arch, sd):
    """{1x1 conv parameter prefix: its input activations (N,C,H,W) f32} for YOLOv9-<arch.size> from the development taps `get(name)`."""
    M = "model.list."
    b = lambda k: get({15: "p3", 18: "p4", 21: "p5"}.get(k, f"b{k}"))      # noqa: E731
    block_in = {2: lambda: b(1), 4: lambda: b(3), 6: lambda: b(5), 8: lambda: b(7), 12: lambda: torch.cat((up(b(9)), b(6)), 1),
                15: lambda: torch.cat((up(b(12)), b(4)), 1), 18: lambda: torch.cat((b(16), b(12)), 1), 21: lambda: torch.cat((b(19), b(9)), 1)}
    out = {}
    for n, blk in enumerate(ELANS):
        p = f"{M}{blk}"
        cat = get(f"cat{n}"); h2 = cat.shape[1] // 4                        # [y0 | y1 | y2 | y3], each 2 * hid channels
        out[p + ".cv1.conv"] = block_in[blk]()
        out[p + ".cv4.conv"] = cat
        for j, br in enumerate(("cv2", "cv3")):
            y = cat[:, (1 + j) * h2:(2 + j) * h2]                           # the branch's input: y1 for cv2, y2 for cv3
            r = f"{p}.{br}.list.0"
            out[r + ".cv1.conv"] = y; out[r + ".cv2.conv"] = y
            out[r + ".cv3.conv"] = get(f"csp{2 * n + j}_ab")               # [a + bottleneck(a) | b]
    for blk in DOWNS:                                                       # ADown (:40-52): cv2 = 1x1 over max_pool(avg_pool(x)'s second half)
        x = F.avg_pool2d(b(blk - 1), 2, 1, 0, False, True)
        out[f"{M}{blk}.cv2.conv"] = F.max_pool2d(x[:, x.shape[1] // 2:], 3, 2, 1)
    x8 = b(8)                                                               # SPPELAN (:127-149)
    out[M + "9.cv1.conv"] = x8
    y = [F.silu(F.conv2d(x8, torch.from_numpy(sd[M + "9.cv1.conv.weight"]), torch.from_numpy(sd[M + "9.cv1.conv.bias"])))]
    for _ in range(3):
        y.append(F.max_pool2d(y[-1], 5, 1, 2))
    out[M + "9.cv5.conv"] = torch.cat(y, 1)
    return out


def gptq_f16(w, H, damp=0.01):
    """w (co, ci) f32, H (ci, ci) f64 -> f16-representable f32 weights; the GPTQ recursion in input-channel order."""
    Wm = w.double().clone(); ci = Wm.shape[1]
    Hd = H.clone(); Hd += torch.eye(ci, dtype=torch.float64) * damp * Hd.diag().mean()
    Hinv = torch.linalg.cholesky(torch.cholesky_inverse(torch.linalg.cholesky(Hd)), upper=True)
    Q = torch.empty_like(Wm)
    for i in range(ci):
        q = Wm[:, i].float().to(torch.float16).double()
        Q[:, i] = q
        if i + 1 < ci:
            Wm[:, i + 1:] -= ((Wm[:, i] - q) / Hinv[i, i])[:, None] * Hinv[i, i + 1:][None, :]
    return Q.float()


def gptq_state_dict(sd, inputs, rows=200000):
    out = dict(sd); g = torch.Generator().manual_seed(0)
    for name, x in inputs.items():
        w = torch.from_numpy(sd[name + ".weight"])
        assert w.shape[2] == 1 and w.shape[1] == x.shape[1], (name, tuple(w.shape), tuple(x.shape))
        X = x.permute(0, 2, 3, 1).reshape(-1, x.shape[1]).double()
        if X.shape[0] > rows: X = X[torch.randperm(X.shape[0], generator=g)[:rows]]
        out[name + ".weight"] = gptq_f16(w[:, :, 0, 0], X.T @ X / X.shape[0]).reshape(w.shape).numpy()
    return out


def calibration_frames(kind, n=4):
    fr = np.random.default_rng(4242).integers(0, 256, (n, 640, 640, 3), dtype=np.uint8)
    if kind == "smooth":                                                    # another input distribution: heavily blurred noise, contrast stretched
        t = torch.from_numpy(fr).float().permute(0, 3, 1, 2)
        for _ in range(3): t = F.avg_pool2d(F.pad(t, (8, 8, 8, 8), mode="reflect"), 17, 1)
        t = (t - t.mean((2, 3), keepdim=True)) / t.std((2, 3), keepdim=True) * 50 + 128
        fr = t.clamp(0, 255).permute(0, 2, 3, 1).to(torch.uint8).numpy().copy()
    return fr


def selftest():
    """conv_inputs() against the CPU oracle: taps built from the oracle's block outputs and from the inputs of OTHER convs; every mapped
    tensor must equal the input the oracle's conv of that name really saw."""
    import oracle.yolov9_oracle as yo
    from clearcam_amd.weights import conditioned_yolov9_state_dict
    sd = conditioned_yolov9_state_dict("c", 1234, exact=False)
    o = yo.YOLOv9Oracle("c", 320, sd); seen = {}
    orig = o._conv2d
    def hooked(x, name, stride=1, groups=1):
        seen.setdefault(name, x.clone()); return orig(x, name, stride, groups)
    o._conv2d = hooked
    fr = np.random.default_rng(3).integers(0, 256, (1, 320, 320, 3), dtype=np.uint8)
    with torch.no_grad():
        o.decode(o.head_raw(o.features(o.network_input(fr))))
    taps = {}
    for k, v in o.block_outputs.items():
        taps[{15: "p3", 18: "p4", 21: "p5"}.get(k, f"b{k}")] = v
    taps["b16"] = seen["model.list.18.cv1.conv"][:, :256]; taps["b19"] = seen["model.list.21.cv1.conv"][:, :512]   # ADown outputs = first part of the next concat
    for n, blk in enumerate(ELANS):
        taps[f"cat{n}"] = seen[f"model.list.{blk}.cv4.conv"]
        for j, br in enumerate(("cv2", "cv3")):
            taps[f"csp{2 * n + j}_ab"] = seen[f"model.list.{blk}.{br}.list.0.cv3.conv"]
    got = conv_inputs(lambda n: taps[n], YOLO_ARCH["c"], sd)
    n1 = [k[:-len(".weight")] for k, v in sd.items() if k.endswith(".weight") and v.ndim == 4 and v.shape[2] == 1 and "dfl" not in k]
    missing = sorted(set(n1) - set(got))
    for name, x in got.items():
        assert name in seen, name
        assert x.shape == seen[name].shape and float((x - seen[name]).abs().max()) <= 1e-6, (name, tuple(x.shape), tuple(seen[name].shape))
    print(f"selftest ok: {len(got)} of {len(n1)} 1x1 convs mapped; left to controlled rounding: {missing}")
    q = gptq_state_dict(sd, {k: got[k] for k in list(got)[:3]})
    k = list(got)[0] + ".weight"
    t = torch.from_numpy(q[k]); assert torch.equal(t.to(torch.float16).float(), t) and not np.array_equal(q[k], sd[k])
    print("gptq ok: weights f16-exact, within", float((t - torch.from_numpy(sd[k])).abs().max() / torch.from_numpy(sd[k]).abs().max()), "of max|w|")


def main():
    from clearcam_amd.weights import conditioned_yolov9_state_dict, synthetic_yolov9_state_dict
    from clearcam_amd.yolov9 import YOLOv9
    from oracle.yolov9_oracle import YOLOv9Oracle, parity_summary, decoded_rows, tolerance_bars
    nf = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1234
    sd = conditioned_yolov9_state_dict("c", seed, exact=False)
    os.environ.update(CLEARCAM_TAP_BLOCKS="1", CLEARCAM_TAP_CSP="1", CLEARCAM_FUSE_CSP="0")
    m = YOLOv9("c", 640, state_dict=sd, dtype="f32")
    m.detect_batch(calibration_frames(os.environ.get("CALIB", "noise")))
    get = lambda n: torch.from_numpy(m.get_tensor(n)).permute(0, 3, 1, 2).contiguous()    # noqa: E731
    inputs = conv_inputs(get, YOLO_ARCH["c"], sd); m.close()
    for k in ("CLEARCAM_TAP_BLOCKS", "CLEARCAM_TAP_CSP", "CLEARCAM_FUSE_CSP"): os.environ.pop(k)
    t0 = time.time(); sdq = gptq_state_dict(sd, inputs); print(f"GPTQ over {len(inputs)} 1x1 convs: {time.time() - t0:.1f} s", flush=True)
    fr = np.random.default_rng(seed + 1).integers(0, 256, (nf, 640, 640, 3), dtype=np.uint8)
    o = YOLOv9Oracle("c", 640, sd); det, dec = [], []
    with torch.no_grad():
        for i in range(0, nf, 4):
            y = o.decode(o.head_raw(o.features(o.network_input(fr[i:i + 4]))))
            dec.append(decoded_rows(y)); det.append(o.scale_boxes((640, 640), o.postprocess(y), (640, 640)).numpy())
    ref, dec_ref = np.concatenate(det), np.concatenate(dec)
    modes = (("f16, controlled rounding", sd, "f16", None), ("f16h as shipped", sd, "f16h", None), ("GPTQ 1x1 + low plane in the stem only", sdq, "f16h", "-1"),
             ("GPTQ 1x1, no low plane", sdq, "f16", None), ("f16s", sd, "f16s", None))
    def make(s, dt, k1):
        if k1 is None: os.environ.pop("CLEARCAM_SPLIT_1X1_LAST", None)
        else: os.environ["CLEARCAM_SPLIT_1X1_LAST"] = k1
        return YOLOv9("c", 640, state_dict=s, dtype=dt)
    for label, s, dt, k1 in modes:
        mm = make(s, dt, k1)
        got = mm.detect_batch(fr); d = mm.get_tensor("decoded"); mm.close()
        p = parity_summary(ref, got, 0.64, dec_ref, d)
        print(f"checkpoint {seed} {label:40s} bars {'ok' if tolerance_bars(p)['all'] else 'NO'}", {k: round(p[k], 4) for k in ("match_frac", "match_frac_clear_of_threshold", "anchor_box_err_px_p99", "anchor_box_err_px_p999", "anchor_box_err_px_max", "anchor_score_err_max")}, flush=True)
    B = 64
    f = torch.from_numpy(np.random.default_rng(1).integers(0, 256, (B, 640, 640, 3), dtype=np.uint8)).cuda()
    for label, s, dt, k1 in modes:
        mm = make(s, dt, k1); mm.set_in_flight(3)
        outs = [torch.empty(B, 300, 6, device="cuda") for _ in range(3)]
        for i in range(9): mm.wait(mm.submit(f, outs[i % 3]))
        torch.cuda.synchronize(); t = time.perf_counter()
        tk = [mm.submit(f, outs[i % 3]) for i in range(30)]
        for k in tk[-3:]: mm.wait(k)
        torch.cuda.synchronize(); dt_s = (time.perf_counter() - t) / 30
        print(f"{label:40s} 3 in flight {dt_s*1e3:.3f} ms ({B/dt_s:.0f} frames/s)", flush=True); mm.close()


if __name__ == "__main__":
    selftest() if "--selftest" in sys.argv else main()

"""One-off tool: calibration tables for the two seeded synthetic YOLOv9 checkpoints.

    python tools/calibrate_synth.py                 # the chaotic checkpoint's LSUV scales -> assets/synth_scales.json
    python tools/calibrate_synth.py cond c          # the well-conditioned checkpoint      -> assets/synth_cond_<size>.npz
    python tools/calibrate_synth.py cond c --seed 7 # ... from another seed's base filters -> assets/synth_cond_<size>_s<seed>.npz
                                                    #   + measured conditioning            -> assets/synth_cond_report.json

A 144-conv SiLU stack without normalisation either explodes or dies under any fixed init gain, so
``clearcam_amd.weights.synthetic_yolov9_state_dict`` multiplies each seeded N(0,1/fan_in) weight by a committed
per-conv scale.  This script derives those scales by running the CPU oracle on a seeded noise batch and normalising
every conv's pre-activation std in execution order (LSUV).

``cond`` does the data-dependent initialisation of ``clearcam_amd.weights.conditioned_yolov9_state_dict`` (see the block
comment there): per conv one gain (spatial std of the centred pre-activation = COND_STD) and a per-channel bias that
centres it; per class / per DFL bin in the head.  It then MEASURES the result and writes the numbers next to the
table: the f32 perturbation gain (white noise of relative RMS 1e-3 injected at the network input and at three depths,
relative RMS growth read at P3/P4/P5) and what 16-bit storage rounding (oracle/lowprec_oracle.py) does to features
and detections.  The generators themselves need neither torch nor the oracle.
"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clearcam_amd import weights as W  # noqa: E402
from oracle.yolov9_oracle import YOLOv9Oracle, match_detections  # noqa: E402

ASSETS = os.path.join(os.path.dirname(W.__file__), "assets")


def calibrate(size: str, seed: int = 1234, res: int = 640):
    sd = W.synthetic_yolov9_state_dict(size, seed, scales={})
    o = YOLOv9Oracle(size, res, sd)
    scales = {}
    orig = o._conv2d

    def hooked(x, name, stride=1, groups=1):
        y = orig(x, name, stride, groups)
        if name not in scales:
            target = 1.0
            if ".m.list." in name and name.endswith("cv2.conv"):
                target = 0.6                      # residual branch
            is_cls = (".cv3.list" in name) and name.endswith(".list.2") and ("model.list.22." in name or "model.list.42." in name)
            if "cv2.list" in name and name.endswith(".list.2") and ("model.list.22." in name or "model.list.42." in name):
                target = 2.0                      # DFL logits
            b = o.sd[name + ".bias"]
            z = y - b.view(1, -1, 1, 1)
            s = float(target / z.std())
            if is_cls:
                # activations of a random unnormalised net are heavy-tailed: place the 1-3e-4
                # quantile of the class logits at the 0.25-score line (logit -1.1 given bias -5)
                s = float(3.9 / torch.quantile(z.flatten()[::7], 1.0 - 3e-4))
            scales[name] = s
            o.sd[name + ".weight"] *= s
            y = orig(x, name, stride, groups)
        return y

    o._conv2d = hooked
    fr = np.random.default_rng(7).integers(0, 256, (4, res, res, 3), dtype=np.uint8)
    with torch.no_grad():
        o.head_raw(o.features(o.network_input(fr)))
    return scales


# ---- the well-conditioned checkpoint ---------------------------------------------------------------------------------

def _frames(n: int, res: int, seed: int, natural: bool):
    if natural:
        from clearcam_amd.streams import natural_frames
        return natural_frames(n, res, res, seed=seed)
    return np.random.default_rng(seed).integers(0, 256, (n, res, res, 3), dtype=np.uint8)


def calibrate_conditioned(size: str, seed: int = 1234, res: int = 640, n_frames: int = 8, eps=None, std=None, natural: bool = False):
    """-> table {"g:<conv>": gain (scalar, or per output channel in the head), "b:<conv>": bias shift per channel, "j:<conv>": jitter}.
    eps / std: the stress variants' share of white filter and pre-activation std (clearcam_amd.weights.COND_STRESS)."""
    o = YOLOv9Oracle(size, res, W.conditioned_base_weights(size, seed, eps))      # unit gains, the seeded N(0,1) bias draws
    std = W.COND_STD if std is None else std
    base_bias = {k: v.clone() for k, v in o.sd.items() if k.endswith(".bias")}
    table = {}
    orig = o._conv2d
    thr_logit = float(np.log(0.25 / 0.75))                       # score 0.25

    def hooked(x, name, stride=1, groups=1):
        if "g:" + name in table:
            return orig(x, name, stride, groups)
        o.sd[name + ".bias"] = torch.zeros_like(base_bias[name + ".bias"])
        z = orig(x, name, stride, groups)                        # unit gain, zero bias
        m, s = z.mean((0, 2, 3)), z.std((0, 2, 3))
        head_out = name.endswith(".list.2") and ("model.list.22." in name or "model.list.42." in name)
        if head_out:
            if ".cv3." in name:                                  # class logits: per-class mean COND_CLS_BIAS, the COND_Q quantile at the threshold
                K = W.COND_ACTIVE_CLASSES                        # classes beyond the first K never fire (gain 0, bias -10)
                zn = ((z - m.view(1, -1, 1, 1)) / s.view(1, -1, 1, 1))[:, :K]
                g = (thr_logit - W.COND_CLS_BIAS) / float(torch.quantile(zn.flatten()[::7], 1.0 - W.COND_Q))
                on = (torch.arange(len(s)) < K).float()
                gain, jitter, shift = on * g / s, 0.0, on * (-(m * g / s) + W.COND_CLS_BIAS) - 10.0 * (1 - on)
            else:                                                # DFL logits: std COND_DFL_STD around a ramp over the 16 bins
                gain, jitter = W.COND_DFL_STD / s, 0.0
                shift = -(m * gain) + W.COND_DFL_RAMP * (torch.arange(len(s)) % 16 - 7.5)
        else:
            target = std * (W.COND_RES_FRAC if (".m.list." in name and name.endswith("cv2.conv")) else 1.0)
            gain = torch.full_like(s, target / float(s.mean()))
            jitter = W.COND_BIAS_JITTER * target
            shift = -(m * gain)
        table["g:" + name] = gain.numpy().astype(np.float32) if head_out else np.float32(gain[0])
        table["j:" + name] = np.float32(jitter)
        table["b:" + name] = shift.numpy().astype(np.float32)
        w = torch.from_numpy(W._storage_exact((o.sd[name + ".weight"] * gain.view(-1, 1, 1, 1)).numpy()))
        o.sd[name + ".weight"] = w
        o.sd[name + ".bias"] = base_bias[name + ".bias"] * jitter + shift
        return orig(x, name, stride, groups)

    o._conv2d = hooked
    fr = _frames(n_frames, res, 7, natural)
    with torch.no_grad():
        o.head_raw(o.features(o.network_input(fr)))
    return table


def conditioning_report(size: str, sd, res: int = 640, n_frames: int = 12, natural: bool = False):
    """Measured properties of a checkpoint: f32 perturbation gains and the 16-bit storage emulation against the f32 oracle."""
    from oracle.lowprec_oracle import LowPrecOracle, rel_rms
    frames = _frames(n_frames, res, 1, natural)
    o = YOLOv9Oracle(size, res, sd)
    rep = {"frames": n_frames, "res": res}
    with torch.no_grad():
        x = o.network_input(frames[:1])
        feats = [f.clone() for f in o.features(x)]
        # white noise of relative RMS 1e-3 (relative to the tensor's spatial std) added to the INPUT of the named conv
        probes = {"network input": "model.list.0.conv"}
        if size != "e":
            probes.update({"block 4 input": "model.list.4.cv1.conv", "block 9 input": "model.list.9.cv1.conv", "block 15 cv4 input": "model.list.15.cv4.conv"})
        gains = {}
        for label, target in probes.items():
            orig = o._conv2d
            gen = torch.Generator().manual_seed(5)

            def hooked(xx, name, stride=1, groups=1, orig=orig, target=target, gen=gen):
                if name == target:
                    sp = (xx - xx.mean((0, 2, 3), keepdim=True)).pow(2).mean().sqrt()
                    xx = xx + torch.randn(xx.shape, generator=gen) * (1e-3 * float(sp))
                return orig(xx, name, stride, groups)
            o._conv2d = hooked
            fp = o.features(x)
            o._conv2d = orig
            gains[label] = [round(float((p - f).pow(2).mean().sqrt() / (f - f.mean((0, 2, 3), keepdim=True)).pow(2).mean().sqrt()) / 1e-3, 3)
                            for p, f in zip(fp, feats)]
        rep["f32_perturbation_gain_at_p3_p4_p5"] = gains
        ref = o.detect_batch(frames)
    rep["detections_per_frame"] = [int((r[:, 4] > 0).sum()) for r in ref]
    rep["classes_detected"] = int(len(np.unique(np.concatenate([r[r[:, 4] > 0][:, 5] for r in ref]))))
    for dt in ("bf16", "f16"):
        lo = LowPrecOracle(size, res, sd, dt)
        with torch.no_grad():
            lf = lo.features(lo.network_input(frames[:1]))
            got = lo.detect_batch(frames)
        n_ref = n_got = n_match = 0
        box_err = sc_err = 0.0
        for b in range(n_frames):
            a, c, m, be, se = match_detections(ref[b], got[b], 0.9)
            n_ref += a; n_got += c; n_match += m; box_err = max(box_err, be); sc_err = max(sc_err, se)
        rep[dt + "_storage_emulation"] = {"p3_p4_p5_rel_rms": [round(rel_rms(a, b), 5) for a, b in zip(lf, feats)],
                                          "matched_iou90": n_match, "n_ref": n_ref, "n_got": n_got,
                                          "match_frac": round(n_match / max(n_ref, n_got, 1), 4),
                                          "max_box_err_px": round(box_err, 3), "max_score_err": round(sc_err, 5)}
    return rep


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "cond":
        argv = sys.argv[2:]
        seed = 1234
        if "--seed" in argv:                                        # further checkpoints: assets/synth_cond_<size>_s<seed>.npz
            seed = int(argv[argv.index("--seed") + 1]); del argv[argv.index("--seed"):argv.index("--seed") + 2]
        stress = None
        if "--stress" in argv:                                      # stress variants: assets/synth_cond_<size>_<name>.npz (weights.COND_STRESS)
            stress = argv[argv.index("--stress") + 1]; del argv[argv.index("--stress"):argv.index("--stress") + 2]
        sizes = argv or ["c"]
        rpath = os.path.join(ASSETS, "synth_cond_report.json")
        report = json.load(open(rpath)) if os.path.exists(rpath) else {}
        for size in sizes:
            tag = f"{size}_{stress}" if stress else (size if seed == 1234 else f"{size}_s{seed}")
            eps, std = W.COND_STRESS[stress] if stress else (None, None)
            nat = stress in W.COND_NATURAL
            table = calibrate_conditioned(size, seed, eps=eps, std=std, natural=nat)
            np.savez_compressed(os.path.join(ASSETS, f"synth_cond_{tag}.npz"), **W.pack_cond_table(size, table))
            W._COND.pop(tag, None)
            report[tag] = conditioning_report(size, W.conditioned_yolov9_state_dict(size, seed, stress=stress), natural=nat)
            report[tag]["frames"] = "natural_frames (1/f + flat regions + rectangles)" if nat else report[tag]["frames"]
            if stress:
                report[tag]["stress"] = {"eps": eps, "preact_std": std}
            report[tag]["design"] = {"eps": W.COND_EPS, "preact_std": W.COND_STD, "dfl_std": W.COND_DFL_STD, "dfl_ramp": W.COND_DFL_RAMP, "cls_bias": W.COND_CLS_BIAS, "quantile": W.COND_Q, "active_classes": W.COND_ACTIVE_CLASSES}
            print(tag, json.dumps(report[tag], indent=1), flush=True)
        json.dump(report, open(rpath, "w"), indent=1, sort_keys=True)
        print("wrote", rpath)
    elif len(sys.argv) > 1 and sys.argv[1] == "report-chaotic":
        print(json.dumps(conditioning_report("c", W.synthetic_yolov9_state_dict("c", 1234), n_frames=2), indent=1))
    else:
        out = {s: calibrate(s) for s in "tsmce"}
        path = os.path.join(ASSETS, "synth_scales.json")
        with open(path, "w") as f:
            json.dump(out, f, indent=0, sort_keys=True)
        print("wrote", path, {k: len(v) for k, v in out.items()})

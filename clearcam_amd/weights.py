"""State-dict plumbing: key names, seeded synthetic weights, safetensors loading.

The reference downloads ``yolov9-{size}.safetensors`` and
``CLIP-ViT-L-14-laion2B-s32B-b82K.safetensors`` at construction
(``detection/yolov9.py:372``, ``models/objects.py:91``).  There is no network
here, so every test/bench uses a *seeded synthetic* state dict with exactly the
reference's key names and shapes; real files drop in
through :func:`load_safetensors`.

The generator uses ``numpy.random.Generator(PCG64(seed))`` only, so the same
weights come out in this container and on the GPU box.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import numpy as np

from .arch import CLIP_L14, YOLO_ARCH, ClipArch, YoloArch

# ----------------------------------------------------------------------------------------------
# YOLOv9 key/shape enumeration (detection/yolov9.py attribute tree)
# ----------------------------------------------------------------------------------------------

ConvSpec = Tuple[str, int, int, int, int, bool]  # (prefix, cin, cout, k, groups, is_bare_conv2d)


def _conv(prefix: str, cin: int, cout: int, k: int, g: int = 1) -> ConvSpec:
    return (prefix + ".conv", cin, cout, k, g, False)


def _repncsp(prefix: str, cin: int, hid: int, n: int) -> List[ConvSpec]:
    out = [_conv(f"{prefix}.cv1", cin, hid, 1), _conv(f"{prefix}.cv2", cin, hid, 1),
           _conv(f"{prefix}.cv3", 2 * hid, 2 * hid, 1)]
    for j in range(n):
        out += [_conv(f"{prefix}.m.list.{j}.cv1", hid, hid, 3), _conv(f"{prefix}.m.list.{j}.cv2", hid, hid, 3)]
    return out


def _elan4(prefix: str, cin: int, hid: int, cout: int, n: int) -> List[ConvSpec]:
    out = [_conv(f"{prefix}.cv1", cin, 4 * hid, 1)]
    for br in ("cv2", "cv3"):
        out += _repncsp(f"{prefix}.{br}.list.0", 2 * hid, hid, n)
        out += [_conv(f"{prefix}.{br}.list.1", 2 * hid, 2 * hid, 3)]
    out += [_conv(f"{prefix}.cv4", 8 * hid, cout, 1)]
    return out


def _down(prefix: str, kind: str, cin: int, cout: int) -> List[ConvSpec]:
    if kind == "adown":
        assert cin == cout
        return [_conv(f"{prefix}.cv1", cin // 2, cout // 2, 3), _conv(f"{prefix}.cv2", cin // 2, cout // 2, 1)]
    return [_conv(f"{prefix}.cv1", cin, cout, 3)]


def yolo_conv_specs(a: YoloArch) -> List[ConvSpec]:
    """Every conv of the t/s/m/c graph in state-dict order of appearance."""
    P = "model.list."
    s: List[ConvSpec] = []
    s += [_conv(P + "0", 3, a.stem, 3), _conv(P + "1", a.stem, 2 * a.stem, 3)]
    if a.b2_kind == "elan1":
        h = a.b2_hidden
        s += [_conv(P + "2.cv1", 2 * a.stem, h, 1), _conv(P + "2.cv2", h // 2, h // 2, 3),
              _conv(P + "2.cv3", h // 2, h // 2, 3), _conv(P + "2.cv4", 2 * h, a.b2_out, 1)]
    else:
        s += _elan4(P + "2", 2 * a.stem, a.b2_hidden, a.b2_out, a.rep_n)
    s += _down(P + "3", a.down_kind, a.b2_out, a.d3_out)
    s += _elan4(P + "4", a.d3_out, a.e4_hidden, a.b4_out, a.rep_n)
    s += _down(P + "5", a.down_kind, a.b4_out, a.d5_out)
    s += _elan4(P + "6", a.d5_out, a.e6_hidden, a.p4, a.rep_n)
    s += _down(P + "7", a.down_kind, a.p4, a.d7_out)
    s += _elan4(P + "8", a.d7_out, a.e8_hidden, a.p5, a.rep_n)
    s += [_conv(P + "9.cv1", a.p5, a.spp_hidden, 1), _conv(P + "9.cv5", 4 * a.spp_hidden, a.p5, 1)]
    s += _elan4(P + "12", a.p5 + a.p4, a.e6_hidden, a.p4, a.rep_n)
    s += _elan4(P + "15", a.p4 + a.b4_out, a.e4_hidden, a.p3, a.rep_n)
    s += _down(P + "16", a.down_kind, a.p3, a.d16_out)
    s += _elan4(P + "18", a.d16_out + a.p4, a.e6_hidden, a.p4, a.rep_n)
    s += _down(P + "19", a.down_kind, a.p4, a.d19_out)
    s += _elan4(P + "21", a.d19_out + a.p5, a.e8_hidden, a.p5, a.rep_n)
    H = P + "22."
    for lvl, cin in enumerate((a.p3, a.p4, a.p5)):
        s += [_conv(f"{H}cv2.list.{lvl}.list.0", cin, 64, 3), _conv(f"{H}cv2.list.{lvl}.list.1", 64, 64, 3, 4),
              (f"{H}cv2.list.{lvl}.list.2", 64, 64, 1, 4, True)]
        s += [_conv(f"{H}cv3.list.{lvl}.list.0", cin, a.cls_hidden, 3),
              _conv(f"{H}cv3.list.{lvl}.list.1", a.cls_hidden, a.cls_hidden, 3),
              (f"{H}cv3.list.{lvl}.list.2", a.cls_hidden, a.nc, 1, 1, True)]
    return s


# YOLOv9-e (detection/yolov9.py:328-371): 43 blocks, auxiliary CBLinear/CBFuse branch, all RepNCSP with n=2.
YOLO_E_CBLINEAR = {10: (64, [64]), 11: (256, [64, 128]), 12: (512, [64, 128, 256]), 13: (1024, [64, 128, 256, 512]),
                   14: (1024, [64, 128, 256, 512, 1024])}           # block -> (cin, split sizes); cout = sum(splits)
YOLO_E_ELAN = {3: (128, 32, 256), 5: (256, 64, 512), 7: (512, 128, 1024), 9: (1024, 128, 1024),
               19: (128, 32, 256), 22: (256, 64, 512), 25: (512, 128, 1024), 28: (1024, 128, 1024),
               32: (1536, 128, 512), 35: (1024, 64, 256), 38: (768, 128, 512), 41: (1024, 256, 512)}   # block -> (cin, hid, cout)
YOLO_E_ADOWN = {4: 256, 6: 512, 8: 1024, 20: 256, 23: 512, 26: 1024, 36: 256, 39: 512}              # block -> channels (in == out)


def yolo_e_conv_specs() -> List[ConvSpec]:
    P = "model.list."
    s: List[ConvSpec] = [_conv(P + "1", 3, 64, 3), _conv(P + "2", 64, 128, 3), _conv(P + "15", 3, 64, 3), _conv(P + "17", 64, 128, 3)]
    for i, (cin, hid, cout) in YOLO_E_ELAN.items():
        s += _elan4(P + str(i), cin, hid, cout, 2)
    for i, c in YOLO_E_ADOWN.items():
        s += _down(P + str(i), "adown", c, c)
    for i, (cin, splits) in YOLO_E_CBLINEAR.items():
        s.append((P + f"{i}.conv", cin, sum(splits), 1, 1, True))          # bare nn.Conv2d (CBLinear.conv)
    s += [_conv(P + "29.cv1", 1024, 256, 1), _conv(P + "29.cv5", 1024, 512, 1)]
    H = P + "42."
    for lvl, cin in enumerate((256, 512, 512)):
        s += [_conv(f"{H}cv2.list.{lvl}.list.0", cin, 64, 3), _conv(f"{H}cv2.list.{lvl}.list.1", 64, 64, 3, 4),
              (f"{H}cv2.list.{lvl}.list.2", 64, 64, 1, 4, True)]
        s += [_conv(f"{H}cv3.list.{lvl}.list.0", cin, 256, 3), _conv(f"{H}cv3.list.{lvl}.list.1", 256, 256, 3),
              (f"{H}cv3.list.{lvl}.list.2", 256, 80, 1, 1, True)]
    return s


def yolo_specs(size: str) -> List[ConvSpec]:
    return yolo_e_conv_specs() if size == "e" else yolo_conv_specs(YOLO_ARCH[size])


def yolo_param_count(size: str) -> int:
    n = 16  # dfl
    for _, cin, cout, k, g, _ in yolo_specs(size):
        n += cout * (cin // g) * k * k + cout
    return n


_SCALES = None


def _synth_scales(size: str) -> Dict[str, float]:
    global _SCALES
    if _SCALES is None:
        import json
        import os
        with open(os.path.join(os.path.dirname(__file__), "assets", "synth_scales.json")) as f:
            _SCALES = json.load(f)
    return _SCALES[size]


def synthetic_yolov9_state_dict(size: str = "c", seed: int = 1234, scales=None) -> Dict[str, np.ndarray]:
    """Seeded state dict with the reference's key names (OIHW f32 weights, f32 biases).

    weight = N(0, 1/fan_in) * scale[conv]; the per-conv scale table (assets/synth_scales.json,
    produced once by tools/calibrate_synth.py) keeps every pre-activation at std ~1 through the
    144-conv SiLU stack, and class-logit biases of about -5 let a few dozen anchors clear the 0.25
    threshold so top-300 / mask-NMS get real work.  Deterministic: PCG64(seed) + committed table.
    """
    nc = 80
    if scales is None:
        scales = _synth_scales(size)
    rng = np.random.Generator(np.random.PCG64(seed))
    sd: Dict[str, np.ndarray] = {}
    for prefix, cin, cout, k, g, bare in yolo_specs(size):
        fan_in = (cin // g) * k * k
        w = rng.standard_normal((cout, cin // g, k, k), dtype=np.float32)
        # zero-sum filters: SiLU outputs have a positive mean, and without normalisation layers
        # that DC term swamps the spatial signal after a few dozen convs
        w -= w.mean(axis=(1, 2, 3), keepdims=True, dtype=np.float64).astype(np.float32)
        w *= np.float32(scales.get(prefix, 1.0) / math.sqrt(fan_in))
        b = rng.standard_normal((cout,), dtype=np.float32) * np.float32(0.1)
        if bare and cout == nc and ".cv3." in prefix:
            b = (rng.standard_normal((cout,), dtype=np.float32) * np.float32(0.5) - np.float32(5.0)).astype(np.float32)
        sd[prefix + ".weight"] = w.astype(np.float32)
        sd[prefix + ".bias"] = b.astype(np.float32)
    head = 42 if size == "e" else 22
    sd[f"model.list.{head}.dfl.conv.weight"] = np.arange(16, dtype=np.float32).reshape(1, 16, 1, 1)
    return sd


# ---- the WELL-CONDITIONED synthetic checkpoint (end-to-end tests of the 16-bit storage modes) -------------------------
# synthetic_yolov9_state_dict() above is a chaotic network: white zero-sum filters renormalised layer by layer amplify a
# perturbation of the input 30-60x by the time it reaches P3..P5, so bf16 storage rounding alone moves a fifth of its
# detections and nothing tight can be asserted end to end in the speed modes.  This second generator builds a network a
# trained detector resembles in the one respect that matters for that test — rounding noise does not grow with depth:
#   * 3x3 filters are a random channel mixing of a binomial low-pass kernel plus COND_EPS of white filter, so white
#     (rounding) noise is attenuated at every 3x3 conv while the smooth signal passes;
#   * pre-activations are centred per channel (bias = -mean over a calibration batch) with spatial std COND_STD = 0.1,
#     where SiLU is nearly linear (x*sigmoid(x) = x/2 + x^2/4 - ...), which removes the chaotic renormalisation;
#   * every weight is exactly representable in bf16 (hence in f16): the checkpoint itself is not re-quantised by the
#     speed modes, the f32 oracle and the 16-bit kernels multiply the same numbers and what the test measures is the
#     kernels' storage rounding — as with the fp16 checkpoints real detectors ship as;
#   * head: class logits are normalised per class (mean COND_CLS_BIAS, the COND_Q quantile at logit(0.25)): ~30 detections
#     per noise frame from dozens of classes survive NMS, with scores in 0.25..0.4 so that a logit error shows up as a
#     small score error; DFL logits have std COND_DFL_STD around a ramp of COND_DFL_RAMP per bin (boxes ~20 strides
#     wide, so neighbouring anchors' boxes overlap far beyond the 0.45 NMS threshold and suppression is decisive).
#   What bf16 storage can NOT avoid on any network with Gaussian-like logits: a detection whose logit sits within the
#   accumulated rounding error (~1e-2 of the logit spread after ~50 layers) of the 0.25 threshold, or of its neighbour's
#   score, flips — a few per cent of the detections, which is where the 95 % bar of the 16-bit tests comes from.
# The per-conv gains and per-channel biases come from tools/calibrate_synth.py (assets/synth_cond_<size>.npz, committed);
# the measured perturbation gains and 16-bit emulation results are in assets/synth_cond_report.json.
COND_EPS, COND_STD, COND_RES_FRAC, COND_BIAS_JITTER = 0.2, 0.1, 0.6, 0.02
COND_DFL_STD, COND_DFL_RAMP, COND_CLS_BIAS, COND_Q, COND_ACTIVE_CLASSES = 0.5, 0.15, -1.8, 3e-4, 80
# Stress variants of the conditioned checkpoint (round 5): the same construction with a larger share of white filter and larger
# pre-activations - a network that amplifies perturbations, between the benign checkpoint (f32 perturbation gain 1.4-2 from the input to
# P3..P5) and the chaotic seeded one (30-60x).  name -> (COND_EPS, COND_STD); measured gains in assets/synth_cond_report.json:
# "g3" 2.8 / 3.5 / 3.9 at P3 / P4 / P5, "g10" 8.1 / 11 / 13.
COND_STRESS = {"g3": (1.0, 0.3), "g10": (1.0, 0.6),
               # "nat" (round 6): the benign construction CALIBRATED ON NATURAL-STATISTICS FRAMES (clearcam_amd.streams.natural_frames: 1/f spectrum,
               # flat regions, hard-edged rectangles) instead of white noise - the low-pass filters pass such frames un-attenuated, so the
               # noise-calibrated tables overflow f16 on them; a trained network's normalisation is fitted to its data the same way
               "nat": (COND_EPS, COND_STD)}
COND_NATURAL = ("nat",)                                  # variants whose calibration / evaluation frames are natural_frames(...)
COND_STRESS_CLS_SHIFT = {"g3": 0.4, "g10": 0.0, "nat": 0.6}         # class-logit bias shift: g3's logits are narrow, without it 0-1 detections per frame
_BINOMIAL3 = (np.outer([1.0, 2.0, 1.0], [1.0, 2.0, 1.0]) / 16.0).astype(np.float32)
_COND = {}


def _round_to_bf16(w: np.ndarray) -> np.ndarray:
    """float32 values rounded to the nearest bf16-representable float32 (round-to-nearest-even on the top 16 bits)."""
    u = np.ascontiguousarray(w, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32).reshape(w.shape)


def _storage_exact(w: np.ndarray) -> np.ndarray:
    """Nearest value that bf16 AND f16 both hold exactly: 8 significant bits, magnitudes below f16's smallest normal
    (2^-14, four orders of magnitude under a typical weight) flushed to zero."""
    r = _round_to_bf16(w)
    r[np.abs(r) < np.float32(2.0 ** -14)] = 0.0
    return r


def conditioned_base_weights(size: str, seed: int = 1234, eps: Optional[float] = None) -> Dict[str, np.ndarray]:
    """The seeded filters BEFORE calibration: unit gain, zero bias except the seeded jitter (see the block comment).
    eps: share of white filter in the 3x3 kernels (default COND_EPS; the stress variants use more)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    eps = COND_EPS if eps is None else eps
    sd: Dict[str, np.ndarray] = {}
    for prefix, cin, cout, k, g, bare in yolo_specs(size):
        cg = cin // g
        if k == 3:
            mix = rng.standard_normal((cout, cg, 1, 1), dtype=np.float32)
            white = rng.standard_normal((cout, cg, 3, 3), dtype=np.float32)
            w = (mix * _BINOMIAL3[None, None] * np.float32(16.0 / 6.0) + np.float32(eps) * white) / np.float32(math.sqrt(1.0 + eps ** 2))
        else:
            w = rng.standard_normal((cout, cg, k, k), dtype=np.float32)
        w -= w.mean(axis=(1, 2, 3), keepdims=True, dtype=np.float64).astype(np.float32)
        sd[prefix + ".weight"] = (w * np.float32(1.0 / math.sqrt(cg * k * k))).astype(np.float32)
        sd[prefix + ".bias"] = rng.standard_normal((cout,), dtype=np.float32)
    head = 42 if size == "e" else 22
    sd[f"model.list.{head}.dfl.conv.weight"] = np.arange(16, dtype=np.float32).reshape(1, 16, 1, 1)
    return sd


def _cond_head_out(prefix: str, bare: bool) -> bool:
    return bare and (prefix.startswith("model.list.22.") or prefix.startswith("model.list.42."))


def pack_cond_table(size: str, table) -> Dict[str, np.ndarray]:
    """{"g:<conv>", "b:<conv>", "j:<conv>"} -> three flat float32 arrays in yolo_specs() order (the committed .npz)."""
    gain, bias, jitter = [], [], []
    for prefix, cin, cout, k, g, bare in yolo_specs(size):
        gv = np.asarray(table["g:" + prefix], np.float32).reshape(-1)
        assert gv.size == (cout if _cond_head_out(prefix, bare) else 1), prefix
        gain.append(gv)
        bias.append(np.asarray(table["b:" + prefix], np.float32).reshape(cout))
        jitter.append(np.float32(table["j:" + prefix]))
    return {"gain": np.concatenate(gain), "bias": np.concatenate(bias), "jitter": np.asarray(jitter, np.float32)}


def unpack_cond_table(size: str, gain: np.ndarray, bias: np.ndarray, jitter: np.ndarray):
    table, gi, bi = {}, 0, 0
    for n, (prefix, cin, cout, k, g, bare) in enumerate(yolo_specs(size)):
        ng = cout if _cond_head_out(prefix, bare) else 1
        table["g:" + prefix] = gain[gi:gi + ng].astype(np.float32); gi += ng
        table["b:" + prefix] = bias[bi:bi + cout].astype(np.float32); bi += cout
        table["j:" + prefix] = np.float32(jitter[n])
    assert gi == len(gain) and bi == len(bias), "conditioned-checkpoint table does not match this architecture"
    return table


def conditioned_yolov9_state_dict(size: str = "c", seed: int = 1234, table=None, exact: bool = True, stress: Optional[str] = None) -> Dict[str, np.ndarray]:
    """Seeded, well-conditioned YOLOv9 state dict (same keys and shapes as the reference's checkpoints).

    weight = base filter x gain[conv] (x per-class gain in the head), rounded to values bf16 and f16 hold exactly;
    bias = jitter[conv] x seeded N(0,1) + shift[conv][channel].  `table` (tests / the calibration tool) overrides the
    committed assets/synth_cond_<size>.npz.

    exact=False keeps the float32 products un-rounded, so that a 16-bit mode's re-quantisation of the WEIGHTS is part of
    what a comparison with the f32 oracle measures (as with a trained f32 checkpoint).  On this network that term is
    the larger one by construction: its 3x3 filters are low-pass, which attenuates white activation-rounding noise at
    every layer but passes the smooth, signal-correlated error of a perturbed filter (per-block table in DESIGN.md section 5).

    stress="g3" / "g10" (COND_STRESS, seed 1234 only): the construction with more white filter and larger pre-activations, a network
    whose f32 perturbation gain is ~3-5 / ~8-14 instead of ~1.5 - how far the 16-bit modes' tolerance claim reaches."""
    if stress is not None and (stress not in COND_STRESS or seed != 1234):
        raise ValueError(f"stress must be one of {sorted(COND_STRESS)} (seed 1234)")
    if table is None:
        tag = f"{size}_{stress}" if stress else (size if seed == 1234 else f"{size}_s{seed}")   # the table is data-dependent: one per set of base filters
        if tag not in _COND:
            import os
            path = os.path.join(os.path.dirname(__file__), "assets", f"synth_cond_{tag}.npz")
            if not os.path.exists(path):
                raise FileNotFoundError(f"no conditioned checkpoint table for size '{size}' seed {seed} stress {stress} ({path}); run tools/calibrate_synth.py cond {size} --seed {seed} / --stress NAME")
            with np.load(path) as z:
                _COND[tag] = unpack_cond_table(size, z["gain"], z["bias"], z["jitter"])
        table = _COND[tag]
    sd = conditioned_base_weights(size, seed, COND_STRESS[stress][0] if stress else None)
    for key in list(sd):
        if not key.endswith(".weight") or sd[key].ndim != 4 or ".dfl." in key:
            continue
        prefix = key[:-len(".weight")]
        gain = np.asarray(table.get("g:" + prefix, 1.0), np.float32).reshape(-1, 1, 1, 1)
        sd[key] = _storage_exact(sd[key] * gain) if exact else (sd[key] * gain).astype(np.float32)
        jitter = np.float32(table.get("j:" + prefix, 0.0))
        shift = np.asarray(table.get("b:" + prefix, 0.0), np.float32)
        sd[prefix + ".bias"] = (sd[prefix + ".bias"] * jitter + shift).astype(np.float32)
    if stress and COND_STRESS_CLS_SHIFT.get(stress):
        sd = shift_class_bias(sd, COND_STRESS_CLS_SHIFT[stress])
    return sd


def shift_class_bias(sd: Dict[str, np.ndarray], shift: float) -> Dict[str, np.ndarray]:
    """Copy of a YOLOv9 state dict with every class-logit bias moved by `shift`.  The seeded weights fire on ~260
    anchors of a noise frame (good for top-k/NMS parity); a negative shift gives the sparse detections of a real
    scene for tracker-facing benchmarks without changing a single FLOP of the detector."""
    out = dict(sd)
    for k, v in sd.items():
        if ".cv3." in k and k.endswith(".2.bias") and v.shape == (80,):
            out[k] = (v + np.float32(shift)).astype(np.float32)
    return out


# ----------------------------------------------------------------------------------------------
# CLIP
# ----------------------------------------------------------------------------------------------

def clip_shapes(a: ClipArch = CLIP_L14) -> Dict[str, Tuple[int, ...]]:
    """Key -> shape for the OpenCLIP state dict (``models/objects.py:29-89``)."""
    W, T = a.v_width, a.t_width
    s: Dict[str, Tuple[int, ...]] = {
        "visual_conv1.weight": (W, 3, a.patch, a.patch),
        "class_embedding": (W,),
        "positional_embedding": (a.v_tokens, W),
        "ln_pre.weight": (W,), "ln_pre.bias": (W,),
        "ln_post.weight": (W,), "ln_post.bias": (W,),
        "proj": (W, a.embed),
        "token_embedding.weight": (a.t_vocab, T),
        "positional_embedding_text": (a.t_ctx, T),
        "ln_final.weight": (T,), "ln_final.bias": (T,),
        "text_projection": (T, a.embed),
    }
    for i in range(a.v_layers):
        p = f"resblocks_img.{i}."
        s.update({p + "ln_1.weight": (W,), p + "ln_1.bias": (W,), p + "ln_2.weight": (W,), p + "ln_2.bias": (W,),
                  p + "in_proj_weight": (3 * W, W), p + "in_proj_bias": (3 * W,),
                  p + "out_proj_weight": (W, W), p + "out_proj_bias": (W,),
                  p + "mlp_c_fc.weight": (a.v_mlp, W), p + "mlp_c_fc.bias": (a.v_mlp,),
                  p + "mlp_c_proj.weight": (W, a.v_mlp), p + "mlp_c_proj.bias": (W,)})
    for i in range(a.t_layers):
        p = f"resblocks.{i}."
        s.update({p + "ln_1.weight": (T,), p + "ln_1.bias": (T,), p + "ln_2.weight": (T,), p + "ln_2.bias": (T,),
                  p + "in_proj_weight": (3 * T, T), p + "in_proj_bias": (3 * T,),
                  p + "attn_out_proj_weight": (T, T), p + "attn_out_proj_bias": (T,),
                  p + "mlp_c_fc.weight": (a.t_mlp, T), p + "mlp_c_fc.bias": (a.t_mlp,),
                  p + "mlp_c_proj.weight": (T, a.t_mlp), p + "mlp_c_proj.bias": (T,)})
    return s


def synthetic_clip_state_dict(a: ClipArch = CLIP_L14, seed: int = 4321) -> Dict[str, np.ndarray]:
    """Seeded CLIP state dict (transformer-style init: N(0, 0.02)-ish matrices, LN gain 1)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    sd: Dict[str, np.ndarray] = {}
    for k, shp in clip_shapes(a).items():
        if k.endswith("ln_1.weight") or k.endswith("ln_2.weight") or k in ("ln_pre.weight", "ln_post.weight", "ln_final.weight"):
            v = 1.0 + 0.1 * rng.standard_normal(shp, dtype=np.float32)
        elif k.endswith(".bias") or k.endswith("_bias"):
            v = 0.02 * rng.standard_normal(shp, dtype=np.float32)
        elif len(shp) == 1:
            v = 0.05 * rng.standard_normal(shp, dtype=np.float32)
        elif k == "visual_conv1.weight":
            v = rng.standard_normal(shp, dtype=np.float32) / math.sqrt(shp[1] * shp[2] * shp[3])
        elif "positional" in k or k == "token_embedding.weight":
            v = 0.05 * rng.standard_normal(shp, dtype=np.float32)
        elif k in ("proj", "text_projection"):
            v = rng.standard_normal(shp, dtype=np.float32) / math.sqrt(shp[0])
        else:  # (out, in) linear weights
            v = rng.standard_normal(shp, dtype=np.float32) * (0.7 / math.sqrt(shp[1]))
        sd[k] = v.astype(np.float32)
    return sd


# ----------------------------------------------------------------------------------------------
# real weights
# ----------------------------------------------------------------------------------------------

def load_safetensors(path: str) -> Dict[str, np.ndarray]:
    """Load a reference checkpoint (same key names) as float32 numpy arrays."""
    from safetensors.numpy import load_file
    return {k: np.ascontiguousarray(v, dtype=np.float32) if v.dtype != np.float32 else v
            for k, v in load_file(path).items()}


# ----------------------------------------------------------------------------------------------
# AdaFace IR-50 (models/adaface.py)
# ----------------------------------------------------------------------------------------------
ADAFACE_BLOCKS = [(64, 64, 2), (64, 64, 1), (64, 64, 1), (64, 128, 2), (128, 128, 1), (128, 128, 1), (128, 128, 1), (128, 256, 2)] + \
                 [(256, 256, 1)] * 13 + [(256, 512, 2), (512, 512, 1), (512, 512, 1)]


def synthetic_adaface_state_dict(seed: int = 777) -> Dict[str, np.ndarray]:
    """Seeded IR-50 state dict with the reference's parameter names (tinygrad get_state_dict of models/adaface.py:61-76).
    He-scaled zero-sum conv filters; BatchNorm statistics near (0, 1) with mild per-channel spread so that folding
    mistakes show; the residual branch is damped so that 24 blocks keep O(1) activations."""
    rng = np.random.Generator(np.random.PCG64(seed))

    def conv(co, ci, k, gain=1.0):
        w = rng.standard_normal((co, ci, k, k), dtype=np.float32)
        w -= w.mean(axis=(1, 2, 3), keepdims=True)
        return (w * np.float32(gain * math.sqrt(2.0 / (ci * k * k)))).astype(np.float32)

    def bn(sd, p, c, affine=True, wscale=1.0):
        if affine:
            sd[p + ".weight"] = (wscale * (1.0 + 0.1 * rng.standard_normal(c))).astype(np.float32)
            sd[p + ".bias"] = (0.05 * rng.standard_normal(c)).astype(np.float32)
        sd[p + ".running_mean"] = (0.05 * rng.standard_normal(c)).astype(np.float32)
        sd[p + ".running_var"] = (1.0 + 0.2 * rng.random(c)).astype(np.float32)

    sd: Dict[str, np.ndarray] = {}
    sd["conv0.weight"] = conv(64, 3, 3)
    bn(sd, "bn0", 64)
    sd["prelu_weight"] = (0.25 + 0.05 * rng.standard_normal(64)).astype(np.float32)
    for i, (cin, depth, stride) in enumerate(ADAFACE_BLOCKS):
        p = f"body.list.{i}."
        bn(sd, p + "res_layer0", cin)
        sd[p + "conv_layer0.weight"] = conv(depth, cin, 3)
        bn(sd, p + "res_layer1", depth)
        sd[p + "prelu_weight"] = (0.25 + 0.05 * rng.standard_normal(depth)).astype(np.float32)
        sd[p + "conv_layer1.weight"] = conv(depth, depth, 3)
        bn(sd, p + "res_layer2", depth, wscale=0.5)
        if cin != depth:
            sd[p + "shortcut_layer0.weight"] = conv(depth, cin, 1, gain=0.7)
            bn(sd, p + "shortcut_layer1", depth)
    bn(sd, "bn", 512)
    sd["linear.weight"] = (rng.standard_normal((512, 512 * 7 * 7), dtype=np.float32) * np.float32(1.0 / math.sqrt(512 * 49))).astype(np.float32)
    sd["linear.bias"] = (0.01 * rng.standard_normal(512)).astype(np.float32)
    bn(sd, "bn2", 512, affine=False)
    return sd


# ----------------------------------------------------------------------------------------------
# BlazeFace (models/blazeface.py)
# ----------------------------------------------------------------------------------------------
BLAZE_BLOCKS = [(24, 24, 1)] * 7 + [(24, 24, 2)] + [(24, 24, 1)] * 7 + [(24, 48, 2)] + [(48, 48, 1)] * 7 + [(48, 96, 2)] + [(96, 96, 1)] * 7


def blazeface_anchors() -> np.ndarray:
    """The 896 MediaPipe BlazeFace anchors (x_center, y_center, w, h) = cell centres of a 16x16 grid (2 per cell) then of an
    8x8 grid (6 per cell), unit size.  The reference loads them from its checkpoint (models/blazeface.py:124)."""
    out = []
    for g, n in ((16, 2), (8, 6)):
        for y in range(g):
            for x in range(g):
                out += [[(x + 0.5) / g, (y + 0.5) / g, 1.0, 1.0]] * n
    return np.asarray(out, np.float32)


def synthetic_blazeface_state_dict(seed: int = 555) -> Dict[str, np.ndarray]:
    """Seeded BlazeFace state dict with the reference's parameter names.  The score head is biased so that a few dozen of
    the 896 anchors clear the 0.85 threshold and the overlap rule gets real work."""
    rng = np.random.Generator(np.random.PCG64(seed))

    def conv(sd, name, co, ci, k, groups=1, gain=1.0, bias=0.05):
        w = rng.standard_normal((co, ci // groups, k, k), dtype=np.float32)
        sd[name + ".weight"] = (w * np.float32(gain * math.sqrt(2.0 / (ci // groups * k * k)))).astype(np.float32)
        sd[name + ".bias"] = (bias * rng.standard_normal(co)).astype(np.float32)

    sd: Dict[str, np.ndarray] = {}
    conv(sd, "conv_tiny", 24, 3, 5)
    for i, (cin, cout, stride) in enumerate(BLAZE_BLOCKS):          # gains chosen so 31 un-normalised ReLU blocks stay O(1)
        p = f"backbone_tiny.list.{i}."
        conv(sd, p + "conv0_tiny", cin, cin, 3, groups=cin, gain=0.6)
        conv(sd, p + "conv1_tiny", cout, cin, 1, gain=0.3)
    conv(sd, "final.conv0_tiny", 96, 96, 3, groups=96, gain=0.6)
    conv(sd, "final.conv1_tiny", 96, 96, 1, gain=0.7)
    for name, co in (("classifier_8_tiny", 2), ("classifier_16_tiny", 6)):
        conv(sd, name, co, 96, 1)
        sd[name + ".bias"] = (sd[name + ".bias"] - np.float32(3.0)).astype(np.float32)     # ~3 % of the anchors clear 0.85
    for name, co in (("regressor_8_tiny", 32), ("regressor_16_tiny", 96)):
        conv(sd, name, co, 96, 1, gain=6.0)
        b = sd[name + ".bias"]
        b[2::16] += 60.0; b[3::16] += 60.0                                                   # box width / height ~0.23 of the image
        sd[name + ".bias"] = b.astype(np.float32)
    sd["anchors"] = blazeface_anchors()
    return sd

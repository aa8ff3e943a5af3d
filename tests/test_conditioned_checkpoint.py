"""The well-conditioned synthetic checkpoint (clearcam_amd.weights.conditioned_yolov9_state_dict) on the CPU.

What the GPU tests of the 16-bit modes rest on: the checkpoint is deterministic, has the reference's key set, every
weight survives bf16 / f16 storage unchanged, its measured perturbation gain is ~1, and a correct 16-bit implementation
(the fp32 oracle with storage rounding applied where the HIP path rounds, oracle/lowprec_oracle.py) meets the bars that
tests/test_gpu_yolo.py holds the kernels to.
"""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

from clearcam_amd import weights as W
from conftest import noise_frames
from oracle import yolov9_oracle as yo
from oracle.lowprec_oracle import LowPrecOracle, rel_rms


@pytest.fixture(scope="module")
def sd():
    return W.conditioned_yolov9_state_dict("c", 1234)


def test_same_keys_and_shapes_as_the_reference_checkpoint(sd, sd_c):
    assert set(sd) == set(sd_c)
    assert all(sd[k].shape == sd_c[k].shape and sd[k].dtype == np.float32 for k in sd)


def _digest(d):
    h = hashlib.sha256()
    for k in sorted(d):
        h.update(np.ascontiguousarray(d[k]).tobytes())
    return h.hexdigest()[:16]


def test_deterministic(sd):
    again = W.conditioned_yolov9_state_dict("c", 1234)
    assert all(np.array_equal(sd[k], again[k]) for k in sd)
    other = W.conditioned_yolov9_state_dict("c", 99)
    assert not np.array_equal(sd["model.list.4.cv1.conv.weight"], other["model.list.4.cv1.conv.weight"])
    # the seeded draws are PCG64 (platform independent); the digest pins generator + committed table together
    golden = os.path.join(os.path.dirname(__file__), "golden")
    assert _digest(sd) == open(os.path.join(golden, "synth_cond_c.sha256")).read().strip()


@pytest.mark.parametrize("seed", [7, 99])
def test_further_checkpoints_are_pinned_and_16bit_exact(seed):
    """The two further conditioned checkpoints (own calibration tables: assets/synth_cond_c_s<seed>.npz): pinned by digest, weights exact in
    both 16-bit storage types in the exact=True form, un-rounded (and different) in the exact=False form the tolerance-mode tests use."""
    d = W.conditioned_yolov9_state_dict("c", seed)
    assert _digest(d) == open(os.path.join(os.path.dirname(__file__), "golden", f"synth_cond_c_s{seed}.sha256")).read().strip()
    raw = W.conditioned_yolov9_state_dict("c", seed, exact=False)
    k = "model.list.4.cv1.conv.weight"
    t = torch.from_numpy(d[k])
    assert torch.equal(t.to(torch.float16).float(), t) and torch.equal(t.to(torch.bfloat16).float(), t)
    r = torch.from_numpy(raw[k])
    assert not torch.equal(r.to(torch.float16).float(), r) and float((r - t).abs().max()) <= float(t.abs().max()) * 2.0 ** -8
    with pytest.raises(FileNotFoundError):
        W.conditioned_yolov9_state_dict("c", 5)                              # no table for this seed: calibrate first


def test_weights_are_exact_in_both_16bit_storage_types(sd):
    for k, v in sd.items():
        if k.endswith(".weight") and v.ndim == 4 and ".dfl." not in k:
            t = torch.from_numpy(v)
            assert torch.equal(t.to(torch.bfloat16).float(), t), k
            assert torch.equal(t.to(torch.float16).float(), t), k           # 8 significant bits, exponents far inside f16's range


@pytest.mark.parametrize("tag", ["c", "c_s7", "c_s99"])
def test_committed_conditioning_report(tag):
    """One calibration table per checkpoint (the gains and bias shifts are data-dependent): seed 1234 and the two further checkpoints the
    GPU parity tests of the tolerance modes run on (tools/calibrate_synth.py cond c --seed N)."""
    rep = json.load(open(os.path.join(os.path.dirname(W.__file__), "assets", "synth_cond_report.json")))[tag]
    for where, g in rep["f32_perturbation_gain_at_p3_p4_p5"].items():
        assert max(g) <= 2.0, (where, g)                                     # white noise injected anywhere does not grow
    assert rep["bf16_storage_emulation"]["match_frac"] >= 0.95 and rep["f16_storage_emulation"]["match_frac"] >= 0.95


@pytest.mark.parametrize("dtype,feat_bar", [("bf16", 3e-2), ("f16", 4e-3)])
def test_storage_rounding_alone_meets_the_gpu_bars(sd, dtype, feat_bar):
    """Twelve 640x640 noise frames (the report's frames); the GPU test runs the same comparison at B=64 against the real kernels."""
    frames = noise_frames(1, 12, 640, 640)
    o = yo.YOLOv9Oracle("c", 640, sd)
    lo = LowPrecOracle("c", 640, sd, dtype)
    tot = np.zeros(3, int)
    sc = 0.0
    for i in range(0, 12, 4):
        ref, got = o.detect_batch(frames[i:i + 4]), lo.detect_batch(frames[i:i + 4])
        for blk in (15, 18, 21):
            assert rel_rms(lo.block_outputs[blk], o.block_outputs[blk]) <= feat_bar
        for b in range(4):
            a, c, k, _, se = yo.match_detections(ref[b], got[b], 0.9)
            tot += (a, c, k); sc = max(sc, se)
    assert tot[0] >= 100 and tot[2] >= 0.95 * max(tot[0], tot[1]), tot
    assert sc <= 1e-2, sc


@pytest.mark.parametrize("stress,lo,hi", [("g3", 2.0, 6.0), ("g10", 6.0, 20.0)])
def test_stress_checkpoints_are_pinned_and_amplify_as_stated(stress, lo, hi):
    """The stress variants (weights.COND_STRESS: more white filter, larger pre-activations): pinned by digest, the reference's key set, and the
    committed report's measured f32 perturbation gain from the network input to P3 / P4 / P5 inside the stated band - between the benign
    checkpoints (<= 2) and the chaotic one (30-60).  tests/test_gpu_yolo.py::test_tolerance_modes_on_stress_checkpoints states what the
    16-bit modes do there."""
    d = W.conditioned_yolov9_state_dict("c", 1234, stress=stress)
    assert _digest(d) == open(os.path.join(os.path.dirname(__file__), "golden", f"synth_cond_c_{stress}.sha256")).read().strip()
    assert set(d) == set(W.conditioned_yolov9_state_dict("c", 1234))
    rep = json.load(open(os.path.join(os.path.dirname(W.__file__), "assets", "synth_cond_report.json")))[f"c_{stress}"]
    g = rep["f32_perturbation_gain_at_p3_p4_p5"]["network input"]
    assert lo <= min(g) and max(g) <= hi, g
    assert rep["stress"] == {"eps": W.COND_STRESS[stress][0], "preact_std": W.COND_STRESS[stress][1]}
    with pytest.raises(ValueError):
        W.conditioned_yolov9_state_dict("c", 7, stress=stress)               # the stress tables exist for seed 1234 only

"""CPU tests of the detector oracle: structure pins, integer letterbox arithmetic, NMS semantics, golden vectors.

The reference holds no boxes/detections fixture ("parity unpinned"); what it does pin —
the YOLOv9 paper's parameter/MAC counts, the state-dict key layout, tinygrad's uint8 lerp — is checked here.
"""
import os

import numpy as np
import pytest
import torch

from clearcam_amd.arch import YOLO_ARCH
from clearcam_amd.weights import synthetic_yolov9_state_dict, yolo_conv_specs, yolo_param_count
from oracle import yolov9_oracle as yo

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_param_counts_match_published():
    # YOLOv9 paper / SURVEY.md Appendix A: t 2.00 M, s 7.11 M, m 19.98 M, c 25.29 M
    assert yolo_param_count("t") == 2_001_840
    assert yolo_param_count("s") == 7_105_888
    assert yolo_param_count("m") == 19_978_672
    assert yolo_param_count("c") == 25_288_768
    assert abs(yolo_param_count("e") / 1e6 - 57.3) < 0.1       # YOLOv9-E: 57.3 M (paper)


def test_state_dict_keys_follow_reference_attribute_tree(sd_c):
    # SURVEY.md Appendix C (detection/yolov9.py attribute names; Sequential adds ".list.")
    for k in ["model.list.0.conv.weight", "model.list.2.cv2.list.0.m.list.0.cv1.conv.bias",
              "model.list.2.cv2.list.1.conv.weight", "model.list.3.cv2.conv.weight", "model.list.9.cv5.conv.bias",
              "model.list.22.cv2.list.1.list.2.weight", "model.list.22.cv3.list.2.list.0.conv.weight",
              "model.list.22.dfl.conv.weight"]:
        assert k in sd_c, k
    assert sd_c["model.list.22.cv2.list.0.list.1.conv.weight"].shape == (64, 16, 3, 3)     # groups=4
    assert sd_c["model.list.22.cv3.list.0.list.2.weight"].shape == (80, 256, 1, 1)
    assert len(yolo_conv_specs(YOLO_ARCH["c"])) == 144                                       # 144 convs in "c"


def test_mac_count_c_640(sd_c):
    """51.068 GMAC per 640x640 frame — counted by hooking every conv of one forward pass."""
    o = yo.YOLOv9Oracle("c", 640, sd_c)
    macs = [0]
    orig = o._conv2d

    def counting(x, name, stride=1, groups=1):
        y = orig(x, name, stride, groups)
        w = o.sd[name + ".weight"]
        macs[0] += y.shape[2] * y.shape[3] * w.numel()
        return y
    o._conv2d = counting
    with torch.no_grad():
        o.head_raw(o.features(torch.zeros(1, 3, 640, 640)))
    assert abs(macs[0] / 1e9 - 51.068) < 0.01


def test_letterbox_geometry():
    assert yo.letterbox_geometry(1080, 1920, 640) == (360, 640, 12, 0)      # -> 384x640
    assert yo.letterbox_geometry(640, 640, 640) == (640, 640, 0, 0)
    assert yo.letterbox_geometry(480, 640, 640) == (480, 640, 0, 0)         # 160 % 32 == 0 -> no pad
    assert yo.letterbox_geometry(720, 1280, 960) == (540, 960, 2, 0)        # 420 % 32 = 4 -> 2 each side -> 544x960


def test_uint8_lerp_fixed_point_known_answers():
    # tinygrad lerp for uint8: w=int16(frac*128+.5); a + ((int8(b-a)*w + 64) >> 7) mod 256
    a = np.array([10, 200, 0, 255, 100], np.uint8)
    b = np.array([20, 100, 255, 0, 100], np.uint8)
    f = np.array([0.5, 0.25, 0.5, 0.5, 0.9], np.float32)
    got = yo._lerp_u8(a, b, f)
    # hand-computed: d=int8(b-a) -> [10, -100, -1, 1, 0]; w -> [64, 32, 64, 64, 115]
    #   t=(d*w+64)>>7 (arithmetic) -> [5, -25, 0, 1, 0]
    assert got.tolist() == [15, 175, 0, 0, 100]     # note 0->255 at 0.5 stays 0 and 255->0 wraps to 0: int8 wrap quirk


def test_resize_identity_and_dtype_paths():
    f = np.random.default_rng(0).integers(0, 256, (37, 53, 3), dtype=np.uint8)
    assert np.array_equal(yo.resize_bilinear(f, 37, 53), f)
    up_u8 = yo.resize_bilinear(f, 74, 106)
    up_f = yo.resize_bilinear(f.astype(np.float32), 74, 106)
    assert up_u8.dtype == np.uint8 and up_f.dtype == np.float32
    # smooth ramp: fixed point within 1 LSB per axis of the float path (no int8 wrap when |b-a| < 128)
    ramp = np.tile(np.arange(60, dtype=np.uint8)[None, :, None] * 2, (20, 1, 3))
    d = yo.resize_bilinear(ramp, 31, 97).astype(np.float32) - yo.resize_bilinear(ramp.astype(np.float32), 31, 97)
    assert np.abs(d).max() <= 2.0


def test_letterbox_golden():
    g = np.load(os.path.join(GOLD, "letterbox_135x240_to_160.npz"))
    f = np.random.default_rng(3).integers(0, 256, (135, 240, 3), dtype=np.uint8)
    assert np.array_equal(yo.letterbox(f, 160), g["u8"])
    assert np.array_equal(yo.letterbox(f.astype(np.float32), 160), g["f32"])
    assert g["u8"].shape == (96, 160, 3) and (g["u8"][:3] == 0).all() and (g["u8"][-3:] == 0).all()   # zero pad, not 114


def _post_input(boxes_xyxy, scores, classes, n_anchor=400):
    out = torch.zeros(1, 84, n_anchor)
    for i, (b, s, c) in enumerate(zip(boxes_xyxy, scores, classes)):
        x1, y1, x2, y2 = b
        out[0, 0, i], out[0, 1, i], out[0, 2, i], out[0, 3, i] = (x1 + x2) / 2, (y1 + y2) / 2, x2 - x1, y2 - y1
        out[0, 4 + c, i] = s
    return out


def test_postprocess_mask_nms_semantics():
    """Non-greedy mask NMS (yolov9.py:439-458): row j dies if ANY earlier row i (even a suppressed one) of the
    same class overlaps > 0.45; different classes never suppress; below-threshold scores become 0."""
    boxes = [(0, 0, 10, 10), (1, 0, 11, 10), (2, 0, 12, 10), (0, 0, 10, 10), (50, 50, 60, 60)]
    scores = [0.9, 0.8, 0.7, 0.6, 0.2]
    classes = [3, 3, 3, 5, 3]
    r = yo.YOLOv9Oracle.postprocess(_post_input(boxes, scores, classes))[0].numpy()
    assert r.shape == (300, 6)
    assert np.allclose(r[0], [0, 0, 10, 10, 0.9, 3])
    assert (r[1] == 0).all()                       # IoU(0,1)=0.818 > .45
    assert (r[2] == 0).all()                       # IoU(0,2)=0.667 > .45 (also killed by the already-dead row 1)
    assert np.allclose(r[3], [0, 0, 10, 10, 0.6, 5])   # other class survives
    assert r[4][4] == 0                             # 0.2 < conf threshold -> score zeroed (row kept unsuppressed)
    # chain: a(0..10) b(4..14) c(8..18): IoU(a,b)=.43 no, IoU(b,c)=.43 no -> all survive
    r = yo.YOLOv9Oracle.postprocess(_post_input([(0, 0, 10, 10), (4, 0, 14, 10), (8, 0, 18, 10)], [.9, .8, .7], [1, 1, 1]))[0].numpy()
    assert (r[:3, 4] > 0).all()


def test_postprocess_stable_ties():
    boxes = [(i * 20, 0, i * 20 + 10, 10) for i in range(6)]
    r = yo.YOLOv9Oracle.postprocess(_post_input(boxes, [0.5] * 6, [0] * 6))[0].numpy()
    assert r[:6, 0].tolist() == [0, 20, 40, 60, 80, 100]      # equal scores keep anchor order


def test_scale_boxes_clip():
    p = torch.tensor([[[-5.0, 10.0, 700.0, 380.0, 0.9, 1.0]]])
    q = yo.YOLOv9Oracle.scale_boxes((384, 640), p, (1080, 1920)).numpy()[0, 0]
    assert np.allclose(q, [0.0, 0.0, 1920.0, 1080.0, 0.9, 1.0])   # (y-12)*3 clipped to the source frame


@pytest.mark.parametrize("name", ["yolo_t_640", "yolo_t_640_from_540x960"])
def test_oracle_reproduces_golden(name, sd_t):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    frames = np.random.default_rng(int(g["seed"])).integers(0, 256, tuple(g["shape"]), dtype=np.uint8)
    det = yo.YOLOv9Oracle("t", int(g["res"]), sd_t).detect_batch(frames)
    for b in range(det.shape[0]):
        n_ref, n_got, n_match, box_err, sc_err = yo.match_detections(g["det"][b], det[b], 0.9)
        assert n_ref == n_got == n_match and box_err < 0.05 and sc_err < 1e-3


def test_oracle_batch_equals_single(sd_t):
    frames = np.random.default_rng(1).integers(0, 256, (2, 640, 640, 3), dtype=np.uint8)
    o = yo.YOLOv9Oracle("t", 640, sd_t)
    both = o.detect_batch(frames)
    one = o(frames[1])
    assert yo.match_detections(both[1], one, 0.9)[:3] == ((both[1][:, 4] > 0).sum(),) * 3

"""ORACLE (test infrastructure, not product code): CPU fp32 restatement of the reference detector.

Restates ``detection/yolov9.py`` (whole file) and ``utils/helpers.py:127-131`` (``resize``) of
roryclear/clearcam in PyTorch-CPU float32 + numpy integer arithmetic.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module; the
product path (``clearcam_amd``) never does.

PIN STATUS: the reference stores no detections/boxes anywhere (``test/tracks.pkl`` and
``test/tracker_inputs.pkl`` are missing blobs), tinygrad (pinned ``fe39cf14``, not vendored) cannot
be imported here and no trained weights exist offline.  What pins this file: the reference's OWN
``detection/yolov9.py`` + ``utils/helpers.resize`` executed unchanged over a PyTorch stand-in for
tinygrad (tools/refshim, tools/make_reference_run_golden.py) on seeded checkpoints for all five sizes;
tests/test_reference_run.py requires row-for-row agreement (same surviving rows and classes, boxes
<= 0.1 px, scores <= 2e-4; measured 0.04 px / 6e-5 = float32 reassociation).  That fixes the layer
wiring, concat orders, decode, top-k, NMS and box scaling to the reference's code.  STILL UNPINNED
("parity unpinned" for these): tinygrad's own kernels, i.e. summation order, the uint8 fixed-point
``interpolate`` / ``lerp`` and the tie order of ``topk`` (restated in both this file and the stand-in),
and trained weights.  Also pinned: parameter count 25.29 M / 51.07 GMAC for size "c"
(the YOLOv9 paper's figures), the state-dict key set (the reference's strict ``load_state_dict``
accepts the synthetic checkpoints) and the integer letterbox arithmetic (tests/test_oracle_yolo.py).
tinygrad semantics taken from SURVEY.md
Appendix B: uint8 ``lerp`` fixed point, zero letterbox padding, ``avg_pool2d`` count_include_pad,
``max_pool2d`` -inf padding, stable descending top-k.  One third-party detail cannot be verified
offline: tinygrad may lower ``x / c`` to ``x * (1/c)``; this restatement divides (<= 1 ulp apart).

Every function cites the reference lines it follows.
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from clearcam_amd.arch import YOLO_ARCH, YoloArch

MAX_DET = 300
CONF_THRESHOLD = 0.25
IOU_THRESHOLD = 0.45


# ----------------------------------------------------------------------------------------------
# letterbox  (detection/yolov9.py:390-404, utils/helpers.py:127-131, tinygrad interpolate/lerp)
# ----------------------------------------------------------------------------------------------

def letterbox_geometry(h: int, w: int, res: int, stride: int = 32) -> Tuple[int, int, int, int]:
    """(new_h, new_w, pad_y, pad_x) exactly as ``YOLOv9.preprocess`` computes them (:390-403)."""
    r = min(res / h, res / w)
    new_w, new_h = int(round(w * r)), int(round(h * r))
    dw, dh = (res - new_w) % stride, (res - new_h) % stride
    dw, dh = dw / 2, dh / 2
    return new_h, new_w, int(round(dh - 0.1)), int(round(dw - 0.1))


def interp_axis_tables(n_in: int, n_out: int) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """low, high, frac for one axis of ``Tensor.interpolate(mode='linear', align_corners=False)``.

    index = clip(scale*(i+0.5)-0.5, 0, n_in-1) evaluated in float32 (Appendix B-1)."""
    scale = np.float32(n_in / n_out)
    arr = np.arange(n_out, dtype=np.float32)
    idx = (scale * (arr + np.float32(0.5))) - np.float32(0.5)
    idx = np.clip(idx, np.float32(0), np.float32(n_in - 1)).astype(np.float32)
    low = np.floor(idx).astype(np.int32)
    high = np.ceil(idx).astype(np.int32)
    frac = (idx - np.floor(idx)).astype(np.float32)
    return low, high, frac


def _lerp_u8(a: np.ndarray, b: np.ndarray, frac: np.ndarray) -> np.ndarray:
    """tinygrad ``Tensor.lerp`` for uint8: 7-bit fixed point with int8-wrapped difference."""
    w = (frac * np.float32(128) + np.float32(0.5)).astype(np.int16)            # truncating cast
    d = (b.astype(np.uint8) - a.astype(np.uint8)).astype(np.uint8).view(np.int8).astype(np.int16)
    t = (d * w + np.int16(64)).astype(np.int16).view(np.uint16) >> np.uint16(7)
    return (a.astype(np.uint16) + t).astype(np.uint8)


def resize_bilinear(img: np.ndarray, new_h: int, new_w: int) -> np.ndarray:
    """``helpers.resize`` (:127-131): separable, W axis first then H, no antialias, dtype kept."""
    assert img.ndim == 3
    lw, hw, fw = interp_axis_tables(img.shape[1], new_w)
    lh, hh, fh = interp_axis_tables(img.shape[0], new_h)
    if img.dtype == np.uint8:
        x = _lerp_u8(img[:, lw, :], img[:, hw, :], fw[None, :, None])
        x = _lerp_u8(x[lh, :, :], x[hh, :, :], fh[:, None, None])
        return x
    x = img.astype(np.float32)
    a, b = x[:, lw, :], x[:, hw, :]
    x = a + (b - a) * fw[None, :, None]
    a, b = x[lh, :, :], x[hh, :, :]
    x = a + (b - a) * fh[:, None, None]
    return x.astype(img.dtype)


def letterbox(frame: np.ndarray, res: int) -> np.ndarray:
    """``YOLOv9.preprocess`` (:390-404): resize + symmetric zero pad (HWC, dtype preserved)."""
    h, w = frame.shape[:2]
    nh, nw, py, px = letterbox_geometry(h, w, res)
    img = resize_bilinear(frame, nh, nw)
    return np.pad(img, ((py, py), (px, px), (0, 0)))


# ----------------------------------------------------------------------------------------------
# graph  (detection/yolov9.py:33-155, 298-326)
# ----------------------------------------------------------------------------------------------

class YOLOv9Oracle:
    """``YOLOv9(size, res)`` for sizes t/s/m/c with an explicit state dict (no download)."""

    def __init__(self, size: str, res: int, state_dict: Dict[str, np.ndarray]):
        self.size = size
        self.arch = YOLO_ARCH.get(size)              # None for "e" (its graph is spelled out in features_e)
        self.rep_n = 2 if size == "e" else self.arch.rep_n
        self.adown_kind = size in ("c", "e")
        self.head = "model.list.42." if size == "e" else "model.list.22."
        self.res = res
        self.sd = {k: torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)) for k, v in state_dict.items()}

    # -- primitive blocks ---------------------------------------------------------------------
    def _conv2d(self, x, name, stride=1, groups=1):
        w = self.sd[name + ".weight"]
        return F.conv2d(x, w, self.sd[name + ".bias"], stride=stride, padding=w.shape[-1] // 2, groups=groups)

    def conv(self, x, name, stride=1, groups=1):
        """``Conv.__call__`` :33-38 — conv + bias then SiLU."""
        return F.silu(self._conv2d(x, name + ".conv", stride, groups))

    def adown(self, x, p):  # :40-52
        x = F.avg_pool2d(x, 2, 1, 0, False, True)
        x1, x2 = x.chunk(2, 1)
        x1 = self.conv(x1, p + ".cv1", stride=2)
        x2 = F.max_pool2d(x2, 3, 2, 1)
        x2 = self.conv(x2, p + ".cv2")
        return torch.cat((x1, x2), 1)

    def aconv(self, x, p):  # :54-63
        return self.conv(F.avg_pool2d(x, 2, 1, 0, False, True), p + ".cv1", stride=2)

    def down(self, x, p):
        return self.adown(x, p) if self.adown_kind else self.aconv(x, p)

    def elan1(self, x, p):  # :65-80
        y = list(self.conv(x, p + ".cv1").chunk(2, 1))
        y.append(self.conv(y[-1], p + ".cv2"))
        y.append(self.conv(y[-1], p + ".cv3"))
        return self.conv(torch.cat(y, 1), p + ".cv4")

    def repncsp(self, x, p):  # :82-105
        x2 = self.conv(x, p + ".cv1")
        for j in range(self.rep_n):
            q = f"{p}.m.list.{j}"
            x2 = x2 + self.conv(self.conv(x2, q + ".cv1"), q + ".cv2")
        return self.conv(torch.cat((x2, self.conv(x, p + ".cv2")), 1), p + ".cv3")

    def elan4(self, x, p):  # :107-125
        x = self.conv(x, p + ".cv1")
        y0, y1 = x.chunk(2, 1)
        y2 = self.conv(self.repncsp(y1, p + ".cv2.list.0"), p + ".cv2.list.1")
        y3 = self.conv(self.repncsp(y2, p + ".cv3.list.0"), p + ".cv3.list.1")
        return self.conv(torch.cat((y0, y1, y2, y3), 1), p + ".cv4")

    def sppelan(self, x, p):  # :127-149
        y = [self.conv(x, p + ".cv1")]
        for _ in range(3):
            y.append(F.max_pool2d(y[-1], 5, 1, 2))
        return self.conv(torch.cat(y, 1), p + ".cv5")

    @staticmethod
    def upsample(x):  # :285-292
        return x.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)

    # -- YOLOv9-e: auxiliary reversible branch (CBLinear / CBFuse) ---------------------------------
    def cblinear(self, x, p, splits):  # :222-228 — bare 1x1 conv, then channel split
        return self._conv2d(x, p + ".conv").split(splits, 1)

    @staticmethod
    def cbfuse(parts, last):  # :230-245 — nearest-resize the selected splits to the last input's size and sum
        th, tw = last.shape[2:]
        acc = None
        for t in parts:
            f = th // t.shape[2]
            u = t.repeat_interleave(f, dim=2).repeat_interleave(tw // t.shape[3], dim=3) if f > 1 else t
            acc = u if acc is None else acc + u
        return acc + last                       # Tensor.stack(*res).sum(0): left-to-right sum, xs[-1] last

    def features_e(self, x: torch.Tensor) -> List[torch.Tensor]:
        """Blocks 0..41 of the "e" graph (:328-370); returns outputs of blocks 35, 38, 41."""
        from clearcam_amd.weights import YOLO_E_CBLINEAR
        P = "model.list."
        y: Dict[int, torch.Tensor] = {0: x}
        y[1] = self.conv(x, P + "1", stride=2)
        y[2] = self.conv(y[1], P + "2", stride=2)
        y[3] = self.elan4(y[2], P + "3"); y[4] = self.adown(y[3], P + "4")
        y[5] = self.elan4(y[4], P + "5"); y[6] = self.adown(y[5], P + "6")
        y[7] = self.elan4(y[6], P + "7"); y[8] = self.adown(y[7], P + "8")
        y[9] = self.elan4(y[8], P + "9")
        cb = {i: self.cblinear(y[src], P + str(i), YOLO_E_CBLINEAR[i][1]) for i, src in ((10, 1), (11, 3), (12, 5), (13, 7), (14, 9))}
        y[15] = self.conv(x, P + "15", stride=2)
        y[16] = self.cbfuse([cb[i][0] for i in (10, 11, 12, 13, 14)], y[15])
        y[17] = self.conv(y[16], P + "17", stride=2)
        y[18] = self.cbfuse([cb[i][1] for i in (11, 12, 13, 14)], y[17])
        y[19] = self.elan4(y[18], P + "19"); y[20] = self.adown(y[19], P + "20")
        y[21] = self.cbfuse([cb[i][2] for i in (12, 13, 14)], y[20])
        y[22] = self.elan4(y[21], P + "22"); y[23] = self.adown(y[22], P + "23")
        y[24] = self.cbfuse([cb[i][3] for i in (13, 14)], y[23])
        y[25] = self.elan4(y[24], P + "25"); y[26] = self.adown(y[25], P + "26")
        y[27] = self.cbfuse([cb[14][4]], y[26])
        y[28] = self.elan4(y[27], P + "28")
        y[29] = self.sppelan(y[28], P + "29")
        y[32] = self.elan4(torch.cat((self.upsample(y[29]), y[25]), 1), P + "32")
        y[35] = self.elan4(torch.cat((self.upsample(y[32]), y[22]), 1), P + "35")
        y[38] = self.elan4(torch.cat((self.adown(y[35], P + "36"), y[32]), 1), P + "38")
        y[41] = self.elan4(torch.cat((self.adown(y[38], P + "39"), y[29]), 1), P + "41")
        self.block_outputs = y
        return [y[35], y[38], y[41]]

    # -- backbone + neck ----------------------------------------------------------------------
    def features(self, x: torch.Tensor) -> List[torch.Tensor]:
        """Blocks 0..21 (:304-325); returns [P3, P4, P5] = outputs of blocks 15, 18, 21."""
        if self.size == "e":
            return self.features_e(x)
        P = "model.list."
        y: Dict[int, torch.Tensor] = {}
        y[0] = self.conv(x, P + "0", stride=2)
        y[1] = self.conv(y[0], P + "1", stride=2)
        y[2] = self.elan1(y[1], P + "2") if self.arch.b2_kind == "elan1" else self.elan4(y[1], P + "2")
        y[3] = self.down(y[2], P + "3")
        y[4] = self.elan4(y[3], P + "4")
        y[5] = self.down(y[4], P + "5")
        y[6] = self.elan4(y[5], P + "6")
        y[7] = self.down(y[6], P + "7")
        y[8] = self.elan4(y[7], P + "8")
        y[9] = self.sppelan(y[8], P + "9")
        y[12] = self.elan4(torch.cat((self.upsample(y[9]), y[6]), 1), P + "12")
        y[15] = self.elan4(torch.cat((self.upsample(y[12]), y[4]), 1), P + "15")
        y[18] = self.elan4(torch.cat((self.down(y[15], P + "16"), y[12]), 1), P + "18")
        y[21] = self.elan4(torch.cat((self.down(y[18], P + "19"), y[9]), 1), P + "21")
        self.block_outputs = y
        return [y[15], y[18], y[21]]

    # -- head ---------------------------------------------------------------------------------
    def head_raw(self, feats: Sequence[torch.Tensor]) -> List[torch.Tensor]:
        """``DDetect`` branches (:202-207): per level (B,144,H,W) = cat(box 64, cls 80)."""
        H = self.head
        out = []
        for i, x in enumerate(feats):
            b = f"{H}cv2.list.{i}.list."
            c = f"{H}cv3.list.{i}.list."
            x0 = self._conv2d(self.conv(self.conv(x, b + "0"), b + "1", groups=4), b + "2", groups=4)
            x1 = self._conv2d(self.conv(self.conv(x, c + "0"), c + "1"), c + "2")
            out.append(torch.cat((x0, x1), 1))
        return out

    def decode(self, raw: Sequence[torch.Tensor]) -> torch.Tensor:
        """``DDetect`` decode (:209-220), ``make_anchors`` :247-261, ``DFL`` :273-282,
        ``dist2bbox`` :263-271 -> (B, 84, A)."""
        B = raw[0].shape[0]
        anchors, strides = [], []
        for x, s in zip(raw, (8, 16, 32)):
            h, w = x.shape[2:]
            sx = (torch.arange(w, dtype=torch.float32) + 0.5).reshape(1, -1).repeat(h, 1).reshape(-1)
            sy = (torch.arange(h, dtype=torch.float32) + 0.5).reshape(-1, 1).repeat(1, w).reshape(-1)
            anchors.append(torch.stack((sx, sy), -1))
            strides.append(torch.full((h * w,), float(s)))
        anchors = torch.cat(anchors).t().unsqueeze(0)      # (1, 2, A)
        strides = torch.cat(strides).unsqueeze(0)          # (1, A)
        cat = torch.cat([x.reshape(B, 144, -1) for x in raw], 2)
        box, cls = cat.split((64, 80), 1)
        a = box.shape[2]
        prob = box.reshape(B, 4, 16, a).transpose(2, 1).softmax(1)
        dist = F.conv2d(prob, self.sd[self.head + "dfl.conv.weight"]).reshape(B, 4, a)
        lt, rb = dist.chunk(2, 1)
        x1y1, x2y2 = anchors - lt, anchors + rb
        dbox = torch.cat(((x1y1 + x2y2) / 2, x2y2 - x1y1), 1) * strides
        return torch.cat((dbox, torch.sigmoid(cls)), 1)

    # -- postprocess --------------------------------------------------------------------------
    @staticmethod
    def postprocess(output: torch.Tensor, max_det=MAX_DET, conf=CONF_THRESHOLD, iou_thr=IOU_THRESHOLD) -> torch.Tensor:
        """``postprocess`` :439-458 with ``compute_iou_matrix`` :423-437.  (B,84,A) -> (B,300,6).

        top-k is *stable* descending (ties -> lower anchor index first), which is how tinygrad's
        sort breaks ties (Appendix B-7 calls the order implementation-defined)."""
        xc, yc, w, h, scores = output[:, 0], output[:, 1], output[:, 2], output[:, 3], output[:, 4:]
        x1, y1, x2, y2 = xc - w / 2, yc - h / 2, xc + w / 2, yc + h / 2
        cls = scores.argmax(1)
        probs = scores.max(1).values
        probs = torch.where(probs >= conf, probs, torch.zeros_like(probs))
        boxes = torch.stack((x1, y1, x2, y2, probs, cls.float()), 2)
        order = torch.sort(probs, dim=1, descending=True, stable=True).indices[:, :max_det]
        boxes = torch.gather(boxes, 1, order.unsqueeze(-1).expand(-1, -1, 6))
        bx1, by1, bx2, by2 = boxes[..., 0], boxes[..., 1], boxes[..., 2], boxes[..., 3]
        areas = (bx2 - bx1) * (by2 - by1)
        ix1 = torch.maximum(bx1[:, :, None], bx1[:, None, :])
        iy1 = torch.maximum(by1[:, :, None], by1[:, None, :])
        ix2 = torch.minimum(bx2[:, :, None], bx2[:, None, :])
        iy2 = torch.minimum(by2[:, :, None], by2[:, None, :])
        inter = (ix2 - ix1).clamp_min(0) * (iy2 - iy1).clamp_min(0)
        ious = inter / (areas[:, :, None] + areas[:, None, :] - inter)
        ious = torch.triu(ious, diagonal=1)
        same = boxes[..., 5][:, :, None] == boxes[..., 5][:, None, :]
        keep = ((ious > iou_thr) & same).sum(1) == 0
        return boxes * keep.unsqueeze(-1)

    @staticmethod
    def scale_boxes(lb_hw: Tuple[int, int], preds: torch.Tensor, src_hw: Tuple[int, int]) -> torch.Tensor:
        """``scale_boxes``/``clip_boxes`` :406-421 (applied to zero rows too)."""
        gain = min(lb_hw[0] / src_hw[0], lb_hw[1] / src_hw[1])
        pad_x = (lb_hw[1] - src_hw[1] * gain) / 2
        pad_y = (lb_hw[0] - src_hw[0] * gain) / 2
        p = preds.clone()
        p[..., [0, 2]] = ((p[..., [0, 2]] - np.float32(pad_x)) / np.float32(gain)).clamp(0, src_hw[1])
        p[..., [1, 3]] = ((p[..., [1, 3]] - np.float32(pad_y)) / np.float32(gain)).clamp(0, src_hw[0])
        return p

    # -- call surfaces ------------------------------------------------------------------------
    def network_input(self, frames: np.ndarray) -> torch.Tensor:
        """(B,H,W,3) BGR u8/f32 -> letterboxed RGB NCHW /255 (:376-379)."""
        lb = np.stack([letterbox(f, self.res) for f in frames])
        x = torch.from_numpy(np.ascontiguousarray(lb[..., ::-1])).permute(0, 3, 1, 2)
        return x.to(torch.float32) / 255.0

    @torch.no_grad()
    def detect_batch(self, frames: np.ndarray) -> np.ndarray:
        """Batch extension of ``__call__``: (B,H,W,3) -> (B,300,6); B=1 is the reference."""
        x = self.network_input(frames)
        y = self.decode(self.head_raw(self.features(x)))
        preds = self.postprocess(y)
        return self.scale_boxes(tuple(x.shape[2:]), preds, frames.shape[1:3]).numpy()

    def __call__(self, frame: np.ndarray) -> np.ndarray:
        """``YOLOv9.__call__`` :375-388: (H,W,3) BGR -> (300,6) [x1,y1,x2,y2,conf,cls]."""
        return self.detect_batch(frame[None])[0]


def decoded_rows(y: torch.Tensor, conf: float = CONF_THRESHOLD) -> np.ndarray:
    """(B,84,A) decode output -> (B,A,6) [x1,y1,x2,y2,score(thresholded),cls] = the rows ``postprocess``
    ranks (:440-448), before top-k; the continuous quantity the HIP ``decoded`` tap is compared with."""
    xc, yc, w, h, scores = y[:, 0], y[:, 1], y[:, 2], y[:, 3], y[:, 4:]
    probs, cls = scores.max(1)
    probs = torch.where(probs >= conf, probs, torch.zeros_like(probs))
    return torch.stack((xc - w / 2, yc - h / 2, xc + w / 2, yc + h / 2, probs, cls.float()), 2).numpy()


def match_detections(ref: np.ndarray, got: np.ndarray, iou_thr: float = 0.9):
    """Parity metric: rows with score>0 matched greedily by class + IoU.

    Returns (n_ref, n_got, n_matched, max_abs_box_err, max_abs_score_err) over matched pairs."""
    r = ref[ref[:, 4] > 0]
    g = got[got[:, 4] > 0]
    used = np.zeros(len(g), bool)
    n, box_err, sc_err = 0, 0.0, 0.0
    for row in r:
        best, bj = -1.0, -1
        for j, c in enumerate(g):
            if used[j] or int(c[5]) != int(row[5]):
                continue
            ix = max(0.0, min(row[2], c[2]) - max(row[0], c[0]))
            iy = max(0.0, min(row[3], c[3]) - max(row[1], c[1]))
            inter = ix * iy
            u = (row[2] - row[0]) * (row[3] - row[1]) + (c[2] - c[0]) * (c[3] - c[1]) - inter
            iou = inter / u if u > 0 else (1.0 if np.allclose(row[:4], c[:4]) else 0.0)
            if iou > best:
                best, bj = iou, j
        if bj >= 0 and best >= iou_thr:
            used[bj] = True
            n += 1
            box_err = max(box_err, float(np.abs(row[:4] - g[bj][:4]).max()))
            sc_err = max(sc_err, float(abs(row[4] - g[bj][4])))
    return len(r), len(g), n, box_err, sc_err


def match_detections_strict(ref: np.ndarray, got: np.ndarray, box_tol: float, iou_thr: float = 0.9, score_margin: float = 0.0,
                            conf: float = CONF_THRESHOLD):
    """The yardstick of the f32 gate applied pair by pair: a detection is matched only if a detection of the same class with
    IoU >= iou_thr exists on the other side AND its four coordinates lie within `box_tol` pixels (north_star's 1e-3 read as
    1e-3 * max(H, W)).  A pair that clears the IoU bar with a larger coordinate error is a different anchor that survived NMS
    among heavily overlapping candidates - a selection flip, counted as unmatched, not as a small error.

    score_margin: an UNMATCHED detection whose own score lies within score_margin of the confidence threshold is reported as
    `borderline`: the reference's `where(p >= 0.25, p, 0)` (detection/yolov9.py:446) is discontinuous there, and a score error
    inside the mode's score tolerance legitimately makes such a row appear or disappear.

    Returns a dict: n_ref, n_got, n_iou (pairs at IoU >= thr), n_strict (pairs also within box_tol), borderline_ref / borderline_got,
    box_err (sorted array of the coordinate errors of the IoU pairs), score_err_max (over strict pairs)."""
    r = ref[ref[:, 4] > 0]
    g = got[got[:, 4] > 0]
    used = np.zeros(len(g), bool)
    hit = np.zeros(len(r), bool)
    errs, n_strict, sc = [], 0, 0.0
    for ri, row in enumerate(r):
        cand = np.nonzero(~used & (g[:, 5].astype(np.int64) == int(row[5])))[0]
        if not len(cand):
            continue
        c = g[cand]
        ix = np.clip(np.minimum(row[2], c[:, 2]) - np.maximum(row[0], c[:, 0]), 0, None)
        iy = np.clip(np.minimum(row[3], c[:, 3]) - np.maximum(row[1], c[:, 1]), 0, None)
        inter = ix * iy
        u = (row[2] - row[0]) * (row[3] - row[1]) + (c[:, 2] - c[:, 0]) * (c[:, 3] - c[:, 1]) - inter
        iou = np.where(u > 0, inter / np.where(u > 0, u, 1), 0.0)
        j = int(np.argmax(iou))
        if iou[j] >= iou_thr:
            used[cand[j]] = True
            hit[ri] = True
            e = float(np.abs(row[:4] - c[j, :4]).max())
            errs.append(e)
            if e <= box_tol:
                n_strict += 1
                sc = max(sc, float(abs(row[4] - c[j, 4])))
    return {"n_ref": len(r), "n_got": len(g), "n_iou": len(errs), "n_strict": n_strict,
            "borderline_ref": int((~hit & (r[:, 4] < conf + score_margin)).sum()), "borderline_got": int((~used & (g[:, 4] < conf + score_margin)).sum()),
            "box_err": np.sort(np.asarray(errs, np.float64)), "score_err_max": sc}


def parity_summary(refs: np.ndarray, gots: np.ndarray, box_tol: float, dec_ref: np.ndarray = None, dec_got: np.ndarray = None,
                   score_margin: float = 2e-3) -> dict:
    """match_detections_strict over a batch, plus (when the per-anchor decoded rows of both sides are given) the continuous
    quantity behind it: coordinate error anchor by anchor over the anchors both sides score over the threshold.
    match_frac = strict matches / max(n_ref, n_got); match_frac_clear_of_threshold leaves out of the denominator the unmatched
    rows whose own score is within `score_margin` of the 0.25 threshold (see match_detections_strict)."""
    tot = {"n_ref": 0, "n_got": 0, "n_iou": 0, "n_strict": 0, "borderline_ref": 0, "borderline_got": 0}
    errs, sc = [], 0.0
    for a, b in zip(refs, gots):
        m = match_detections_strict(a, b, box_tol, score_margin=score_margin)
        for k in tot:
            tot[k] += m[k]
        errs.append(m["box_err"]); sc = max(sc, m["score_err_max"])
    errs = np.sort(np.concatenate(errs)) if errs else np.zeros(0)
    den = max(tot["n_ref"], tot["n_got"], 1)
    den2 = max(tot["n_ref"] - tot["borderline_ref"], tot["n_got"] - tot["borderline_got"], 1)
    out = dict(tot, match_frac=tot["n_strict"] / den, match_frac_clear_of_threshold=min(1.0, tot["n_strict"] / den2), score_margin=score_margin,
               match_frac_iou_only=tot["n_iou"] / den, box_tol_px=box_tol, score_err_max=sc,
               box_err_px_p50=float(np.quantile(errs, 0.5)) if len(errs) else 0.0, box_err_px_p99=float(np.quantile(errs, 0.99)) if len(errs) else 0.0,
               box_err_px_max_strict=float(errs[errs <= box_tol].max()) if (errs <= box_tol).any() else 0.0,
               box_err_px_max_iou_pairs=float(errs.max()) if len(errs) else 0.0)
    if dec_ref is not None and dec_got is not None:
        both = (dec_ref[..., 4] > 0) & (dec_got[..., 4] > 0)
        ae = np.abs(dec_ref[..., :4] - dec_got[..., :4]).max(-1)[both]
        out.update(anchors_both_over_thr=int(both.sum()), anchor_box_err_px_p50=float(np.quantile(ae, 0.5)) if len(ae) else 0.0,
                   anchor_box_err_px_p99=float(np.quantile(ae, 0.99)) if len(ae) else 0.0, anchor_box_err_px_p999=float(np.quantile(ae, 0.999)) if len(ae) else 0.0,
                   anchor_box_err_px_max=float(ae.max()) if len(ae) else 0.0, anchors_over_tol=int((ae > box_tol).sum()),
                   anchor_score_err_max=float(np.abs(dec_ref[..., 4] - dec_got[..., 4])[both].max()) if both.any() else 0.0)
    return out


def tolerance_bars(s: dict) -> dict:
    """The bars a 16-bit TOLERANCE mode (f16 activations, weights exact or nearly exact: dtypes "f16s" / "f16h") is held to against this
    f32 oracle on a checkpoint whose float32 weights are not pre-rounded, from a ``parity_summary`` with the decoded rows given.  The
    tolerance is the north star's 1e-3 x max(H, W) px (``box_tol``), applied at three levels:

      detections   >= 98.5 % of the rows strictly matched (same class, IoU >= 0.9, all four coordinates within the tolerance), rows whose
                   own score sits within 2e-3 of the 0.25 threshold left out (>= 97.5 % with every row counted); scores within 2e-3
      anchors      99.9 % of the anchors both sides score over the threshold within the tolerance
      tail         no anchor beyond 1.5 x the tolerance.  f16 ACTIVATION rounding has a tail: rounding flips cascade (one flipped element
                   flips ~50 downstream), so two correct implementations are independent noise realisations within three layers, and an
                   ill-conditioned P5 region of one frame can read 0.6 px where the typical worst anchor of a frame set reads 0.2-0.35
                   ("The tail"); the modes with rounded weights (plain f16 / bf16) reach 0.8-1.2 / 8 px.
    """
    tol = s["box_tol_px"]
    bars = {"detections": s["match_frac_clear_of_threshold"] >= 0.985 and s["match_frac"] >= 0.975 and s["anchor_score_err_max"] <= 2e-3,
            "anchors": s["anchor_box_err_px_p999"] <= tol, "tail": s["anchor_box_err_px_max"] <= 1.5 * tol}
    bars["all"] = all(bars.values())
    return bars

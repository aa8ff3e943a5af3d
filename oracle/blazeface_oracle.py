"""CPU oracle for the BlazeFace face detector, `BlazeFace.__call__(img)` (models/blazeface.py:165-192).

TEST INFRASTRUCTURE ONLY (tests/ import it; nothing under clearcam_amd/ does).  PyTorch-CPU fp32 + numpy restatement.
PIN STATUS: tinygrad (pinned fe39cf14) is not available offline and models/blazeface.safetensors is a missing large
blob of the reference checkout; the reference keeps no face-detection fixture ("parity unpinned" against trained weights and
tinygrad's kernels).  What pins this file: models/blazeface.py itself, executed unchanged over a PyTorch stand-in for tinygrad
(tools/refshim) on the seeded checkpoint, three image shapes - tests/test_reference_run.py, identical rows, |diff| 0.0.  (That
run is what caught an earlier misreading of the overlap rule's axis.)  Restated from the file as written:
  preprocess (:166-179)  scale = min(256/w, 256/h); new = int(w*scale), int(h*scale) (truncation); helpers.resize (tinygrad
                         bilinear, align_corners=False; uint8 input uses tinygrad's 7-bit fixed-point lerp like the detector's
                         letterbox); centred zero pad to 256x256; x/127.5 - 1 (the padding becomes -1); no channel flip
  forward (:139-163)     pad (1,2) + conv 5x5 s2 (3->24) + ReLU; 31 BlazeBlocks (:4-30): depthwise 3x3 (+bias) -> 1x1 (+bias),
                         residual add (stride 2: input zero-padded bottom/right by 2, shortcut max-pooled 2x2 and, where the
                         width grows, zero-padded in channels), then ReLU; FinalBlazeBlock (:42-59); four 1x1 heads;
                         outputs flattened in (H, W, anchor) order: 512 anchors of the 16x16 map then 384 of the 8x8 map
  decode (:204-226)      boxes / keypoints relative to the 896 anchors (scale 256), score = sigmoid(clip(raw, +-100))
  postprocess (:228-238) rows below 0.85 zeroed; sort by score descending (stable); a row survives iff it is >= 0.85 and
                         is overlapped by NO BETTER-RANKED row by IoU > 0.3 (triu of the (1,N,N) mask summed over axis 1); others zeroed
  back-map (:188-192)    * 256; columns 0,2 -= pad_top; columns 1,3 -= pad_left; ALL 17 columns /= scale (as written)
"""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F

from .yolov9_oracle import resize_bilinear

# (cin, cout, stride) of backbone_tiny[0..30] (models/blazeface.py:88-119)
BLOCKS = [(24, 24, 1)] * 7 + [(24, 24, 2)] + [(24, 24, 1)] * 7 + [(24, 48, 2)] + [(48, 48, 1)] * 7 + [(48, 96, 2)] + [(96, 96, 1)] * 7


class BlazeFaceOracle:
    def __init__(self, state_dict: Dict[str, np.ndarray]):
        self.sd = {k: torch.from_numpy(np.asarray(v, np.float32)) for k, v in state_dict.items()}

    def _conv(self, x, name, stride=1, padding=0, groups=1):
        return F.conv2d(x, self.sd[name + ".weight"], self.sd[name + ".bias"], stride=stride, padding=padding, groups=groups)

    def _block(self, x, i, cin, cout, stride):
        p = f"backbone_tiny.list.{i}."
        if stride == 2:
            h = F.pad(x, (0, 2, 0, 2))
            x = F.max_pool2d(x, 2, 2)
            h = self._conv(h, p + "conv0_tiny", stride=2, groups=cin)
        else:
            h = self._conv(x, p + "conv0_tiny", padding=1, groups=cin)
        if cout > cin:
            x = F.pad(x, (0, 0, 0, 0, 0, cout - cin))
        h = self._conv(h, p + "conv1_tiny")
        return F.relu(x + h)

    def network_input(self, img: np.ndarray):
        h0, w0 = img.shape[:2]
        scale = min(256 / w0, 256 / h0)
        new_w, new_h = int(w0 * scale), int(h0 * scale)
        r = resize_bilinear(np.asarray(img), new_h, new_w)
        pad_top, pad_left = (256 - new_h) // 2, (256 - new_w) // 2
        x = np.zeros((256, 256, 3), r.dtype)
        x[pad_top:pad_top + new_h, pad_left:pad_left + new_w] = r
        x = torch.from_numpy(x.astype(np.float32)).permute(2, 0, 1).unsqueeze(0) / 127.5 - 1.0
        return x, scale, pad_top, pad_left

    def forward(self, x):
        x = F.relu(self._conv(F.pad(x, (1, 2, 1, 2)), "conv_tiny", stride=2))
        for i, (cin, cout, stride) in enumerate(BLOCKS):
            x = self._block(x, i, cin, cout, stride)
        h = F.pad(x, (0, 2, 0, 2))
        h = self._conv(h, "final.conv0_tiny", stride=2, groups=96)
        h = F.relu(self._conv(h, "final.conv1_tiny"))
        b = x.shape[0]
        c = torch.cat([self._conv(x, "classifier_8_tiny").permute(0, 2, 3, 1).reshape(b, -1, 1),
                       self._conv(h, "classifier_16_tiny").permute(0, 2, 3, 1).reshape(b, -1, 1)], 1)
        r = torch.cat([self._conv(x, "regressor_8_tiny").permute(0, 2, 3, 1).reshape(b, -1, 16),
                       self._conv(h, "regressor_16_tiny").permute(0, 2, 3, 1).reshape(b, -1, 16)], 1)
        return r, c

    def decode(self, raw, score_raw):
        a = self.sd["anchors"]
        ax, ay, aw, ah = a[:, 0], a[:, 1], a[:, 2], a[:, 3]
        out = torch.zeros(896, 17)
        xc = raw[:, 0] / 256.0 * aw + ax
        yc = raw[:, 1] / 256.0 * ah + ay
        w = raw[:, 2] / 256.0 * aw
        h = raw[:, 3] / 256.0 * ah
        out[:, 0], out[:, 1], out[:, 2], out[:, 3] = yc - h / 2.0, xc - w / 2.0, yc + h / 2.0, xc + w / 2.0
        for k in range(6):
            out[:, 4 + 2 * k] = raw[:, 4 + 2 * k] / 256.0 * aw + ax
            out[:, 5 + 2 * k] = raw[:, 5 + 2 * k] / 256.0 * ah + ay
        s = torch.sigmoid(score_raw.clamp(-100.0, 100.0))
        out[:, 16] = s
        return out * (s >= 0.85).float()[:, None]

    @staticmethod
    def postprocess(det: torch.Tensor) -> torch.Tensor:
        order = torch.argsort(det[:, 16], descending=True, stable=True)
        b = det[order]
        x1, y1, x2, y2 = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
        area = (x2 - x1) * (y2 - y1)
        w = torch.clamp(torch.minimum(x2[:, None], x2[None, :]) - torch.maximum(x1[:, None], x1[None, :]), min=0)
        h = torch.clamp(torch.minimum(y2[:, None], y2[None, :]) - torch.maximum(y1[:, None], y1[None, :]), min=0)
        inter = w * h
        iou = torch.triu(inter / (area[:, None] + area[None, :] - inter), diagonal=1)      # NaN (0/0) compares False
        # the reference sums the (1,N,N) mask over axis=1, i.e. over the HIGHER-ranked row index i of triu's (i < j) pairs:
        # row j is dropped when any better-scored row overlaps it (models/blazeface.py:232-234)
        keep = ((iou > 0.3).sum(0) == 0) & (b[:, 16] >= 0.85)
        return b * keep.float()[:, None]

    @torch.no_grad()
    def raw(self, img: np.ndarray):
        x, scale, pt, pl = self.network_input(img)
        r, c = self.forward(x)
        return r[0], c[0, :, 0]

    @torch.no_grad()
    def __call__(self, img: np.ndarray) -> np.ndarray:
        """(H,W,3) uint8/float image -> (896,17) float32 [ymin,xmin,ymax,xmax, 6 x (kx,ky), score] in source pixels."""
        x, scale, pad_top, pad_left = self.network_input(img)
        r, c = self.forward(x)
        det = self.postprocess(self.decode(r[0], c[0, :, 0])) * 256
        det[:, [0, 2]] -= pad_top
        det[:, [1, 3]] -= pad_left
        return (det / scale).numpy()

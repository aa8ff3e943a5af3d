"""CPU oracle for the AdaFace IR-50 face embedder, `ADAFACE.__call__` (models/adaface.py:61-95).

TEST INFRASTRUCTURE ONLY: imported by tests/ (and nothing under clearcam_amd/).  PyTorch-CPU fp32 restatement.
PIN STATUS: tinygrad (pinned fe39cf14, not vendored) and the adaface_ir50_ms1mv2 checkpoint the reference downloads
(adaface.py:77) are not available offline and the reference keeps no face-embedding fixture ("parity unpinned" against trained
weights and tinygrad's kernels).  What pins this file: models/adaface.py itself, executed unchanged over a PyTorch stand-in for
tinygrad (tools/refshim) on the seeded checkpoint - tests/test_reference_run.py, |diff| <= 5e-6 (measured 2e-7).  Restated: the
graph as written plus tinygrad's documented inference-mode BatchNorm ((x - running_mean) * rsqrt(running_var + 1e-5) * w + b).

Graph (adaface.py): x (112,112,3) BGR -> [:,:,::-1] -> /255 -> (x-0.5)/0.5 -> CHW -> conv0 3x3 (3->64, no bias) -> bn0 ->
PReLU(64) -> 24 x BasicBlockIR(in, depth, stride) [:23-54, table :58] -> bn (512) -> flatten (C,H,W order) -> linear
25088->512 -> bn2 (no affine) -> x / sqrt(sum(x*x)).
BasicBlockIR: shortcut = x[:, :, ::s, ::s] if in == depth else bn(conv1x1 stride s);
              r = bn0(x) -> conv3x3 s1 (no bias) -> bn1 -> PReLU(depth) -> conv3x3 stride s (no bias) -> bn2;  out = r + shortcut.
"""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F

EPS = 1e-5
# (in_channel, depth, stride) of the 24 blocks (models/adaface.py:58)
BLOCKS = [(64, 64, 2), (64, 64, 1), (64, 64, 1), (64, 128, 2), (128, 128, 1), (128, 128, 1), (128, 128, 1), (128, 256, 2)] + \
         [(256, 256, 1)] * 13 + [(256, 512, 2), (512, 512, 1), (512, 512, 1)]


class AdaFaceOracle:
    def __init__(self, state_dict: Dict[str, np.ndarray]):
        self.sd = {k: torch.from_numpy(np.asarray(v, np.float32)) for k, v in state_dict.items() if not k.endswith("num_batches_tracked")}

    def _bn(self, x, p, affine=True):
        sd = self.sd
        shape = (1, -1, 1, 1) if x.ndim == 4 else (1, -1)
        y = (x - sd[p + ".running_mean"].view(shape)) * torch.rsqrt(sd[p + ".running_var"].view(shape) + EPS)
        return y * sd[p + ".weight"].view(shape) + sd[p + ".bias"].view(shape) if affine else y

    @staticmethod
    def _prelu(x, w):
        return torch.where(x > 0, x, w.view(1, -1, 1, 1) * x)

    def _block(self, x, i, cin, depth, stride):
        sd, p = self.sd, f"body.list.{i}."
        if cin == depth:
            sc = x[:, :, ::stride, ::stride]                          # MaxPool2d(1, stride)
        else:
            sc = self._bn(F.conv2d(x, sd[p + "shortcut_layer0.weight"], stride=stride), p + "shortcut_layer1")
        r = self._bn(x, p + "res_layer0")
        r = self._bn(F.conv2d(r, sd[p + "conv_layer0.weight"], padding=1), p + "res_layer1")
        r = self._prelu(r, sd[p + "prelu_weight"])
        r = self._bn(F.conv2d(r, sd[p + "conv_layer1.weight"], stride=stride, padding=1), p + "res_layer2")
        return r + sc

    def network_input(self, img: np.ndarray) -> torch.Tensor:
        x = torch.from_numpy(np.ascontiguousarray(np.asarray(img)[:, :, ::-1].astype(np.float32)))
        x = ((x / 255.0) - 0.5) / 0.5
        return x.permute(2, 0, 1).unsqueeze(0)

    def features(self, img: np.ndarray, upto: int = 24) -> torch.Tensor:
        sd = self.sd
        x = self.network_input(img)
        x = self._prelu(self._bn(F.conv2d(x, sd["conv0.weight"], padding=1), "bn0"), sd["prelu_weight"])
        for i, (cin, depth, stride) in enumerate(BLOCKS[:upto]):
            x = self._block(x, i, cin, depth, stride)
        return x

    @torch.no_grad()
    def __call__(self, img: np.ndarray) -> np.ndarray:
        """(112,112,3) uint8/float BGR -> (1,512) float32, unit norm."""
        sd = self.sd
        x = self._bn(self.features(img), "bn")
        x = x.reshape(1, -1) @ sd["linear.weight"].t() + sd["linear.bias"]
        x = self._bn(x, "bn2", affine=False)
        return (x / torch.sqrt(torch.sum(x * x))).numpy()

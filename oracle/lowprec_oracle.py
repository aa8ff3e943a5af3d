"""ORACLE (test infrastructure, not product code): the fp32 detector oracle with 16-bit STORAGE rounding.

``YOLOv9Oracle`` (oracle/yolov9_oracle.py) restates the reference in float32.  The HIP speed modes
(dtype f16 / bf16) keep the reference's arithmetic but store every activation and every weight in a
16-bit type: MFMA products of 16-bit operands accumulate in f32, the epilogue (bias, SiLU, residual
add) runs in f32 and the result is rounded ONCE when it is written (clearcam_amd/csrc/conv_mfma.hip
``conv_epilogue``).  This class applies exactly those roundings to the oracle:

  * every conv weight            -> storage type (bias stays f32); ``feedback=True`` = the library's default rounding (yolo.hip
                                    ``round_controlled``): each weight goes to one of its two neighbours in the storage type so
                                    that the output channel's total, each input channel's tap-sum and each tap's channel-sum of
                                    the rounding residuals stay near zero
  * the network input  x/255     -> storage type (detect.hip ``stem_fused_kernel`` / ``preprocess_kernel``)
  * every ``Conv`` output        -> storage type after SiLU; a RepNBottleneck's ``x + cv2(cv1(x))``
                                    (detection/yolov9.py:89) is rounded once after the add
  * ADown/AConv's 2x2 average    -> storage type (conv_direct.hip ``pool_vec_kernel``; max-pooling picks
                                    an already-rounded value, so it adds nothing)
  * the head's last 1x1 convs    -> f32 (the ``raw`` tensors are f32), decode / top-k / NMS in f32

It is NOT a reference restatement and pins nothing about the reference; it predicts what a correct
16-bit implementation can differ from the f32 oracle by, so that (a) tools/calibrate_synth.py can
measure whether a synthetic checkpoint is well enough conditioned to test the speed modes end to end
and (b) tests can separate "rounding the design allows" from "a kernel is wrong".  Only ``tests/`` and
``tools/`` import it.
"""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F

from oracle.yolov9_oracle import YOLOv9Oracle

_TORCH_T = {"f16": torch.float16, "bf16": torch.bfloat16}


class LowPrecOracle(YOLOv9Oracle):
    def __init__(self, size: str, res: int, state_dict: Dict[str, np.ndarray], dtype: str, feedback: bool = False):
        super().__init__(size, res, state_dict)
        self.t = _TORCH_T[dtype]
        for k in list(self.sd):
            if k.endswith(".weight") and self.sd[k].ndim == 4 and "dfl" not in k:
                self.sd[k] = self.q_feedback(self.sd[k]) if feedback else self.q(self.sd[k])

    def q_feedback(self, w: torch.Tensor) -> torch.Tensor:
        """clearcam_amd/csrc/yolo.hip ``round_controlled``: the same float32 operations in the same order (two passes per output channel)."""
        co, ci, k = w.shape[0], w.shape[1], w.shape[2]
        taps = k * k
        wf = w.reshape(co, ci, taps).to(torch.float32)
        r = self.q(wf)
        inf = torch.tensor(float("inf"), dtype=self.t)
        lo = torch.where(r > wf, torch.nextafter(r.to(self.t), -inf).to(torch.float32), r)
        hi = torch.where(r < wf, torch.nextafter(r.to(self.t), inf).to(torch.float32), r)
        rr = [float(t // k) - (k - 1) / 2.0 for t in range(taps)]
        ss = [float(t % k) - (k - 1) / 2.0 for t in range(taps)]
        mom = taps > 1
        z = lambda *s: torch.zeros(*s, dtype=torch.float32)      # noqa: E731
        q, d = torch.empty_like(wf), torch.empty_like(wf)
        e_row, e_col, e_tot, m_r, m_s = z(co, ci), z(co, taps), z(co), z(co), z(co)

        def cost(er, ec, dd, t):
            v = (er + dd) * (er + dd) + (ec + dd) * (ec + dd) + 2.0 * ((e_tot + dd) * (e_tot + dd))
            if mom:
                v = v + ((m_r + dd * rr[t]) * (m_r + dd * rr[t]) + (m_s + dd * ss[t]) * (m_s + dd * ss[t]))
            return v

        for c in range(ci):                                      # pass 1: sequential, running sums
            er = z(co)
            for t in range(taps):
                dl, dh = lo[:, c, t] - wf[:, c, t], hi[:, c, t] - wf[:, c, t]
                up = cost(er, e_col[:, t], dh, t) < cost(er, e_col[:, t], dl, t)
                dd = torch.where(up, dh, dl)
                q[:, c, t] = torch.where(up, hi[:, c, t], lo[:, c, t])
                d[:, c, t] = dd
                er = er + dd
                e_col[:, t] = e_col[:, t] + dd
                e_tot = e_tot + dd
                if mom:
                    m_r, m_s = m_r + dd * rr[t], m_s + dd * ss[t]
            e_row[:, c] = er
        for c in range(ci):                                      # pass 2: every weight again, all others fixed
            for t in range(taps):
                dcur = d[:, c, t].clone()
                dl, dh = (lo[:, c, t] - wf[:, c, t]) - dcur, (hi[:, c, t] - wf[:, c, t]) - dcur
                up = cost(e_row[:, c], e_col[:, t], dh, t) < cost(e_row[:, c], e_col[:, t], dl, t)
                de = torch.where(up, dh, dl)
                q[:, c, t] = torch.where(up, hi[:, c, t], lo[:, c, t])
                d[:, c, t] = dcur + de
                e_row[:, c] = e_row[:, c] + de
                e_col[:, t] = e_col[:, t] + de
                e_tot = e_tot + de
                if mom:
                    m_r, m_s = m_r + de * rr[t], m_s + de * ss[t]
        return q.reshape(w.shape)

    def q(self, x: torch.Tensor) -> torch.Tensor:
        return x.to(self.t).to(torch.float32)

    def conv(self, x, name, stride=1, groups=1):
        return self.q(super().conv(x, name, stride, groups))

    def _silu_conv_f32(self, x, name):
        return F.silu(self._conv2d(x, name + ".conv"))

    def repncsp(self, x, p):  # detection/yolov9.py:82-105; the residual add is the second conv's epilogue
        x2 = self.conv(x, p + ".cv1")
        for j in range(self.rep_n):
            q = f"{p}.m.list.{j}"
            x2 = self.q(x2 + self._silu_conv_f32(self.conv(x2, q + ".cv1"), q + ".cv2"))
        return self.conv(torch.cat((x2, self.conv(x, p + ".cv2")), 1), p + ".cv3")

    def adown(self, x, p):  # :40-52
        x = self.q(F.avg_pool2d(x, 2, 1, 0, False, True))
        x1, x2 = x.chunk(2, 1)
        x1 = self.conv(x1, p + ".cv1", stride=2)
        x2 = F.max_pool2d(x2, 3, 2, 1)
        x2 = self.conv(x2, p + ".cv2")
        return torch.cat((x1, x2), 1)

    def aconv(self, x, p):  # :54-63
        return self.conv(self.q(F.avg_pool2d(x, 2, 1, 0, False, True)), p + ".cv1", stride=2)

    def cblinear(self, x, p, splits):  # bare conv written in the storage type
        return self.q(self._conv2d(x, p + ".conv")).split(splits, 1)

    def cbfuse(self, parts, last):  # f32 sum, one rounding (conv_direct.hip fuse_kernel)
        return self.q(YOLOv9Oracle.cbfuse(parts, last))

    def network_input(self, frames: np.ndarray) -> torch.Tensor:
        return self.q(super().network_input(frames))


def gptq_f16(w: torch.Tensor, H: torch.Tensor, damp: float = 0.01) -> torch.Tensor:
    """Checker for clearcam_amd/csrc/calibrate.hip ``gptq_round_f16`` (dtype "f16c"): w (co, ci) f32, H (ci, ci) f64 second moments of the
    conv's input -> f16-representable f32 weights.  The GPTQ column walk (Frantar et al. 2022, restated from the paper): H + damp mean(diag) I
    is inverted, U = chol(H^-1)^T upper triangular; column i is rounded to nearest f16 and its error, divided by U[i, i], is fed forward into
    the columns behind it along U[i, i+1:].  torch.linalg factorisations in float64 (the library uses its own plain loops)."""
    Wm = w.double().clone()
    ci = Wm.shape[1]
    Hd = H.clone().double()
    Hd += torch.eye(ci, dtype=torch.float64) * damp * Hd.diag().mean()
    U = torch.linalg.cholesky(torch.cholesky_inverse(torch.linalg.cholesky(Hd)), upper=True)
    Q = torch.empty_like(Wm)
    for i in range(ci):
        q = Wm[:, i].float().to(torch.float16).double()
        Q[:, i] = q
        if i + 1 < ci:
            Wm[:, i + 1:] -= ((Wm[:, i] - q) / U[i, i])[:, None] * U[i, i + 1:][None, :]
    return Q.float()


def rel_rms(a: torch.Tensor, b: torch.Tensor) -> float:
    return float(torch.sqrt(((a - b) ** 2).mean() / (b ** 2).mean()))

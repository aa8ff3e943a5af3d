"""Architecture tables for the detector and the CLIP towers.

Host-side description of the graphs the hot path runs.  The numbers restate the
reference's hyper-parameter tables:

* YOLOv9 t/s/m/c: ``detection/yolov9.py:298-326`` (graph) and ``:461-464`` (SIZES).
* OpenCLIP ViT-L/14: ``models/objects.py:29-89``.

The reference keeps a 22-letter positional list per size; here every entry has a
role name and every value that is a sum of two others is *derived* (and
asserted) instead of tabulated, so a typo cannot silently change the graph.
"""
from __future__ import annotations

from dataclasses import dataclass


@dataclass(frozen=True)
class YoloArch:
    size: str
    stem: int        # block 0 out; block 1 out = 2*stem
    b2_kind: str     # "elan1" (t, s) or "elan4" (m, c)
    b2_hidden: int   # elan1: cv1 out (== block-2 out); elan4: RepNCSP width
    b2_out: int
    down_kind: str   # "adown" (c) or "aconv" (t, s, m)
    d3_out: int      # block 3 out
    e4_hidden: int   # block 4 / 15 RepNCSP width
    p3: int          # block 4 out is b4_out; block 15 out (P3 to the head)
    b4_out: int
    d5_out: int      # block 5 out
    e6_hidden: int   # blocks 6, 12, 18 RepNCSP width
    p4: int          # blocks 6, 12, 18 out
    d7_out: int      # block 7 out
    e8_hidden: int   # blocks 8, 21 RepNCSP width
    p5: int          # blocks 8, 9, 21 out
    spp_hidden: int  # block 9 cv1 out
    d16_out: int     # block 16 out
    d19_out: int     # block 19 out
    cls_hidden: int  # head cv3 hidden width
    rep_n: int       # RepNBottleneck count per RepNCSP

    @property
    def nc(self) -> int:
        return 80

    @property
    def reg_max(self) -> int:
        return 16


def _mk(size, stem, b2_kind, b2_hidden, b2_out, down_kind, d3_out, e4_hidden, p3, b4_out, d5_out,
        e6_hidden, p4, d7_out, e8_hidden, p5, spp_hidden, d16_out, d19_out, cls_hidden, rep_n):
    return YoloArch(size, stem, b2_kind, b2_hidden, b2_out, down_kind, d3_out, e4_hidden, p3, b4_out,
                    d5_out, e6_hidden, p4, d7_out, e8_hidden, p5, spp_hidden, d16_out, d19_out,
                    cls_hidden, rep_n)


YOLO_ARCH = {
    #          stem  b2      hid out  down     d3   e4h p3   b4o  d5   e6h  p4   d7   e8h  p5   spp  d16 d19  cls n
    "t": _mk("t", 16, "elan1", 32, 32, "aconv", 64, 16, 64, 64, 96, 24, 96, 128, 32, 128, 64, 48, 64, 80, 3),
    "s": _mk("s", 32, "elan1", 64, 64, "aconv", 128, 32, 128, 128, 192, 48, 192, 256, 64, 256, 128, 96, 128, 128, 3),
    "m": _mk("m", 32, "elan4", 32, 128, "aconv", 240, 60, 240, 240, 360, 90, 360, 480, 120, 480, 240, 184, 240, 240, 1),
    "c": _mk("c", 64, "elan4", 32, 256, "adown", 256, 64, 256, 512, 512, 128, 512, 512, 128, 512, 256, 256, 512, 256, 1),
}


@dataclass(frozen=True)
class ClipArch:
    """OpenCLIP ViT-L/14 (laion2B-s32B-b82K) — ``models/objects.py:29-89``."""
    image_size: int = 224
    patch: int = 14
    v_width: int = 1024
    v_layers: int = 24
    v_heads: int = 16
    v_mlp: int = 4096
    t_ctx: int = 77
    t_vocab: int = 49408
    t_width: int = 768
    t_layers: int = 12
    t_heads: int = 12
    t_mlp: int = 3072
    embed: int = 768

    @property
    def v_tokens(self) -> int:
        return (self.image_size // self.patch) ** 2 + 1


CLIP_L14 = ClipArch()
# OpenAI / OpenCLIP ViT-B/32 (BASELINE.json's north star names it; the reference code itself instantiates ViT-L/14).
# Same graph, different sizes: 7x7 patches + class token = 50 tokens, 12 heads of 64, 512-d joint space.
CLIP_B32 = ClipArch(image_size=224, patch=32, v_width=768, v_layers=12, v_heads=12, v_mlp=3072,
                    t_ctx=77, t_vocab=49408, t_width=512, t_layers=12, t_heads=8, t_mlp=2048, embed=512)
# A shrunken tower with the same structure, for fast CPU/GPU unit tests.
CLIP_TINY = ClipArch(image_size=56, patch=14, v_width=128, v_layers=2, v_heads=2, v_mlp=256,
                     t_ctx=77, t_vocab=512, t_width=64, t_layers=2, t_heads=1, t_mlp=128, embed=64)

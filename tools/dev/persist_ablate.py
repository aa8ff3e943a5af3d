"""dev tool: timing ablations of conv_persist_kernel (bf16, 16x16x32) through cc_conv_bench: where does a tile's time go?
   python tools/dev/persist_ablate.py
Needs a development build of the library: the ablation instantiations are compiled only with -DCC_PERSIST_ABLATIONS
(HIPCC_EXTRA="-DCC_PERSIST_ABLATIONS" python -m clearcam_amd.build --force); without it every row times the full kernel."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from clearcam_amd import _lib  # noqa: E402

L = _lib.lib()
ABL = [("full", 0), ("no DMA in loop", 1), ("no DMA, no frag reads", 2), ("no MFMA", 3), ("no epilogue", 4), ("MFMA only (no DMA/reads/barriers)", 5), ("MFMA only, no epilogue", 6), ("no global stores", 7), ("no first pieces", 8), ("no first pieces, no stores", 9), ("MFMA only + epilogue w/o first pieces", 10)]
SHAPES = [
    ("gemm 65535x4096x1024 (CLIP fc) act none", 255, 257, 1, 1024, 4096, 1, 1, 0),
    ("gemm 65535x4096x1024 (CLIP fc, gelu)", 255, 257, 1, 1024, 4096, 1, 1, 2),
    ("3x3 256->256 @80x80 B64 (head)", 64, 80, 80, 256, 256, 3, 1, 1),
    ("1x1 256->256 @160x160 B64", 64, 160, 160, 256, 256, 1, 1, 1),
]
for name, B, H, W, Cin, Cout, k, stride, act in SHAPES:
    Ho, Wo = (H + 2 * (k // 2) - k) // stride + 1, (W + 2 * (k // 2) - k) // stride + 1
    gf = 2.0 * B * Ho * Wo * Cout * Cin * k * k / 1e9
    tiles = ((B * Ho * Wo + 255) // 256) * (Cout // 256)
    print(f"{name}: {gf:.1f} GF, {tiles} tiles = {tiles / 256:.2f} per CU, {Cin * k * k // 64} K tiles each", flush=True)
    for rnd in range(2):
        for label, abl in ABL:
            _lib.check(L.cc_dev_set(b"phase_flags", 512 + (abl << 12)))
            ms = C.c_float()
            rc = L.cc_conv_bench(2, B, H, W, Cin, Cout, k, stride, act, 7, 20, C.byref(ms))
            per_tile = ms.value * 1e3 / max(1.0, tiles / 256)
            print(f"   [{rnd}] {label:36} {ms.value * 1e3:8.1f} us  {gf / ms.value:6.0f} TF-equivalent   {per_tile:6.2f} us per tile-round", flush=True)
_lib.check(L.cc_dev_set(b"phase_flags", -1))

"""dev (build with HIPCC_EXTRA=-DCC_PERSIST_ABLATIONS, run with CLEARCAM_BENCH_DUMP=1): s_memtime ticks block 0's waves spend in the eight-wave kernel's K loops,
of which at barriers and at the counted DMA wait (phase_flags 512 + 11 << 12; bf16).   python tools/dev/persist_stamps.py"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from clearcam_amd import _lib
L = _lib.lib()
for (name, B, H, W, Cin, Cout, k) in [("3x3 256->256 @80x80 B64", 64, 80, 80, 256, 256, 3), ("gemm 65535x4096x1024 (CLIP fc)", 255, 257, 1, 1024, 4096, 1), ("1x1 1024->512 @40x40 B64", 64, 40, 40, 1024, 512, 1)]:
    for flags in (512, 512 + (11 << 12)):
        L.cc_dev_set(b"phase_flags", flags)
        t = C.c_float()
        rc = L.cc_conv_bench(2, B, H, W, Cin, Cout, k, 1, 1, 7, 5, C.byref(t))
        print(f"{name}: flags {flags} rc {rc} {t.value * 1e3:.1f} us per launch (rows above with stamps: per wave of block 0 - ticks in K loops, at barriers, at the DMA wait, K tiles per tile)", flush=True)
L.cc_dev_set(b"phase_flags", -1)

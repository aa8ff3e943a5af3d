# dev tool (GPU): timing ablations of conv_stream_kernel on 256 -> 256 at 160x160, B = 64 (cc_dev_set("stream_abl", bits); results are wrong
# with any bit set): 1 no MFMA, 2 no fragment reads, 4 no activation arithmetic, 8 no DMA inside the loop, 16 no stores.
# Needs a development build of the library: HIPCC_EXTRA=-DCC_STREAM_ABLATIONS python clearcam_amd/build.py --force
import ctypes as C, sys
from clearcam_amd import _lib
L = _lib.lib()
H = int(sys.argv[1]) if len(sys.argv) > 1 else 160
only = int(sys.argv[2]) if len(sys.argv) > 2 else None
if len(sys.argv) > 3: _lib.check(L.cc_dev_set(b"stream_flags", int(sys.argv[3])))
for dt, planes in ((1, 1), (3, 2)):
    for abl, what in ((0, "full"), (1, "no MFMA"), (2, "no fragment reads"), (3, "no MFMA, no reads"), (4, "no activation"), (7, "no MFMA / reads / activation"), (8, "no DMA in loop"),
                      (16, "no stores"), (20, "no activation, no stores"), (24, "no DMA, no stores"), (23, "memory only: DMA + barriers")):
        if only is not None and abl != only: continue
        ms = C.c_float()
        _lib.check(L.cc_dev_set(b"stream_abl", abl))
        _lib.check(L.cc_conv_bench(dt, 64, H, H, 256, 256, 1, 1, 1, 10, 20, C.byref(ms)))
        print(f"planes {planes} abl {abl:2d} {what:32s} {ms.value:.4f} ms", flush=True)
_lib.check(L.cc_dev_set(b"stream_abl", 0))

"""dev: every distinct conv shape of a per-launch table (gpurun_out/ab_*.csv) under every forced kernel variant (cc_conv_bench):
which kernel the selection rules SHOULD pick.   python tools/dev/variant_sweep.py gpurun_out/ab_fused2_b64.csv 64 [f16|bf16]
Columns "phase" / "persist": variant 7 with the one-tile-per-block kernel / the persistent loop forced (cc_dev_set phase_flags)."""
import csv, ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from clearcam_amd import _lib
L = _lib.lib()
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["kind"] == "conv"]
B = int(sys.argv[2])
DT = {"f16": 1, "bf16": 2, "f16s": 3}[sys.argv[3] if len(sys.argv) > 3 else "f16"]
shapes = {}
for r in rows:
    M, Cin, Cout, ks, st = int(float(r["M"])), int(r["Cin"]), int(r["Cout"]), int(r["ks"]), int(r["stride"])
    key = (M, Cin, Cout, ks, st)
    shapes.setdefault(key, [0, 0.0]); shapes[key][0] += 1; shapes[key][1] += float(r["ms"])
names = {0: "auto", 2: "generic", 3: "halo", 4: "ws", 5: "big256", 6: "big128", 7: "phase", 77: "persist", 8: "wave", 10: "stream", 12: "tile64", 93: "s64/3"}
print("shape (M Cin Cout k s) x n, in-plan us | " + " ".join(f"{n:>7}" for n in names.values()))
tot_auto = tot_best = 0.0
for (M, Cin, Cout, ks, st), (n, ms) in sorted(shapes.items(), key=lambda kv: -kv[1][1]):
    Ho = int(round((M / B) ** 0.5)); H = Ho * st
    if Ho * Ho * B != M or Cin % 8 or Cout % 8:
        continue
    res = {}
    for v in names:
        t = C.c_float()
        L.cc_dev_set(b"phase_flags", 32 if v == 7 else (512 if v == 77 else -1))
        rc = L.cc_conv_bench(DT, B, H, H, Cin, Cout, ks, st, 1, 7 if v == 77 else v, 20, C.byref(t))
        L.cc_dev_set(b"phase_flags", -1)
        res[v] = t.value * 1e3 if rc == 0 else float("nan")
    best = min((x, v) for v, x in res.items() if x == x)
    tot_auto += n * res[0]; tot_best += n * best[0]
    print(f"{M:8d} {Cin:5d} {Cout:4d} {ks} {st} x{n:2d} {ms / n * 1e3:7.1f} | " + " ".join(f"{res[v]:7.1f}" for v in names) + f"  best {names[best[1]]}", flush=True)
print(f"sum over launches: auto {tot_auto / 1e3:.3f} ms, best-per-shape {tot_best / 1e3:.3f} ms")

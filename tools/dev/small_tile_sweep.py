"""dev: the few-tile conv configurations on the shapes of a per-launch table (cc_conv_bench): auto, 128x32 tiles with 4 / 6 stages (91 / 92), 128x64 / 3 (93),
64x32 tiles on 256 threads with 6 / 4 stages (94 / 95).   python tools/dev/small_tile_sweep.py gpurun_out/r06z_yolo_per_launch_b1.csv 1 [f16|f16s]"""
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
    key = (int(float(r["M"])), int(r["Cin"]), int(r["Cout"]), int(r["ks"]), int(r["stride"]))
    shapes.setdefault(key, [0, 0.0]); shapes[key][0] += 1; shapes[key][1] += float(r["ms"])
names = {0: "auto", 91: "128x32/4", 93: "128x64/3", 95: "64x32/4", 97: "32x32/4", 6: "big128", 7: "8-wave", 2: "generic"}
print("shape (M Cin Cout k s) x n, in-plan us | " + " ".join(f"{n:>9}" for n in names.values()))
tot = {v: 0.0 for v in names}; tot_best = 0.0
for (M, Cin, Cout, ks, st), (n, ms) in sorted(shapes.items(), key=lambda kv: -kv[1][1]):
    Ho = int(round((M / B) ** 0.5)); H = Ho * st
    if Ho * Ho * B != M or Cin % 8 or Cout % 8:
        continue
    res = {v: float("inf") for v in names}
    for _ in range(2):
        for v in names:
            t = C.c_float()
            if L.cc_conv_bench(DT, B, H, H, Cin, Cout, ks, st, 1, v, 30, C.byref(t)) == 0:
                res[v] = min(res[v], t.value * 1e3)
    best = min((x, v) for v, x in res.items())
    for v in names: tot[v] += n * res[v]
    tot_best += n * best[0]
    print(f"{M:8d} {Cin:5d} {Cout:4d} {ks} {st} x{n:2d} {ms / n * 1e3:7.1f} | " + " ".join(f"{res[v]:9.1f}" for v in names) + f"  best {names[best[1]]}", flush=True)
print("sum over launches (us): " + "  ".join(f"{names[v]} {tot[v]:.0f}" for v in names) + f"  best-per-shape {tot_best:.0f}")

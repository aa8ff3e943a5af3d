# dev tool: per-layer A/B of two per-launch profile CSVs (CLEARCAM_PROFILE_CSV dumps)
import csv, sys, collections
def load(p):
    g = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(p)):
        if r["kind"] != "conv": continue
        k = (int(float(r["M"])), int(r["Cin"]), int(r["Cout"]), int(r["ks"]), int(r["stride"]))
        g[k][0] += float(r["ms"]); g[k][1] += 1
    return g
a, b = load(sys.argv[1]), load(sys.argv[2])
rows = sorted(a, key=lambda k: -(abs(a[k][0] - b[k][0])))
print("shape (M,Cin,Cout,k,s)            n    A ms    B ms   B/A")
for k in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 18]:
    print(f"{str(k):34s} {a[k][1]:3d} {a[k][0]:7.3f} {b[k][0]:7.3f} {b[k][0]/a[k][0]:6.2f}")
print("total", round(sum(v[0] for v in a.values()), 3), round(sum(v[0] for v in b.values()), 3))

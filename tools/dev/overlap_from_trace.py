"""dev tool: how much of the time do kernels of different batches really run side by side?  Reads a rocprofv3 --kernel-trace CSV
(Start_Timestamp / End_Timestamp / Queue_Id per dispatch) and prints, for the steady-state window, the share of wall time with
0, 1, 2, 3+ kernels in flight, the sum of kernel durations against the wall time, and the dispatches per hardware queue."""
import csv
import glob
import sys
from collections import Counter

files = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
rows = []
for fn in files:
    for r in csv.DictReader(open(fn)):
        name = r["Kernel_Name"]
        if "spin_kernel" in name:
            continue
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), name))
rows.sort()
lo, hi = rows[int(len(rows) * 0.4)][0], rows[int(len(rows) * 0.9)][0]      # steady state by dispatch count: skip plan building / ramp-up and the drain
ev = []
for s, e, q, n in rows:
    s2, e2 = max(s, lo), min(e, hi)
    if e2 > s2:
        ev.append((s2, 1)); ev.append((e2, -1))
ev.sort()
hist, cur, last = Counter(), 0, lo
for t, d in ev:
    hist[min(cur, 3)] += t - last
    cur += d; last = t
hist[min(cur, 3)] += hi - last
wall = hi - lo
busy = sum(min(e, hi) - max(s, lo) for s, e, q, n in rows if min(e, hi) > max(s, lo))
print(f"window {wall / 1e6:.2f} ms; sum of kernel durations {busy / 1e6:.2f} ms = {busy / wall:.2f} x the wall time")
for k in range(4):
    print(f"  {k}{'+' if k == 3 else ' '} kernels in flight: {100.0 * hist[k] / wall:5.1f} % of the time")
qs = Counter(q for s, e, q, n in rows if s >= lo and e <= hi)
print("dispatches per hardware queue in the window:", dict(qs))

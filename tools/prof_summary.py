"""Summarise rocprofv3 output (kernel-trace CSV and/or PMC counter CSV) into a small text table.

usage: python tools/prof_summary.py <dir-with-rocprofv3-csvs> [--match conv_mfma] > profiles/<name>.txt
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name: str) -> str:
    name = name.replace("void ", "").replace("cc::", "")
    return name[:110]


def main():
    d = sys.argv[1]
    traces = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    counters = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if traces:
        agg = defaultdict(lambda: [0, 0.0, 1e30, 0.0])
        total = 0.0
        for t in traces:
            for r in csv.DictReader(open(t)):
                dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
                a = agg[short(r["Kernel_Name"])]
                a[0] += 1; a[1] += dur; a[2] = min(a[2], dur); a[3] = max(a[3], dur)
                total += dur
        print(f"# kernel trace: {sum(a[0] for a in agg.values())} dispatches, {total/1e3:.3f} ms GPU time")
        print(f"{'kernel':112} {'calls':>7} {'total_us':>12} {'avg_us':>10} {'min_us':>9} {'max_us':>9} {'pct':>6}")
        for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            print(f"{k:112} {a[0]:7d} {a[1]:12.1f} {a[1]/a[0]:10.2f} {a[2]:9.2f} {a[3]:9.2f} {100*a[1]/total:6.2f}")
    if counters:
        agg = defaultdict(lambda: defaultdict(float))
        calls = defaultdict(int)
        for t in counters:
            for r in csv.DictReader(open(t)):
                k = short(r["Kernel_Name"])
                agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
                calls[(k, r["Counter_Name"])] += 1
        print("# PMC counters (sum over dispatches; FETCH_SIZE/WRITE_SIZE are in KiB-units as reported by rocprofv3)")
        for k in sorted(agg):
            for c, v in sorted(agg[k].items()):
                print(f"{k:112} {c:16} sum={v:16.1f} dispatches={calls[(k, c)]}")


if __name__ == "__main__":
    main()

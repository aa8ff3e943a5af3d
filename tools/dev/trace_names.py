# dev tool: print per-dispatch durations of GEMM-like kernels from a rocprofv3 kernel trace directory
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
last = None
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "conv_" in n or "Cijk" in n:
        us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        key = (n[:60], r.get("Grid_Size_X", r.get("Grid_Size", "")))
        if key != last: print()
        last = key
        print(f"{key[0]} g{key[1]} {us:.0f}", end=" ")
print()

#!/bin/bash
# dev: any set of PMC counters (one pass) for the kernels of a workload; per-kernel sums to gpurun_out/<tag>_pmc.txt
#   COUNTERS="TCC_HIT_sum TCC_MISS_sum" bash tools/dev/pmc_counters.sh <tag> <python script> [args]
tag=$1; shift
export TMPDIR=/tmp PYTHONPATH=$PWD
root=$PWD
cd /tmp && rm -rf /tmp/pc_$tag
rocprofv3 --kernel-trace --pmc $COUNTERS --output-format csv -d /tmp/pc_$tag -o pc -- python $root/"$@" > /tmp/pc_$tag.log 2>&1
cd $root
tail -3 /tmp/pc_$tag.log | cut -c1-200
python - "$tag" <<'PY'
import csv, glob, sys
from collections import defaultdict
tag = sys.argv[1]
agg = defaultdict(lambda: defaultdict(float)); calls = defaultdict(int); names = []
for f in glob.glob(f"/tmp/pc_{tag}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void ", "").replace("cc::", "")[:70]
        if r["Counter_Name"] not in names: names.append(r["Counter_Name"])
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == names[0]: calls[k] += 1
out = open(f"gpurun_out/{tag}_pmc.txt", "w")
hdr = f"{'kernel':72} {'calls':>6} " + " ".join(f"{n:>18}" for n in names)
print(hdr); out.write(hdr + "\n")
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1].values()))[:16]:
    line = f"{k:72} {calls[k]:6d} " + " ".join(f"{v.get(n, 0):18.4e}" for n in names)
    print(line); out.write(line + "\n")
PY

#!/bin/bash
# dev: one rocprofv3 PMC pass with a caller-chosen counter list; per-kernel sums (raw values and per-call) to gpurun_out/<tag>_pmc.txt
#   bash tools/dev/pmc_any.sh <tag> "SQ_WAVE_CYCLES SQ_INSTS_LDS ..." <python script> [args]
tag=$1; ctrs=$2; shift; shift
export TMPDIR=/tmp PYTHONPATH=$PWD
root=$PWD
cd /tmp && rm -rf /tmp/pa_$tag
rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d /tmp/pa_$tag -o pa -- python $root/"$@" > /tmp/pa_$tag.log 2>&1
cd $root
python - "$tag" "$ctrs" <<'PY'
import csv, glob, sys
from collections import defaultdict
tag, names = sys.argv[1], sys.argv[2].split()
agg = defaultdict(lambda: defaultdict(float)); calls = defaultdict(int)
for f in glob.glob(f"/tmp/pa_{tag}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void ", "").replace("cc::", "")[:60]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == names[0]: calls[k] += 1
out = open(f"gpurun_out/{tag}_pmc.txt", "a")
hdr = f"{'kernel':62} {'calls':>6} " + " ".join(f"{n[3:] if n.startswith('SQ_') else n:>22}" for n in names) + "   (per call)"
print(hdr); out.write(hdr + "\n")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get(names[0], 0))[:10]:
    c = max(calls[k], 1)
    line = f"{k:62} {calls[k]:6d} " + " ".join(f"{v.get(n, 0) / c:22.4e}" for n in names)
    print(line); out.write(line + "\n")
PY

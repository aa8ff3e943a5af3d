#!/bin/bash
# dev: SQ counters (one pass, eight slots) for the kernels of a workload; per-kernel sums to gpurun_out/<tag>_sq.txt
#   bash tools/dev/pmc_kernels.sh <tag> <python script> [args]
tag=$1; shift
export TMPDIR=/tmp PYTHONPATH=$PWD
root=$PWD
cd /tmp && rm -rf /tmp/sq_$tag
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES \
  --output-format csv -d /tmp/sq_$tag -o sq -- python $root/"$@" > /tmp/sq_$tag.log 2>&1
cd $root
python - "$tag" <<'PY'
import csv, glob, sys
from collections import defaultdict
tag = sys.argv[1]
agg = defaultdict(lambda: defaultdict(float)); calls = defaultdict(int)
for f in glob.glob(f"/tmp/sq_{tag}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void ", "").replace("cc::", "")[:70]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVE_CYCLES": calls[k] += 1
names = ["SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_BUSY_CU_CYCLES", "SQ_LDS_BANK_CONFLICT", "SQ_VALU_MFMA_BUSY_CYCLES"]
out = open(f"gpurun_out/{tag}_sq.txt", "w")
hdr = f"{'kernel':72} {'calls':>6} " + " ".join(f"{n[3:]:>20}" for n in names) + f" {'MFMA_pipe_busy':>16}" + "   (fractions of WAVE_CYCLES; MFMA_BUSY in cycles, the others in quad-cycles; last column: matrix-pipe busy share of a SIMD = MFMA_BUSY cycles / (4 cycles x BUSY_CU quad-cycles; BUSY_CU sums the four SIMDs: it reads 0.50 of WAVE_CYCLES for a kernel with two waves per SIMD))"
print(hdr); out.write(hdr + "\n")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:14]:
    w = v.get("SQ_WAVE_CYCLES", 1.0) or 1.0
    line = f"{k:72} {calls[k]:6d} {w:20.3e} " + " ".join(f"{v.get(n, 0) / w:20.3f}" for n in names[1:]) + f" {v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / max(v.get('SQ_BUSY_CU_CYCLES', 0), 1.0) / 4.0:16.3f}"
    print(line); out.write(line + "\n")
PY

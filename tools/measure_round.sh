#!/bin/bash
# GPU-box script: everything profiles/<tag>_* is made from.   bash tools/measure_round.sh r02d
#   1 bench line (default bench.py run: three batches in flight + the back-to-back number) -> gpurun_out/<tag>_bench_line.json
#   2 per-launch tables of the bench plan, B=64/1  -> gpurun_out/<tag>_yolo_per_launch.csv, <tag>_yolo_per_launch_b1.csv
#   3 rocprofv3 --kernel-trace --stats, 10 replays, ONE batch in flight (no launch shares the chip with another) -> gpurun_out/<tag>_yolo_kernel_stats.csv (+ .txt summary)
#   4 rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, kernel trace only) -> gpurun_out/pmc_traffic.json, <tag>_pmc_traffic.txt
#   5 rocprofv3 SQ pass (wave cycles, waits, VALU / LDS activity, bank conflicts, SQ_VALU_MFMA_BUSY_CYCLES) -> gpurun_out/<tag>_plan_sq.txt
# The PMC record carries the digest of clearcam_amd/csrc; bench.py quotes it only while the digest matches, so run this on the final code
# and run the bench LAST (step 1 is executed after step 4 for that reason).
set -u
tag=${1:-r02d}
export CLEARCAM_BENCH_DTYPE=${CLEARCAM_BENCH_DTYPE:-f16h}      # the storage mode every record below is taken in (bench.py's default)
dt=$CLEARCAM_BENCH_DTYPE
export TMPDIR=/tmp PYTHONPATH=$PWD
root=$PWD
mkdir -p gpurun_out
CLEARCAM_PROFILE_CSV=$root/gpurun_out/${tag}_yolo_per_launch.csv python tools/dev/prof_csv.py 64 $dt > gpurun_out/${tag}_prof64.txt 2>&1
CLEARCAM_PROFILE_CSV=$root/gpurun_out/${tag}_yolo_per_launch_b1.csv python tools/dev/prof_csv.py 1 $dt > gpurun_out/${tag}_prof1.txt 2>&1
(cd /tmp && rm -rf /tmp/kt_$tag && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$tag -o kt -- python $root/tools/dev/plan_passes.py 9 > $root/gpurun_out/${tag}_rocprof.log 2>&1)
f=$(find /tmp/kt_$tag -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/${tag}_yolo_kernel_stats.csv
python tools/prof_summary.py /tmp/kt_$tag > gpurun_out/${tag}_yolo_${dt}_b64.txt 2>/dev/null
python - "$tag" <<'PY'
import csv, json, os, sys
from bench import kernel_source_digest
tag = sys.argv[1]
rows = list(csv.DictReader(open(f"gpurun_out/{tag}_yolo_kernel_stats.csv")))
from tools.kernel_family import family as fam
tot = {}
for r in rows:
    tot[fam(r["Name"])] = tot.get(fam(r["Name"]), 0.0) + float(r["TotalDurationNs"]) / 1e6
import torch
dt = os.environ.get("CLEARCAM_BENCH_DTYPE", "f16h")
rec = {"kernel_source_digest": kernel_source_digest(), "tag": tag, "replays": 10, "dtype": dt,
       "device_name": torch.cuda.get_device_name(0) if torch.cuda.is_available() else None,
       "command": f"rocprofv3 --kernel-trace --stats -- python tools/dev/plan_passes.py 9   (10 replays of the bench plan, YOLOv9-C {dt} B=64 640x640)",
       "conv_ms_per_step": tot.get("conv", 0.0) / 10, "pool_ms_per_step": tot.get("pool", 0.0) / 10, "stem_ms_per_step": tot.get("stem", 0.0) / 10,
       "all_kernels_ms_per_step": sum(tot.values()) / 10}
json.dump(rec, open("gpurun_out/kernel_trace.json", "w"), indent=1)
json.dump(rec, open("profiles/kernel_trace.json", "w"), indent=1)
print(json.dumps(rec))
PY
(cd /tmp && python $root/tools/pmc_traffic.py $tag > $root/gpurun_out/${tag}_pmc.log 2>&1)
cp profiles/pmc_traffic.json gpurun_out/pmc_traffic.json 2>/dev/null
bash tools/dev/pmc_kernels.sh ${tag}_plan tools/dev/plan_passes.py 3 > gpurun_out/${tag}_sq_stdout.txt 2>&1      # the MFMA-busy table of the plan
python bench.py > gpurun_out/${tag}_bench_line.json 2> gpurun_out/${tag}_bench.err
cp bench_detail.json gpurun_out/${tag}_bench_detail.json 2>/dev/null
tail -c 600 gpurun_out/${tag}_bench_line.json; echo; tail -3 gpurun_out/${tag}_pmc.log | cut -c1-400; head -12 gpurun_out/${tag}_yolo_${dt}_b64.txt

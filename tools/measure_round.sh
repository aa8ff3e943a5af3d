#!/bin/bash
# GPU-box script: everything profiles/<tag>_* is made from.   bash tools/measure_round.sh r02d
#   1 bench line (default bench.py run)            -> gpurun_out/<tag>_bench_line.json
#   2 per-launch tables of the bench plan, B=64/1  -> gpurun_out/<tag>_yolo_per_launch.csv, <tag>_yolo_per_launch_b1.csv
#   3 rocprofv3 --kernel-trace --stats, 10 replays -> gpurun_out/<tag>_yolo_kernel_stats.csv (+ .txt summary)
#   4 rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, kernel trace only) -> gpurun_out/pmc_traffic.json, <tag>_pmc_traffic.txt
# The PMC record carries the digest of clearcam_amd/csrc; bench.py quotes it only while the digest matches, so run this on the final code
# and run the bench LAST (step 1 is executed after step 4 for that reason).
set -u
tag=${1:-r02d}
export TMPDIR=/tmp PYTHONPATH=$PWD
root=$PWD
mkdir -p gpurun_out
CLEARCAM_PROFILE_CSV=$root/gpurun_out/${tag}_yolo_per_launch.csv python tools/dev/prof_csv.py 64 bf16 > gpurun_out/${tag}_prof64.txt 2>&1
CLEARCAM_PROFILE_CSV=$root/gpurun_out/${tag}_yolo_per_launch_b1.csv python tools/dev/prof_csv.py 1 bf16 > gpurun_out/${tag}_prof1.txt 2>&1
(cd /tmp && rm -rf /tmp/kt_$tag && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$tag -o kt -- python $root/tools/dev/plan_passes.py 9 > $root/gpurun_out/${tag}_rocprof.log 2>&1)
f=$(find /tmp/kt_$tag -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/${tag}_yolo_kernel_stats.csv
python tools/prof_summary.py /tmp/kt_$tag > gpurun_out/${tag}_yolo_bf16_b64.txt 2>/dev/null
(cd /tmp && python $root/tools/pmc_traffic.py $tag > $root/gpurun_out/${tag}_pmc.log 2>&1)
cp profiles/pmc_traffic.json gpurun_out/pmc_traffic.json 2>/dev/null
python bench.py > gpurun_out/${tag}_bench_line.json 2> gpurun_out/${tag}_bench.err
tail -c 600 gpurun_out/${tag}_bench_line.json; echo; tail -3 gpurun_out/${tag}_pmc.log | cut -c1-400; head -12 gpurun_out/${tag}_yolo_bf16_b64.txt

export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python tools/dev/tail_run.py 128 f16,f16h,f16s,f16c,bf16 g3,g10 2>&1 | grep -E "^#|^checkpoint" | tee gpurun_out/r05p_stress_128.txt

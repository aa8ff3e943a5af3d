export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_yolo.py -q -k "calibrated_mode" 2>&1 | tail -3
timeout 1500 python tools/dev/tail_run.py 256 f16,f16h,f16s,f16c,f16c:smooth,f16c:blocks 2>&1 | grep -E "^#|^checkpoint" | tee gpurun_out/r05o_tail_256.txt

export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 | tee gpurun_out/r05q_pytest_gpu.txt

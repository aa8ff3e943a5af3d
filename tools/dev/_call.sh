export PYTHONPATH=$PWD TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_yolo.py -x -q -k "pool_rows" 2>&1 | tail -3

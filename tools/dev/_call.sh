export PYTHONPATH=$PWD TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_yolo.py -x -q -s -k "split_weight_mode_all_sizes" 2>&1 | grep -v "amdgpu.ids\|stream probe" | tail -8 | cut -c1-400

export PYTHONPATH=$PWD TMPDIR=/tmp
timeout 600 python tools/dev/soak.py flight 2>&1 | grep -v amdgpu.ids | tail -3

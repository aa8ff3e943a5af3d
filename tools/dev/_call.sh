export PYTHONPATH=$PWD TMPDIR=/tmp
for fl in 2 10 2 10; do CLEARCAM_STREAM_FLAGS=$fl timeout 300 python tools/dev/step_time.py f16h 2>&1 | grep "^f16h" | cut -c1-120 | sed "s/^/flags=$fl /"; done
CLEARCAM_STREAM_FLAGS=10 timeout 300 python -m pytest tests/test_gpu_yolo.py -x -q -k "stream" 2>&1 | tail -1

export PYTHONPATH=$PWD TMPDIR=/tmp
root=$PWD
timeout 600 python -m pytest tests/test_gpu_yolo.py -x -q -k "pool or adown or batch_invariance" 2>&1 | tail -2
python - <<'PY'
# the 2x2 average with four rows per thread against one row per thread: same bits on the detector's ADown shapes (subprocess per mode: the switch is read once)
import subprocess, sys, os, numpy as np
code = """
import numpy as np, sys
from clearcam_amd.weights import conditioned_yolov9_state_dict
from clearcam_amd.yolov9 import YOLOv9
fr = np.random.default_rng(3).integers(0, 256, (5, 480, 640, 3), dtype=np.uint8)
m = YOLOv9('c', 640, state_dict=conditioned_yolov9_state_dict('c', 1234), dtype='f16h')
np.save(sys.argv[1], m.detect_batch(fr))
"""
outs = []
for rows in ("1", "4", "8"):
    path = f"/tmp/pool_rows_{rows}.npy"
    subprocess.run([sys.executable, "-c", code, path], check=True, env=dict(os.environ, CLEARCAM_POOL_ROWS=rows))
    outs.append(np.load(path))
print("pool rows 1 vs 4 vs 8: identical detections:", np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2]), "dets", int((outs[0][..., 4] > 0).sum()))
PY
for r in 1 4 8 1 4 8; do CLEARCAM_POOL_ROWS=$r timeout 300 python tools/dev/step_time.py f16h 2>&1 | grep "^f16h" | cut -c1-150 | sed "s/^/pool_rows=$r /"; done

export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out
timeout 2400 bash tools/measure_round.sh r05z > gpurun_out/r05z_measure.log 2>&1; echo "measure rc=$?"
tail -c 1500 gpurun_out/r05z_measure.log
python - <<'PY'
import time, numpy as np
from clearcam_amd.weights import synthetic_yolov9_state_dict
from clearcam_amd.yolov9 import YOLOv9
sd = synthetic_yolov9_state_dict("c", 1234)
f = np.random.default_rng(1).integers(0, 256, (1, 640, 640, 3), dtype=np.uint8)
for dt in ("f16h", "f16c", "f16"):
    m = YOLOv9("c", 640, state_dict=sd, dtype=dt)
    for _ in range(20): m.detect_batch(f)
    t = time.perf_counter()
    for _ in range(200): m.detect_batch(f)
    print(f"single frame {dt}: {(time.perf_counter() - t) / 200 * 1e3:.3f} ms per call", flush=True); m.close()
PY

# dev tool: dump the detector output of a few synthetic cameras over time (tracker workload capture)
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from clearcam_amd.weights import shift_class_bias, synthetic_yolov9_state_dict
from clearcam_amd.yolov9 import YOLOv9
from clearcam_amd.streams import StreamPipeline, make_cameras
shift = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
model = YOLOv9("c", 640, state_dict=shift_class_bias(synthetic_yolov9_state_dict("c", 1234), shift), dtype="bf16")
cams = make_cameras(4)
pipe = StreamPipeline(model, 4, track=False)
out = []
for t in range(40):
    pipe.submit([c.read() for c in cams]); p, _ = pipe.collect(); out.append(p.copy())
out = np.stack(out)
print("dets/frame", (out[..., 4] > 0.25).sum() / (40 * 4), "score quantiles", np.quantile(out[..., 4][out[..., 4] > 0], [0.1, 0.5, 0.9]))
os.makedirs("gpurun_out", exist_ok=True)
np.savez_compressed(f"gpurun_out/stream_dets_{int(shift)}.npz", dets=out.astype(np.float32))

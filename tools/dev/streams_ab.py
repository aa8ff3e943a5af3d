"""dev tool: the 64-camera 1080p pipeline (PCIe upload included) by pipeline depth, detector slots on / off and number of copy streams."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from clearcam_amd.streams import StreamPipeline, make_cameras  # noqa: E402
from clearcam_amd.weights import shift_class_bias, synthetic_yolov9_state_dict  # noqa: E402
from clearcam_amd.yolov9 import YOLOv9  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
sd = shift_class_bias(synthetic_yolov9_state_dict("c", 1234), -20.0)
cams = make_cameras(n, seed=100)
for depth, infl, ncopy in ((2, False, 1), (3, False, 1), (2, True, 1), (3, True, 1), (2, False, 1), (3, False, 1)):
    m = YOLOv9("c", 640, state_dict=sd, dtype="f16")
    pipe = StreamPipeline(m, n, depth=depth, in_flight=infl, copy_streams=ncopy)
    st = pipe.run(cams, 24 if n >= 32 else 60)
    st2 = pipe.run(None, 24 if n >= 32 else 60)
    print(f"cams {n} depth {depth} detector slots {'on ' if infl else 'off'} copy streams {ncopy}: {st['frames_per_sec']:7.0f} frames/s  {st['ms_per_batch']:.2f} ms/batch  "
          f"H2D {st['h2d_GBps']:.1f} GB/s  latency p50 {st['latency_ms_p50']:.1f} ms   frames resident: {st2['frames_per_sec']:7.0f} frames/s", flush=True)
    pipe.close(); m.close()

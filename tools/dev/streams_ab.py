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
for depth, infl, ncopy in ((None, None, 1), (2, False, 1), (None, None, 1), (None, None, 1)):
    m = YOLOv9("c", 640, state_dict=sd, dtype="f16")
    pipe = StreamPipeline(m, n, depth=depth, in_flight=infl, copy_streams=ncopy)
    mode = os.environ.get("AB_MODE", "")
    if infl and mode == "resident_only":
        st = {"frames_per_sec": 0.0, "ms_per_batch": 0.0, "h2d_GBps": 0.0, "latency_ms_p50": 0.0}
    else:
        st = pipe.run(cams, 24 if n >= 32 else 60)
    st2 = pipe.run(None, 24 if n >= 32 else 60) if not (infl and mode == "upload_only") else {"frames_per_sec": 0.0}
    gpu_ms = m.last_gpu_ms() if not pipe.in_flight else float("nan")
    if not pipe.in_flight and os.environ.get("AB_REBUILD"):
        m.set_in_flight(1)                                    # drops the cached plans: the next call captures and instantiates again
        st3 = pipe.run(None, 60)
        print(f"      after rebuilding the plan: frames resident {st3['frames_per_sec']:.0f} frames/s, detect on the GPU {m.last_gpu_ms():.2f} ms", flush=True)
    prof = m.profile(iters=2)
    g_all = m.profile_graph(2, 5)
    extra = f"eager per-launch events: conv {prof['conv_ms']:.2f} + pool {prof['pool_ms']:.2f} + stem {prof['stem_ms']:.2f} ms; whole step as a fresh graph {g_all:.2f} ms"
    print(f"cams {n} depth {pipe.depth} detector slots {'on ' if pipe.in_flight else 'off'} copy streams {ncopy}: {st['frames_per_sec']:7.0f} frames/s  {st['ms_per_batch']:.2f} ms/batch  "
          f"H2D {st['h2d_GBps']:.1f} GB/s  latency p50 {st['latency_ms_p50']:.1f} ms   frames resident: {st2['frames_per_sec']:7.0f} frames/s   last detect call on the GPU {gpu_ms:.2f} ms   {extra}", flush=True)
    pipe.close(); m.close()

"""dev tool: does a plain pipeline created after slotted pipelines were torn down replay its graph slowly?  Run once with the stream
pool (default) and once with CLEARCAM_STREAM_POOL=0 (streams destroyed with their handle, the round-3 behaviour)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from clearcam_amd.streams import StreamPipeline, make_cameras  # noqa: E402
from clearcam_amd.weights import synthetic_yolov9_state_dict  # noqa: E402
from clearcam_amd.yolov9 import YOLOv9  # noqa: E402

sd = synthetic_yolov9_state_dict("c", 1234)
cams = make_cameras(8, 1080, 1920, ring=5)


def plain(tag):
    m = YOLOv9("c", 640, state_dict=sd, dtype="f16")
    p = StreamPipeline(m, 8, depth=2, in_flight=False, track=False)
    st = p.run(cams, 30)
    ms = []
    for _ in range(5):
        p.submit(cams.read_all()); p.collect(); ms.append(m.last_gpu_ms())
    print(f"{tag}: plain pipeline {st['frames_per_sec']:.0f} frames/s, detect on the GPU {sorted(ms)[2]:.2f} ms", flush=True)
    p.close(); m.close()


print("stream pool:", os.environ.get("CLEARCAM_STREAM_POOL", "1"))
plain("first")
for depth in (4, 3, 4):
    m = YOLOv9("c", 640, state_dict=sd, dtype="f16")
    p = StreamPipeline(m, 8, depth=depth, in_flight=True, track=False)
    st = p.run(cams, 30)
    print(f"  slotted pipeline depth {depth}: {st['frames_per_sec']:.0f} frames/s", flush=True)
    p.close(); m.close()
    plain("right after it")
    plain("the one after")

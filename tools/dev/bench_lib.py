# dev tool: YOLOv9-C B=64 step time with an alternative library (CC_LIB) - for timing-only experiments
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from clearcam_amd import _lib
if os.environ.get("CC_LIB"): _lib.LIB_PATH = os.environ["CC_LIB"]
from clearcam_amd.weights import synthetic_yolov9_state_dict
from clearcam_amd.yolov9 import YOLOv9
m = YOLOv9("c", 640, state_dict=synthetic_yolov9_state_dict("c", 1234), dtype="bf16")
f = torch.from_numpy(np.random.default_rng(1).integers(0, 256, (64, 640, 640, 3), dtype=np.uint8)).cuda()
o = torch.empty(64, 300, 6, device="cuda")
for _ in range(5): m.detect_batch_device(f, o)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(20): m.detect_batch_device(f, o)
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 20
print("ms/step %.3f fps %.0f" % (dt * 1e3, 64 / dt), m.profile(iters=3)["conv_ms"])

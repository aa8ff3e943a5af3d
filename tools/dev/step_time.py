# dev tool: YOLOv9-C step time per dtype, back-to-back cc_yolo_detect calls and n batches in flight; argv: dtypes (comma list) [batch] [in_flight]
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from clearcam_amd.weights import synthetic_yolov9_state_dict
from clearcam_amd.yolov9 import YOLOv9
dtypes = (sys.argv[1] if len(sys.argv) > 1 else "f16,f16s").split(",")
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
depth = int(sys.argv[3]) if len(sys.argv) > 3 else 3
sd = synthetic_yolov9_state_dict("c", 1234)
f = torch.from_numpy(np.random.default_rng(1).integers(0, 256, (B, 640, 640, 3), dtype=np.uint8)).cuda()
for dt in dtypes:
    m = YOLOv9("c", 640, state_dict=sd, dtype=dt)
    o = torch.empty(B, 300, 6, device="cuda")
    for _ in range(5): m.detect_batch_device(f, o)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): m.detect_batch_device(f, o)
    torch.cuda.synchronize(); one = (time.perf_counter() - t) / 20
    prof = m.profile(iters=3)
    m.set_in_flight(depth)
    outs = [torch.empty(B, 300, 6, device="cuda") for _ in range(depth)]
    for i in range(2 * depth): m.wait(m.submit(f, outs[i % depth]))
    torch.cuda.synchronize(); t = time.perf_counter()
    tk = [m.submit(f, outs[i % depth]) for i in range(30)]
    for k in tk[-depth:]: m.wait(k)
    torch.cuda.synchronize(); many = (time.perf_counter() - t) / 30
    print(f"{dt}: B={B} back-to-back {one*1e3:.3f} ms ({B/one:.0f} frames/s), {depth} in flight {many*1e3:.3f} ms ({B/many:.0f} frames/s)  "
          f"conv {prof['conv_ms']:.3f} pool {prof['pool_ms']:.3f} decode {prof['decode_ms']:.3f} stem {prof['stem_ms']:.3f} ms, {prof['conv_launches']} conv launches", flush=True)
    m.close()

# dev tool: soak test - many iterations of every entry point, device memory must not drift
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from clearcam_amd.arch import CLIP_TINY
from clearcam_amd.weights import shift_class_bias, synthetic_clip_state_dict, synthetic_yolov9_state_dict, synthetic_adaface_state_dict, synthetic_blazeface_state_dict
from clearcam_amd.yolov9 import YOLOv9
from clearcam_amd.objects import OpenCLIP, EmbeddingIndex, preprocess_crops
from clearcam_amd.streams import StreamPipeline, make_cameras
from clearcam_amd.adaface import ADAFACE
from clearcam_amd.blazeface import BlazeFace
def free(): torch.cuda.synchronize(); return torch.cuda.mem_get_info()[0] / 1e6
rng = np.random.default_rng(0)
m = YOLOv9("t", 640, state_dict=shift_class_bias(synthetic_yolov9_state_dict("t", 1234), -20.0), dtype="bf16")
pipe = StreamPipeline(m, 16, (540, 960)); cams = make_cameras(16, 540, 960)
clip = OpenCLIP(state_dict=synthetic_clip_state_dict(CLIP_TINY, 4321), arch=CLIP_TINY, dtype="bf16")
ix = EmbeddingIndex(64, 200000)
face = ADAFACE(state_dict=synthetic_adaface_state_dict(), dtype="bf16"); blaze = BlazeFace(state_dict=synthetic_blazeface_state_dict(), dtype="bf16")
crops = [rng.integers(0, 256, (int(rng.integers(40, 200)), int(rng.integers(40, 200)), 3), dtype=np.uint8) for _ in range(64)]
frame = rng.integers(0, 256, (540, 960, 3), dtype=np.uint8); f112 = rng.integers(0, 256, (112, 112, 3), dtype=np.uint8)
def one_round(n):
    pipe.run(cams, n, warmup=1)
    for _ in range(n):
        x = preprocess_crops(crops, CLIP_TINY.image_size); e = clip.precompute_embedding(x.cpu().numpy()).numpy()
        if len(ix) + len(e) <= 200000: ix.add(e)
        ix.search(e[:8], 10); m(frame); face(f112); blaze(frame)
# batches in flight: depth changes (stream + plan churn), device and pinned-host submissions, CLIP slots
m2 = YOLOv9("t", 320, state_dict=synthetic_yolov9_state_dict("t", 1234), dtype="f16")
fd = torch.from_numpy(rng.integers(0, 256, (4, 320, 320, 3), dtype=np.uint8)).cuda(); fh = fd.cpu().pin_memory()
od = [torch.empty(4, 300, 6, device="cuda") for _ in range(4)]; oh = [torch.empty(4, 300, 6).pin_memory() for _ in range(4)]
xc = torch.rand(2, 3, CLIP_TINY.image_size, CLIP_TINY.image_size, device="cuda") * 2 - 1
ec = [torch.empty(2, CLIP_TINY.embed, device="cuda") for _ in range(3)]
def flight_round(n):
    for depth in (1, 3, 2, 4):
        m2.set_in_flight(depth)
        ts = [m2.submit(fh if k & 1 else fd, oh[k % depth] if k & 1 else od[k % depth]) for k in range(n)]
        for t in ts[-depth:]:
            m2.wait(t, host=True)
    clip.set_in_flight(3)
    ts = [clip.submit_image(xc, ec[k % 3]) for k in range(n)]
    for t in ts[-3:]:
        clip.wait(t, host=True)
    clip.set_in_flight(1)
    torch.cuda.synchronize()
_orig_round = one_round
part = sys.argv[1] if len(sys.argv) > 1 else "all"                  # all | entry (every entry point) | flight (depth changes, submissions)
def one_round(n):
    if part in ("all", "entry"): _orig_round(n)
    if part in ("all", "flight"): flight_round(n)
one_round(5); base = free(); t0 = time.time()
for r in range(6):
    one_round(40); print(f"round {r}: free {free():.0f} MB (drift {base - free():+.1f} MB), index rows {len(ix)}", flush=True)
print("seconds", round(time.time() - t0, 1), "final drift MB", round(base - free(), 1))

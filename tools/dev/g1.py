import numpy as np, time, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from clearcam_amd.yolov9 import YOLOv9
from clearcam_amd.weights import synthetic_yolov9_state_dict
from oracle.yolov9_oracle import YOLOv9Oracle, match_detections
size = sys.argv[1] if len(sys.argv) > 1 else "c"
sd = synthetic_yolov9_state_dict(size, 1234)
o = YOLOv9Oracle(size, 640, sd)
fr = np.random.default_rng(1).integers(0,256,(2,640,640,3),dtype=np.uint8)
with torch.no_grad():
    x = o.network_input(fr); feats = o.features(x); raw = o.head_raw(feats); dec = o.decode(raw)
ref = o.detect_batch(fr)
for dt in ("f32","f16","bf16"):
    m = YOLOv9(size, 640, state_dict=sd, dtype=dt)
    out = m.detect_batch(fr)
    inp = m.get_tensor("input")
    print(dt, "input err", np.abs(inp - x.permute(0,2,3,1).numpy()).max())
    for n,f in zip(("p3","p4","p5"), feats):
        g = m.get_tensor(n); r = f.permute(0,2,3,1).numpy()
        print("  ", n, "max abs err %.3g  rel(rms) %.3g" % (np.abs(g-r).max(), np.sqrt(((g-r)**2).mean())/np.sqrt((r**2).mean())))
    for i in range(3):
        g = m.get_tensor("raw%d"%i); r = raw[i].permute(0,2,3,1).numpy()
        print("   raw%d max abs err %.3g rel %.3g" % (i, np.abs(g-r).max(), np.sqrt(((g-r)**2).mean())/np.sqrt((r**2).mean())))
    for b in range(2):
        print("   match", match_detections(ref[b], out[b], 0.5))
    if dt == "f32":
        print("   exact rows equal:", [(np.abs(ref[b]-out[b]).max()) for b in range(2)])
    t=time.time(); 
    for _ in range(5): m.detect_batch(fr)
    print("   ms/call B=2:", (time.time()-t)/5*1e3, "gpu ms", m.last_gpu_ms())

# dev tool (GPU): the anchor with the largest box error against the f32 oracle, per storage mode (conditioned checkpoint, un-rounded weights, the
# 64 frames of tests/test_gpu_yolo.py::test_detect_split_weight_mode_with_unrounded_weights); saves that frame's decoded rows for CPU analysis
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from clearcam_amd.weights import conditioned_yolov9_state_dict
from clearcam_amd.yolov9 import YOLOv9
from oracle.yolov9_oracle import YOLOv9Oracle, decoded_rows
sd = conditioned_yolov9_state_dict("c", 1234, exact=False)
fr = np.random.default_rng(1).integers(0, 256, (64, 640, 640, 3), dtype=np.uint8)
o = YOLOv9Oracle("c", 640, sd); dec = []; p5 = []
with torch.no_grad():
    for i in range(0, 64, 4):
        f = o.features(o.network_input(fr[i:i + 4])); p5.append(f[2].permute(0, 2, 3, 1).numpy())
        dec.append(decoded_rows(o.decode(o.head_raw(f))))
dref = np.concatenate(dec); p5 = np.concatenate(p5)
out = {}
for dt in ("f16s", "f16h", "f16", "f32"):
    m = YOLOv9("c", 640, state_dict=sd, dtype=dt)
    m.detect_batch(fr); d = m.get_tensor("decoded"); g5 = m.get_tensor("p5"); m.close()
    both = (d[..., 4] > 0) & (dref[..., 4] > 0)
    e = np.where(both, np.abs(d[..., :4] - dref[..., :4]).max(-1), 0)
    order = np.argsort(-e, axis=None)[:6]
    rel = np.sqrt(((g5 - p5) ** 2).mean((1, 2, 3)) / (p5 ** 2).mean((1, 2, 3)))
    print(dt, "P5 rel rms per frame: max %.2e at %d, median %.2e" % (rel.max(), rel.argmax(), np.median(rel)))
    for k in order:
        fi, ai = np.unravel_index(k, e.shape)
        print(f"  {dt}: err {e[fi, ai]:.3f} px frame {fi} anchor {ai} ref {dref[fi, ai].round(2)} got {d[fi, ai].round(2)}")
    fi = np.unravel_index(order[0], e.shape)[0]
    out[dt + "_frame"] = np.array(fi); out[dt + "_decoded"] = d[fi]; out[dt + "_ref"] = dref[fi]
np.savez_compressed("gpurun_out/outlier_probe.npz", **out)

"""Generate tests/golden/*.npz from the CPU oracle (seeded weights + seeded frames).

The reference itself cannot run here (tinygrad/cv2/weights absent), so these vectors pin
the ORACLE (guarding it against drift) and give the GPU tests a fixture that does not need the oracle's
forward pass.  Reference-owned golden data (test/clip_images/embeddings.pkl, the tokenizer KATs) is
restated in tests/golden/reference_pins.json by hand with file:line provenance.
Run from the repo root:  python tools/make_golden.py
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clearcam_amd.weights import synthetic_yolov9_state_dict  # noqa: E402
from oracle.yolov9_oracle import YOLOv9Oracle, letterbox  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def yolo_case(name, size, res, seed, shape):
    sd = synthetic_yolov9_state_dict(size, 1234)
    o = YOLOv9Oracle(size, res, sd)
    frames = np.random.default_rng(seed).integers(0, 256, shape, dtype=np.uint8)
    with torch.no_grad():
        x = o.network_input(frames)
        feats = o.features(x)
        raw = o.head_raw(feats)
    det = o.detect_batch(frames)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), size=size, res=res, seed=seed, shape=np.array(shape),
                        det=det, p3_mean=np.array([float(f.mean()) for f in feats]),
                        p3_std=np.array([float(f.std()) for f in feats]),
                        raw0_corner=raw[0][:, :, :4, :4].permute(0, 2, 3, 1).numpy().astype(np.float32))
    print(name, "detections per frame:", (det[..., 4] > 0).sum(1).tolist())


def letterbox_case():
    f = np.random.default_rng(3).integers(0, 256, (135, 240, 3), dtype=np.uint8)
    lb = letterbox(f, 160)
    lbf = letterbox(f.astype(np.float32), 160)
    np.savez_compressed(os.path.join(OUT, "letterbox_135x240_to_160.npz"), u8=lb, f32=lbf)
    print("letterbox", lb.shape)


def face_cases():
    """Face path: AdaFace embedding, BlazeFace detections, OpenCV-compatible resize / warp, cubic crop preprocessing."""
    from clearcam_amd.weights import synthetic_adaface_state_dict, synthetic_blazeface_state_dict
    from oracle.adaface_oracle import AdaFaceOracle
    from oracle.blazeface_oracle import BlazeFaceOracle
    from oracle import cv_resize_oracle, cv_warp_oracle
    rng = np.random.default_rng(21)
    face = rng.integers(0, 256, (112, 112, 3), dtype=np.uint8)
    img = rng.integers(0, 256, (360, 480, 3), dtype=np.uint8)
    crop = rng.integers(0, 256, (57, 131, 3), dtype=np.uint8)
    M = cv_warp_oracle.get_rotation_matrix_2d((65.5, 28.0), 17.5, 1.3)
    np.savez_compressed(os.path.join(OUT, "face_path.npz"),
                        face=face, adaface=AdaFaceOracle(synthetic_adaface_state_dict(777))(face),
                        img=img, blazeface=BlazeFaceOracle(synthetic_blazeface_state_dict(555))(img),
                        crop=crop, crop_cubic_224=cv_resize_oracle.resize_cubic_u8(crop, 224),
                        crop_linear_200x90=cv_warp_oracle.resize_linear_u8(crop, (200, 90)),
                        warp_M=M, crop_warp_150x80=cv_warp_oracle.warp_affine_u8(crop, M, (150, 80)))
    print("face_path fixtures written")


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    if "--face-only" in sys.argv:
        face_cases(); sys.exit(0)
    yolo_case("yolo_t_640", "t", 640, 1, (2, 640, 640, 3))
    yolo_case("yolo_t_640_from_540x960", "t", 640, 6, (1, 540, 960, 3))
    yolo_case("yolo_c_640", "c", 640, 1, (1, 640, 640, 3))
    letterbox_case()
    face_cases()

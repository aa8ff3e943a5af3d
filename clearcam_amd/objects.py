"""Host-side mirror of ``models/objects.py``: ``OpenCLIP`` (image + text towers) and ``ObjectFinder``
(embedding store + search), same names / signatures / return conventions, compute in libclearcam_hip.

    finder = ObjectFinder(); finder.init_clip(state_dict=sd)
    emb = jit_infer(finder.model.precompute_embedding, Tensor(img).unsqueeze(0), finder.jit_cache).numpy()   # clearcam.py:1285
    q   = finder.model._encode_text("white van").numpy()                                                     # clearcam.py:667
    hits = finder.search(query="white van", top_k=10)                                                         # clearcam.py:669

There is no CPU path: without the HIP library / a GPU every call raises.
"""
from __future__ import annotations

import ctypes as C
import os
import pickle
from typing import Dict, List, Optional, Tuple

import numpy as np

from . import _lib
from .arch import CLIP_L14, ClipArch
from .helpers import Tensor, as_numpy
from .weights import load_safetensors
from .yolov9 import DTYPES


def event_img_info(image: str) -> dict:
    """``clearcam.py:1193``: '<ts>_<track id>_<class id>' crop file stem."""
    p = image.split("_")
    return {"ts": int(float(p[0])), "object_id": int(p[1]), "class_id": int(p[2])}


class OpenCLIP:
    """``models/objects.py:21-186``.  ``state_dict`` replaces the HuggingFace download (:91)."""

    def __init__(self, base_path: str = "data/cameras", state_dict: Optional[Dict[str, np.ndarray]] = None,
                 weights: Optional[str] = None, arch: ClipArch = CLIP_L14, dtype: str = "bf16", device: int = 0,
                 tokenizer=None):
        self.base_path, self.arch, self.dtype, self.device = base_path, arch, dtype, device
        self._tokenizer = tokenizer
        if state_dict is None:
            path = weights or os.path.join(os.environ.get("CLEARCAM_WEIGHTS_DIR", "weights"),
                                           "CLIP-ViT-L-14-laion2B-s32B-b82K.safetensors")
            if not os.path.exists(path):
                raise FileNotFoundError(f"{path} not found: pass state_dict= or weights= (the reference downloads the "
                                        "checkpoint from HuggingFace; there is no network here)")
            state_dict = load_safetensors(path)
        L = _lib.lib()
        cfg = _lib.ClipConfig(*[getattr(arch, f) for f, _ in _lib.ClipConfig._fields_])
        self._h = C.c_void_p()
        _lib.check(L.cc_clip_create(C.byref(self._h), C.byref(cfg), DTYPES[dtype], device))
        for name, arr in state_dict.items():
            if name == "attn_mask":            # a Tensor attribute in the reference (:76); recomputed in-kernel
                continue
            a = np.ascontiguousarray(arr, dtype=np.float32)
            shp = (C.c_int64 * max(a.ndim, 1))(*a.shape)
            _lib.check(L.cc_clip_load(self._h, name.encode(), _lib.ptr(a), shp, a.ndim))
        _lib.check(L.cc_clip_finalize(self._h))

    @property
    def tokenizer(self):
        if self._tokenizer is None:
            from .clip_tokenizer import SimpleTokenizer
            self._tokenizer = SimpleTokenizer()
        return self._tokenizer

    # -- reference surfaces ---------------------------------------------------------------------------
    def precompute_embedding(self, x) -> Tensor:
        """:94-133 — (B,3,224,224) float32 -> Tensor (B,768), unit norm."""
        a = np.ascontiguousarray(as_numpy(x), dtype=np.float32)
        s = self.arch.image_size
        if a.ndim != 4 or a.shape[1:] != (3, s, s):
            raise ValueError(f"expected (B,3,{s},{s}), got {a.shape}")
        out = np.empty((a.shape[0], self.arch.embed), np.float32)
        _lib.check(_lib.lib().cc_clip_encode_image(self._h, _lib.ptr(a), a.shape[0], 0, _lib.ptr(out), 0, None))
        return Tensor(out)

    def precompute_embedding_device(self, x, out):
        """Device-resident variant (CUDA torch tensors in and out, no host copies) for benchmarks."""
        import torch
        s = torch.cuda.current_stream(x.device).cuda_stream
        _lib.check(_lib.lib().cc_clip_encode_image(self._h, _lib.ptr(x), x.shape[0], 1, _lib.ptr(out), 1, C.c_void_p(s)))
        return out

    def encode_tokens(self, tokens) -> np.ndarray:
        """Batch form of ``encode_text`` (:145-186): (B,77) int -> (B,768) float32."""
        t = np.ascontiguousarray(as_numpy(tokens), dtype=np.int32)
        if t.ndim != 2 or t.shape[1] != self.arch.t_ctx:
            raise ValueError(f"expected (B,{self.arch.t_ctx}) tokens, got {t.shape}")
        out = np.empty((t.shape[0], self.arch.embed), np.float32)
        _lib.check(_lib.lib().cc_clip_encode_text(self._h, _lib.ptr(t), t.shape[0], _lib.ptr(out), 0, None))
        return out

    def _encode_text(self, query: str, realize: bool = False):
        """:135-143 — str -> Tensor (768,) (ndarray if realize)."""
        emb = self.encode_tokens(self.tokenizer.tokens_for_model(query))[0]
        return emb if realize else Tensor(emb)

    def last_gpu_ms(self) -> float:
        ms = C.c_float()
        _lib.check(_lib.lib().cc_clip_last_gpu_ms(self._h, C.byref(ms)))
        return ms.value

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            _lib.lib().cc_clip_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class EmbeddingIndex:
    """Device-resident (N,dim) float32 matrix with scan + top-k (cc_index_*)."""

    def __init__(self, dim: int = 768, capacity: int = 1 << 20, device: int = 0):
        self.dim, self.capacity, self.device = dim, capacity, device
        self._h = C.c_void_p()
        _lib.check(_lib.lib().cc_index_create(C.byref(self._h), dim, capacity, device))

    def __len__(self) -> int:
        n = C.c_int64()
        _lib.check(_lib.lib().cc_index_size(self._h, C.byref(n)))
        return n.value

    def add(self, emb) -> None:
        on_dev = bool(getattr(emb, "is_cuda", False))
        a = emb.contiguous().float() if on_dev else np.ascontiguousarray(as_numpy(emb), dtype=np.float32).reshape(-1, self.dim)
        _lib.check(_lib.lib().cc_index_add(self._h, _lib.ptr(a), a.shape[0], int(on_dev)))

    def scores(self, q) -> np.ndarray:
        qa = np.ascontiguousarray(as_numpy(q), dtype=np.float32).reshape(-1, self.dim)
        out = np.empty((qa.shape[0], len(self)), np.float32)
        if len(self):
            _lib.check(_lib.lib().cc_index_scores(self._h, _lib.ptr(qa), qa.shape[0], _lib.ptr(out), 0, None))
        return out

    def search(self, q, k: int) -> Tuple[np.ndarray, np.ndarray]:
        """(Q,dim) queries -> (idx (Q,k) int32, score (Q,k) f32), descending, ties by lower row id; -1/-inf past N."""
        qa = np.ascontiguousarray(as_numpy(q), dtype=np.float32).reshape(-1, self.dim)
        idx = np.empty((qa.shape[0], k), np.int32)
        sc = np.empty((qa.shape[0], k), np.float32)
        _lib.check(_lib.lib().cc_index_search(self._h, _lib.ptr(qa), qa.shape[0], k, _lib.ptr(idx), _lib.ptr(sc), 0, None))
        return idx, sc

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            _lib.lib().cc_index_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def preprocess_crops(crops, size: int = 224, device: int = 0):
    """cc_crop_preprocess over a list of (H,W,3) uint8 arrays (host) -> torch device tensor (B,3,size,size) f32."""
    import torch
    crops = [np.ascontiguousarray(c, dtype=np.uint8) for c in crops]
    for c in crops:
        if c.ndim != 3 or c.shape[2] != 3 or c.size == 0:
            raise ValueError(f"crop must be a non-empty (H,W,3) uint8 array, got {c.shape}")
    B = len(crops)
    if B == 0:
        raise ValueError("no crops")
    sizes = np.array([c.size for c in crops], np.int64)
    offsets = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
    packed = np.concatenate([c.reshape(-1) for c in crops])
    hs = np.array([c.shape[0] for c in crops], np.int32)
    ws = np.array([c.shape[1] for c in crops], np.int32)
    out = torch.empty((B, 3, size, size), dtype=torch.float32, device=torch.device("cuda", device))
    _lib.check(_lib.lib().cc_crop_preprocess(_lib.ptr(packed), _lib.ptr(offsets), _lib.ptr(hs), _lib.ptr(ws), B, 0, size, _lib.ptr(out), device,
                                   torch.cuda.current_stream(device).cuda_stream))
    return out


class ObjectFinder:
    """``models/objects.py:188-422`` minus the face pipeline (out of scope)."""

    def __init__(self, base_path: str = "data/cameras"):
        self.base_path = base_path
        self.image_embeddings: Dict[str, np.ndarray] = {}
        self.face_embeddings: Dict[str, np.ndarray] = {}
        self.clip = False
        self.jit_cache: dict = {}
        self.model: Optional[OpenCLIP] = None
        self._index: Optional[EmbeddingIndex] = None
        self._index_paths: List[str] = []
        self._index_src = None

    def init_clip(self, **clip_kwargs):
        """:199-206 (warm-up included: the first call per shape builds and captures the hipGraph)."""
        if self.clip:
            return
        self.clip = True
        self.model = OpenCLIP(**clip_kwargs)
        s = self.model.arch.image_size
        for _ in range(2):
            self.model.precompute_embedding(np.random.rand(1, 3, s, s).astype(np.float32))

    def turn_off_clip(self):
        self.clip = False
        if self.model is not None:
            self.model.close()
        self.model = None

    # -- face models (:213-222): BlazeFace detector + AdaFace embedder on the same engine ------------------------
    def init_face(self, blazeface_kwargs: Optional[dict] = None, adaface_kwargs: Optional[dict] = None):
        """:213-217.  `self.blazeface(Tensor(img)).numpy()` -> (896,17) and `self.adaface(Tensor(face112)).numpy()` -> (1,512)
        are the reference's two face call surfaces (objects.py:254, clearcam.py:674,1236); `img_to_face` joins them."""
        if getattr(self, "face", False):
            return
        from .adaface import ADAFACE
        from .blazeface import BlazeFace
        self.blazeface = BlazeFace(**(blazeface_kwargs or {}))
        self.adaface = ADAFACE(**(adaface_kwargs or {}))
        self.face = True

    def turn_off_face(self):
        """:219-222"""
        for name in ("blazeface", "adaface"):
            m = getattr(self, name, None)
            if m is not None:
                m.close()
            setattr(self, name, None)
        self.face = False

    # -- face crop alignment (:243-354) ------------------------------------------------------------------------
    EYE_LEFT, EYE_RIGHT, FACE_SIZE = (38.0, 51.0), (73.0, 51.0), 112      # where the eyes land in the aligned face

    def img_to_face(self, orig):
        """RGB crop -> aligned 112x112 face (channel-swapped like the reference's final cvtColor) or None.  Steps as in the
        reference: letterbox to 640 and run BlazeFace; first surviving detection; reject faces narrower than 50 px; cut a
        square of twice the face size around the eyes' midpoint; rotate it so the eyes are level; scale and shift so they
        land on EYE_LEFT / EYE_RIGHT.  The pixel operations are clearcam_amd.cvops (GPU, OpenCV-compatible)."""
        from . import cvops
        full = np.ascontiguousarray(orig)
        h, w = full.shape[:2]
        scale = 640 / max(h, w)
        dev = self.blazeface.device                               # pixel ops on the GPU that runs the face models
        small = cvops.resize_linear(full, (int(w * scale), int(h * scale)), dev)
        gap_w, gap_h = 640 - small.shape[1], 640 - small.shape[0]
        top, left = gap_h // 2, gap_w // 2
        boxed = cvops.copy_make_border(small, top, gap_h - top, left, gap_w - left)
        found = self.blazeface(Tensor(boxed)).numpy()
        found = found[found[:, 0] != 0]
        if found.shape[0] == 0:
            return None
        first = found[0]
        offset = np.array([left, top])
        y1, x1, y2, x2 = ((float(first[0]) - top) / scale, (float(first[1]) - left) / scale,
                          (float(first[2]) - top) / scale, (float(first[3]) - left) / scale)
        eye_a = (np.array([first[4], first[5]]) - offset) / scale
        eye_b = (np.array([first[6], first[7]]) - offset) / scale
        if (x2 - x1) < 50:
            return None
        goal_a, goal_b = np.array(self.EYE_LEFT), np.array(self.EYE_RIGHT)
        mid = (eye_a + eye_b) / 2
        tilt = np.degrees(np.arctan2(eye_b[1] - eye_a[1], eye_b[0] - eye_a[0]))
        side = max(x2 - x1, y2 - y1) * 2.0
        H, W = full.shape[:2]
        cx1, cy1 = max(0, int(mid[0] - side / 2)), max(0, int(mid[1] - side / 2))
        cx2, cy2 = min(W, int(mid[0] + side / 2)), min(H, int(mid[1] + side / 2))
        if cx2 <= cx1 or cy2 <= cy1:
            return None
        cut = full[cy1:cy2, cx1:cx2]
        ch, cw = cut.shape[:2]
        corner = np.array([cx1, cy1])
        rot = cvops.rotation_matrix_2d((cw / 2, ch / 2), float(tilt), 1.0)
        c, s_ = abs(rot[0, 0]), abs(rot[0, 1])
        out_w, out_h = int(ch * s_ + cw * c), int(ch * c + cw * s_)
        rot[0, 2] += out_w / 2 - cw / 2
        rot[1, 2] += out_h / 2 - ch / 2
        level = cvops.warp_affine(cut, rot, (out_w, out_h), dev)
        a_rot = rot[:, :2] @ (eye_a - corner) + rot[:, 2]
        b_rot = rot[:, :2] @ (eye_b - corner) + rot[:, 2]
        zoom = np.linalg.norm(goal_b - goal_a) / np.linalg.norm(b_rot - a_rot)
        place = np.array([[zoom, 0, goal_a[0] - a_rot[0] * zoom], [0, zoom, goal_a[1] - a_rot[1] * zoom]], np.float32)
        face = cvops.warp_affine(level, place, (self.FACE_SIZE, self.FACE_SIZE), dev)
        return np.ascontiguousarray(face[:, :, ::-1])

    def preprocess_face(self, img):
        """:232-241 for an already decoded RGB array: a 112x112x3 image is taken as is, anything else is aligned."""
        img = np.asarray(img)
        return img if img.shape == (112, 112, 3) else self.img_to_face(img)

    def preprocess(self, img):
        """:237-242 — cv2.resize(img,(224,224),INTER_CUBIC) -> f32/255 -> (x-0.5)/0.5 -> CHW, on the GPU
        (cc_crop_preprocess: OpenCV's 8-bit fixed-point cubic).  Returns (3,224,224) float32 like the reference."""
        return self.preprocess_crops([img]).cpu().numpy()[0]

    def preprocess_crops(self, crops, size: int = 224, device: Optional[int] = None):
        """Batch form of :237-242: list of (H,W,3) uint8 crops of any size -> device tensor (B,3,size,size) float32,
        the input `OpenCLIP.precompute_embedding_device` takes.  One H2D copy of the packed crops, one kernel."""
        return preprocess_crops(crops, size, self.model.device if (device is None and self.model) else (device or 0))

    # -- store ------------------------------------------------------------------------------------
    def _load_all_embeddings(self, face: bool = False):
        """:392-422 — merge every per-day embeddings.pkl under base_path, drop stale keys."""
        valid, target = set(), (self.face_embeddings if face else self.image_embeddings)
        if os.path.isdir(self.base_path):
            for cam in os.listdir(self.base_path):
                objects = os.path.join(self.base_path, cam, "faces" if face else "objects")
                if not os.path.isdir(objects):
                    continue
                for day in os.listdir(objects):
                    f = os.path.join(objects, day, "embeddings.pkl")
                    if not os.path.exists(f):
                        continue
                    with open(f, "rb") as fh:
                        emb = pickle.load(fh).get("embeddings", {})
                    valid.update(emb.keys())
                    target.update(emb)
        for k in set(target) - valid:
            del target[k]

    # -- append-only store (clearcam_amd/store.py) instead of one pickle per folder ---------------------------
    def attach_store(self, capacity: Optional[int] = None) -> int:
        """Load every stored crop under base_path (store files and/or the reference's pickles) straight into the device
        matrix: no per-crop dict entries, one H2D copy.  Subsequent `search` calls scan it; `add_embedding` appends.
        Returns the number of rows."""
        from .store import load_all
        paths, rows = load_all(self.base_path, 768)
        if self._index is not None:
            self._index.close()
        dim = rows.shape[1] if len(paths) else 768
        self._index = EmbeddingIndex(dim, max(capacity or 0, 2 * len(paths), 1024), device=self.model.device if self.model else 0)
        if len(paths):
            self._index.add(rows)
        self._index_paths, self._index_src, self._attached = list(paths), "store", True
        return len(paths)

    def add_embedding(self, path: str, emb) -> None:
        """What clearcam.py:1282-1287 does per crop (load pickle, insert, rewrite pickle) as one appended row: on disk in
        the crop's day folder, in the dict the reference exposes, and in the attached device matrix."""
        from .store import EmbeddingStore
        e = np.asarray(as_numpy(emb), np.float32).reshape(1, -1)
        EmbeddingStore(os.path.dirname(path), e.shape[1]).append([path], e)
        self.image_embeddings[path] = e
        if getattr(self, "_attached", False):
            self._index.add(e)
            self._index_paths.append(path)

    def _device_index(self, embeddings: Dict[str, np.ndarray]) -> EmbeddingIndex:
        """(Re)build the HBM-resident matrix when the dict changed (the reference reloads before every search)."""
        sig = (id(embeddings), len(embeddings))
        if self._index is None or self._index_src != sig:
            if self._index is not None:
                self._index.close()
            paths = [p for p, e in embeddings.items() if e is not None]
            dim = np.asarray(embeddings[paths[0]]).size if paths else 768
            self._index = EmbeddingIndex(dim, max(len(paths), 1), device=self.model.device if self.model else 0)
            if paths:
                self._index.add(np.stack([np.asarray(embeddings[p], np.float32).reshape(-1) for p in paths]))
            self._index_paths, self._index_src = paths, sig
        return self._index

    # -- search (:356-390) ------------------------------------------------------------------------------
    def search(self, query=None, top_k=10, cam_name=None, timestamp=None, text_embedding=None, is_face=False):
        embeddings = self.face_embeddings if is_face else self.image_embeddings
        attached = getattr(self, "_attached", False) and not is_face and len(self._index_paths) > 0
        if not embeddings and not attached:
            print("No embeddings available.")
            return []
        if text_embedding is None:
            text_embedding = self.model._encode_text(query).numpy()
        q = np.asarray(as_numpy(text_embedding), np.float32).reshape(-1)
        index = self._index if attached else self._device_index(embeddings)
        scores = index.scores(q)[0]                         # one HBM pass instead of a Python loop of N dot products
        sims = []
        for path, sim in zip(self._index_paths, scores):
            norm = path.replace("\\", "/")
            if cam_name and f"/cameras/{cam_name}/" not in norm:
                continue
            if timestamp and f"/objects/{timestamp}/" not in norm and "/objects/video/" not in norm:
                continue
            fn = os.path.basename(path)
            if fn.lower().endswith(".jpg"):
                oid = event_img_info(fn.split(".jpg")[0])["object_id"] if "_" in fn else None
                sims.append((path, float(sim), oid))
        if any(s[2] for s in sims):
            best = {}
            for path, score, oid in sims:
                if oid is not None and (oid not in best or score > best[oid][1]):
                    best[oid] = (path, score)
            results = list(best.values()) + [(p, s) for p, s, o in sims if o is None]
        else:
            results = [(p, s) for p, s, _ in sims]
        results.sort(key=lambda x: x[1], reverse=True)
        return results[:top_k]

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

    # -- batches in flight (many small batches: the reference encodes one crop per call, :356-363) -----------------------------------
    def set_in_flight(self, n: int) -> None:
        """n slots (own stream, buffers and graph each) for submit_image(); results are bit-identical to precompute_embedding."""
        _lib.check(_lib.lib().cc_clip_set_in_flight(self._h, n))

    def submit_image(self, x, out) -> int:
        """Queue one (B,3,224,224) float32 batch -> `out` (B,768) float32 on the next slot; CUDA or PINNED-host torch tensors.
        Device input is taken as ready on the current torch stream, which does not wait for the result - wait(ticket) does."""
        import torch
        if not x.is_contiguous() or not out.is_contiguous() or x.dtype != torch.float32 or out.dtype != torch.float32:
            raise ValueError("x and out must be contiguous float32 tensors")
        if (not x.is_cuda and not x.is_pinned()) or (not out.is_cuda and not out.is_pinned()):
            raise ValueError("host tensors handed to submit_image() must be pinned")
        t = C.c_longlong()
        s = torch.cuda.current_stream(x.device).cuda_stream if x.is_cuda else 0
        _lib.check(_lib.lib().cc_clip_submit_image(self._h, _lib.ptr(x), x.shape[0], int(x.is_cuda), _lib.ptr(out), int(out.is_cuda),
                                                   C.c_void_p(s), C.byref(t)))
        return t.value

    def wait(self, ticket: int, host: bool = False) -> None:
        """Make the current torch stream (host=True: the calling thread) wait for a submission's embeddings."""
        import torch
        s = None if host else C.c_void_p(torch.cuda.current_stream().cuda_stream)
        _lib.check(_lib.lib().cc_clip_wait(self._h, ticket, s))

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
    """Device-resident (N,dim) matrix with scan + top-k (cc_index_*).  storage "f32" (exact f32 scores) or "bf16" (half the
    bytes per scan, scores within ~1e-3 for unit vectors).  `capacity` is the initial allocation; the matrix grows."""

    def __init__(self, dim: int = 768, capacity: int = 1 << 20, device: int = 0, storage: str = "f32"):
        self.dim, self.device, self.storage = dim, device, storage
        self._h = C.c_void_p()
        _lib.check(_lib.lib().cc_index_create_ex(C.byref(self._h), dim, max(int(capacity), 1), device, {"f32": 0, "bf16": 2}[storage]))

    @property
    def capacity(self) -> int:
        c = C.c_int64()
        _lib.check(_lib.lib().cc_index_info(self._h, C.byref(c), None, None))
        return c.value

    def __len__(self) -> int:
        n = C.c_int64()
        _lib.check(_lib.lib().cc_index_size(self._h, C.byref(n)))
        return n.value

    def add(self, emb, groups=None) -> None:
        """Append rows: (n,dim) / (n,1,dim) / (dim,) float32, host array or CUDA torch tensor.  Rows of any other width are
        rejected (a 512-d face row must never land in the 768-d crop matrix).  groups: one int per row (search filter unit)."""
        on_dev = bool(getattr(emb, "is_cuda", False))
        if on_dev:
            a = emb.contiguous().float()
            if a.shape[-1] != self.dim:
                raise ValueError(f"expected {self.dim}-d rows, got {tuple(a.shape)}")
            a = a.reshape(-1, self.dim)
        else:
            a = np.asarray(as_numpy(emb), dtype=np.float32)
            if a.ndim == 0 or a.shape[-1] != self.dim:
                raise ValueError(f"expected {self.dim}-d rows, got {a.shape}")
            a = np.ascontiguousarray(a.reshape(-1, self.dim))
        g = None
        if groups is not None:
            g = np.ascontiguousarray(np.asarray(groups, np.int32).reshape(-1))
            if g.shape[0] != a.shape[0] or (g.size and (g.min() < 0 or g.max() >= 1 << 24)):
                raise ValueError("groups: one id in [0, 2^24) per row")
        _lib.check(_lib.lib().cc_index_add_grouped(self._h, _lib.ptr(a), a.shape[0], int(on_dev), _lib.ptr(g) if g is not None else None))

    def scores(self, q) -> np.ndarray:
        qa = np.ascontiguousarray(as_numpy(q), dtype=np.float32).reshape(-1, self.dim)
        out = np.empty((qa.shape[0], len(self)), np.float32)
        if len(self):
            _lib.check(_lib.lib().cc_index_scores(self._h, _lib.ptr(qa), qa.shape[0], _lib.ptr(out), 0, None))
        return out

    def search(self, q, k: int, allowed=None) -> Tuple[np.ndarray, np.ndarray]:
        """(Q,dim) queries -> (idx (Q,k) int32, score (Q,k) f32), descending, ties by lower row id; -1/-inf past N.
        allowed: optional uint8 array, one entry per group id — only rows of groups with a non-zero entry compete."""
        qa = np.ascontiguousarray(as_numpy(q), dtype=np.float32).reshape(-1, self.dim)
        idx = np.empty((qa.shape[0], k), np.int32)
        sc = np.empty((qa.shape[0], k), np.float32)
        al = None if allowed is None else np.ascontiguousarray(allowed, dtype=np.uint8)
        _lib.check(_lib.lib().cc_index_search_groups(self._h, _lib.ptr(qa), qa.shape[0], k, _lib.ptr(al) if al is not None else None,
                                                     0 if al is None else int(al.size), _lib.ptr(idx), _lib.ptr(sc), 0, None))
        return idx, sc

    def search_device(self, q, k: int):
        """The same with the result left on the GPU: (idx int32, score f32) CUDA torch tensors on the current torch stream,
        for callers that keep going on the device (clearcam_amd.dist.ShardedIndex: all-gather + merge)."""
        import torch
        dev = torch.device("cuda", self.device)
        if not bool(getattr(q, "is_cuda", False)):
            # on_device=1 tells the library that BOTH the queries and the outputs are device pointers: a host query is uploaded
            # here first, on the caller's stream (a host pointer handed to a device-to-device copy only works by accident of
            # ROCm's pointer inference, and the temporary could be freed while the copy is pending)
            q = torch.as_tensor(np.ascontiguousarray(as_numpy(q), dtype=np.float32)).to(dev)
        qa = q.contiguous().float().reshape(-1, self.dim)
        idx = torch.empty((qa.shape[0], k), dtype=torch.int32, device=dev)
        sc = torch.empty((qa.shape[0], k), dtype=torch.float32, device=dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        _lib.check(_lib.lib().cc_index_search_groups(self._h, _lib.ptr(qa), qa.shape[0], k, None, 0, _lib.ptr(idx), _lib.ptr(sc), 1, C.c_void_p(stream)))
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


class _RowTable:
    """Host-side description of the rows of one device matrix, built ONCE when rows are added, so that a search costs
    O(groups + candidates) Python instead of the reference's O(N) loop (models/objects.py:365-376).

    The reference filters a crop by two substring tests on its '/'-normalised path — f"/cameras/{cam}/" and
    f"/objects/{day}/" or "/objects/video/" (:368-371).  Both patterns end in '/', so they can only match inside the
    directory part of the path: rows are grouped by directory string and the tests run once per group.  Group 0 holds the
    rows the reference never returns (file name not *.jpg, :374)."""

    def __init__(self):
        self.paths: List[str] = []
        self.dirs: List[str] = [""]            # group id -> directory part (group 0: never allowed)
        self.dir_id: Dict[str, int] = {}
        self.truthy: List[bool] = [False]      # group has a crop whose track id is truthy (the `any(item[2] ...)` of :378)
        self.bad: List[bool] = [False]         # group has a *.jpg name event_img_info cannot parse (the reference raises on it)
        self.group: List[int] = []
        self.oid: List[Optional[int]] = []     # track id per row exactly as event_img_info returns it (any sign), None = no '_' in the name

    def describe(self, path: str) -> Tuple[int, int]:
        norm = path.replace("\\", "/")
        fn = os.path.basename(path)
        if not fn.lower().endswith(".jpg"):
            return 0, None
        d = norm[:norm.rfind("/") + 1]
        g = self.dir_id.get(d)
        if g is None:
            g = self.dir_id[d] = len(self.dirs)
            self.dirs.append(d); self.truthy.append(False); self.bad.append(False)
        oid = None
        if "_" in fn:
            try:
                oid = event_img_info(fn.split(".jpg")[0])["object_id"]
            except Exception:                  # noqa: BLE001  names the reference raises on take the exact slow path at search time
                self.bad[g] = True
                oid = None
        if oid:                                # the reference's `any(item[2] ...)`: id 0 and None are falsy
            self.truthy[g] = True
        return g, oid

    def extend(self, paths) -> np.ndarray:
        groups = np.empty(len(paths), np.int32)
        for i, p in enumerate(paths):
            g, oid = self.describe(p)
            groups[i] = g
            self.group.append(g); self.oid.append(oid)
        self.paths.extend(paths)
        return groups

    def allowed(self, cam_name, timestamp) -> np.ndarray:
        al = np.zeros(len(self.dirs), np.uint8)
        for g in range(1, len(self.dirs)):
            d = self.dirs[g]
            if cam_name and f"/cameras/{cam_name}/" not in d:
                continue
            if timestamp and f"/objects/{timestamp}/" not in d and "/objects/video/" not in d:
                continue
            al[g] = 1
        return al


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
        # one device matrix + row table per source: "store" (attach_store), "image" / "face" (the reference's dicts).
        # They never share state: a face search cannot replace the attached crop matrix, and 512-d rows cannot enter it.
        self._dev: Dict[str, Tuple[EmbeddingIndex, _RowTable]] = {}
        self._dev_sig: Dict[str, tuple] = {}
        self._version = {"image": 0, "face": 0}
        self._stores: Dict[str, object] = {}
        self.index_storage = "f32"             # "bf16" halves the scan bytes (scores within ~1e-3)

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
        """:392-422 — merge every per-day embeddings.pkl under base_path, drop stale keys.  Folders written by
        `add_embedding` (append-only store files, clearcam_amd/store.py) are read as well, so the rows this class itself
        stored are not mistaken for stale keys."""
        from .store import EmbeddingStore
        valid, target = set(), (self.face_embeddings if face else self.image_embeddings)
        key = "face" if face else "image"
        files = []
        if os.path.isdir(self.base_path):
            for cam in os.listdir(self.base_path):
                objects = os.path.join(self.base_path, cam, "faces" if face else "objects")
                if not os.path.isdir(objects):
                    continue
                for day in os.listdir(objects):
                    for name in ("embeddings.pkl", "embeddings.idx", "embeddings.f32"):
                        f = os.path.join(objects, day, name)
                        if os.path.exists(f):
                            st_ = os.stat(f)
                            files.append((f, st_.st_mtime_ns, st_.st_size))
        sig = (tuple(sorted(files)), len(target))
        if getattr(self, "_files_sig", {}).get(key) == sig:
            return                                           # nothing on disk changed since the last reload: keep dict and device matrix
        for folder in sorted({os.path.dirname(f) for f, _, _ in files}):
            f = os.path.join(folder, "embeddings.pkl")
            if os.path.exists(f):
                with open(f, "rb") as fh:
                    emb = pickle.load(fh).get("embeddings", {})
                valid.update(emb.keys())
                target.update(emb)
            if os.path.exists(os.path.join(folder, "embeddings.idx")):
                st = EmbeddingStore(folder, 512 if face else 768)
                rows = st.rows()
                for i, p in enumerate(st.paths()):
                    valid.add(p)
                    target[p] = np.array(rows[i:i + 1])
        for k in set(target) - valid:
            del target[k]
        self._version[key] += 1
        if not hasattr(self, "_files_sig"):
            self._files_sig = {}
        self._files_sig[key] = (sig[0], len(target))

    # -- append-only store (clearcam_amd/store.py) instead of one pickle per folder ---------------------------
    def _new_index(self, dim: int, rows: int) -> EmbeddingIndex:
        return EmbeddingIndex(dim, max(2 * rows, 1024), device=self.model.device if self.model else 0, storage=self.index_storage)

    def _replace(self, key: str, index: EmbeddingIndex, table: "_RowTable"):
        old = self._dev.pop(key, None)
        if old is not None:
            old[0].close()
        self._dev[key] = (index, table)

    def attach_store(self, capacity: Optional[int] = None) -> int:
        """Load every stored crop under base_path (store files and/or the reference's pickles) straight into the device
        matrix: no per-crop dict entries, one H2D copy.  Subsequent `search` calls scan it; `add_embedding` appends (the
        matrix grows as needed).  Returns the number of rows."""
        from .store import load_all
        paths, rows = load_all(self.base_path, 768)
        index = EmbeddingIndex(768, max(capacity or 0, 2 * len(paths), 1024), device=self.model.device if self.model else 0,
                               storage=self.index_storage)
        table = _RowTable()
        if len(paths):
            index.add(rows, table.extend(list(paths)))
        self._replace("store", index, table)
        self._attached = True
        return len(paths)

    def add_embedding(self, path: str, emb) -> None:
        """What clearcam.py:1282-1287 does per crop (load pickle, insert, rewrite pickle) as one appended row: on disk in
        the crop's day folder, in the dict the reference exposes, and in the attached device matrix."""
        from .store import EmbeddingStore
        e = np.asarray(as_numpy(emb), np.float32).reshape(1, -1)
        if e.shape[1] != 768:
            raise ValueError(f"expected a 768-d crop embedding, got {e.shape[1]}")
        folder = os.path.dirname(path)
        st = self._stores.get(folder)
        if st is None:
            st = self._stores[folder] = EmbeddingStore(folder, 768)
        st.append([path], e)
        self.image_embeddings[path] = e
        self._version["image"] += 1
        if getattr(self, "_attached", False):
            index, table = self._dev["store"]
            index.add(e, table.extend([path]))

    def _device_index(self, embeddings: Dict[str, np.ndarray], key: str = "image") -> Tuple[EmbeddingIndex, "_RowTable"]:
        """(Re)build the HBM-resident matrix of one of the reference's dicts when it changed (the reference reloads before
        every search; `_load_all_embeddings` / `add_embedding` bump the version when they change anything)."""
        sig = (id(embeddings), len(embeddings), self._version[key])
        if key not in self._dev or self._dev_sig.get(key) != sig:
            paths = [p for p, e in embeddings.items() if e is not None]
            dim = int(np.asarray(embeddings[paths[0]]).size) if paths else (512 if key == "face" else 768)
            index, table = self._new_index(dim, len(paths)), _RowTable()
            if paths:
                index.add(np.stack([np.asarray(embeddings[p], np.float32).reshape(-1) for p in paths]), table.extend(paths))
            self._replace(key, index, table)
            self._dev_sig[key] = sig
        return self._dev[key]

    # -- search (:356-390) ------------------------------------------------------------------------------
    @staticmethod
    def _rank(table: "_RowTable", rows, scores, grouped: bool):
        """The tail of the reference's search (:378-390) over candidate rows given in ascending row order (= the dict's
        iteration order): best crop per track id, id-less crops appended, stable sort by score."""
        if grouped:
            best, loose = {}, []
            for r, sc in zip(rows, scores):
                oid = table.oid[r]
                if oid is not None:
                    if oid not in best or sc > best[oid][1]:
                        best[oid] = (table.paths[r], sc)
                else:
                    loose.append((table.paths[r], sc))
            results = list(best.values()) + loose
        else:
            results = [(table.paths[r], sc) for r, sc in zip(rows, scores)]
        results.sort(key=lambda x: x[1], reverse=True)
        return results

    def search(self, query=None, top_k=10, cam_name=None, timestamp=None, text_embedding=None, is_face=False):
        embeddings = self.face_embeddings if is_face else self.image_embeddings
        attached = getattr(self, "_attached", False) and not is_face and len(self._dev["store"][1].paths) > 0
        if not embeddings and not attached:
            print("No embeddings available.")
            return []
        if text_embedding is None:
            text_embedding = self.model._encode_text(query).numpy()
        q = np.asarray(as_numpy(text_embedding), np.float32).reshape(-1)
        index, table = self._dev["store"] if attached else self._device_index(embeddings, "face" if is_face else "image")
        if q.size != index.dim:
            raise ValueError(f"query has {q.size} dimensions, the {'face' if is_face else 'crop'} index {index.dim}")
        allowed = table.allowed(cam_name, timestamp)
        live = np.flatnonzero(allowed)
        if live.size == 0 or top_k <= 0:
            return []
        exact_loop = any(table.bad[g] for g in live)
        grouped = any(table.truthy[g] for g in live)
        # GPU: filtered top-K' (one HBM pass + radix select); host: the reference's best-per-track-id tail on K' rows.
        # The answer is exact once the top_k-th result scores strictly above the weakest candidate (no unseen row can
        # enter or change it) or every allowed row has been seen; otherwise K' grows, and past 1024 the full score vector
        # is ranked (many crops of few tracks above everything else: rare).
        kk = min(1024, max(128, 16 * int(top_k)))
        while not exact_loop:
            idx, sc = index.search(q, kk, allowed)
            keep = idx[0] >= 0
            rows, scs = idx[0][keep], sc[0][keep]
            order = np.argsort(rows, kind="stable")
            results = self._rank(table, rows[order].tolist(), [float(v) for v in scs[order]], grouped)
            if rows.size < kk:                              # every allowed row was a candidate: this IS the reference's loop
                return results[:top_k]
            if len(results) >= top_k and results[top_k - 1][1] > float(scs.min()):
                # Bit-equal scores among the answers (duplicate crops): the reference orders them by the first appearance of their
                # track id over ALL rows (dict order + stable sort), which a candidate list cannot know - rank the whole vector then.
                head = [sc_ for _, sc_ in results[:top_k + 1]]
                if len(set(head)) == len(head):
                    return results[:top_k]
                break
            if kk == 1024:
                break
            kk = min(1024, kk * 4)
        scores = index.scores(q)[0]                         # the whole score vector, ranked like the reference does it
        rows = [r for r in range(len(table.paths)) if allowed[table.group[r]]]
        if exact_loop:                                      # names the reference itself would raise on: let it raise the same way
            for r in rows:
                fn = os.path.basename(table.paths[r])
                if "_" in fn:
                    event_img_info(fn.split(".jpg")[0])
        return self._rank(table, rows, [float(scores[r]) for r in rows], grouped)[:top_k]

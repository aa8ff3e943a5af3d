"""Append-only embedding store, the data format on either side of the search path.

The reference keeps `{"embeddings": {crop_path: (1,768) float32}}` in one `embeddings.pkl` per camera/day folder, loads
and REWRITES the whole pickle for every new crop (clearcam.py:1282-1287: O(N) bytes per crop, O(N^2) per day) and walks and
unpickles every folder again before every search (models/objects.py:392-422, called at clearcam.py:1104-1105).

Here a folder holds two append-only files:
  embeddings.f32   N x dim float32 rows, no header      -> np.memmap, handed to cc_index_add as is (one H2D copy)
  embeddings.idx   N lines, the crop path of each row (UTF-8, '\\n' separated)
A crop costs one row + one line (the writer caches the consistent prefix and validates it with two stat calls).  Rows are written before their index line, and a reader takes
N = min(complete rows, complete lines), so a crash between the two writes loses at most the last crop and never
mis-pairs a path with a vector.  `import_pickle` / `as_reference_dict` convert from / to the reference's format, so
existing data keeps working and the reference can still read what this store holds.
"""
from __future__ import annotations

import os
import pickle
from typing import Dict, Iterable, List, Sequence, Tuple

import numpy as np

DATA, INDEX, PICKLE = "embeddings.f32", "embeddings.idx", "embeddings.pkl"


class EmbeddingStore:
    def __init__(self, folder: str, dim: int = 768):
        self.folder, self.dim = folder, dim
        self.data_path, self.index_path = os.path.join(folder, DATA), os.path.join(folder, INDEX)
        self._n = self._idx_bytes = None                        # rows / index bytes of the consistent prefix, once known

    # -- write ------------------------------------------------------------------------------------------
    def _sizes(self) -> Tuple[int, int]:
        return (os.path.getsize(self.data_path) if os.path.exists(self.data_path) else 0,
                os.path.getsize(self.index_path) if os.path.exists(self.index_path) else 0)

    def _state(self) -> Tuple[int, int]:
        """(rows, index bytes) of the consistent prefix.  The index file is read in full only when the cached state no
        longer matches the file sizes (first use, another writer, a torn write): an append costs two stat calls, one row
        and one line — not a re-read of the whole folder."""
        if self._n is not None and self._sizes() == (self._n * self.dim * 4, self._idx_bytes):
            return self._n, self._idx_bytes
        lines = self._lines()
        rows = self._sizes()[0] // (self.dim * 4)
        n = min(rows, len(lines))
        self._n, self._idx_bytes = n, sum(len(p.encode("utf-8")) + 1 for p in lines[:n])
        return self._n, self._idx_bytes

    def append(self, paths: Sequence[str], embs, durable: bool = True) -> int:
        """Append len(paths) rows; embs (n,dim) or (n,1,dim) float32.  Returns the new row count.  durable=False skips the
        two fsyncs (the pairing guarantee needs only the order rows-then-lines; fsync adds power-loss durability)."""
        e = np.ascontiguousarray(np.asarray(embs, np.float32).reshape(len(paths), -1))
        if e.shape[1] != self.dim:
            raise ValueError(f"expected {self.dim}-d embeddings, got {e.shape[1]}")
        for p in paths:
            if "\n" in p:
                raise ValueError("newline in path")
        os.makedirs(self.folder, exist_ok=True)
        n, ib = self._state()                                   # also the recovery point after a torn write
        text = "".join(p + "\n" for p in paths).encode("utf-8")
        with open(self.data_path, "r+b" if os.path.exists(self.data_path) else "wb") as f:
            f.seek(n * self.dim * 4)
            f.truncate()
            f.write(e.tobytes())
            f.flush()
            if durable:
                os.fsync(f.fileno())
        with open(self.index_path, "r+b" if os.path.exists(self.index_path) else "wb") as f:
            f.seek(ib)
            f.truncate()
            f.write(text)
            f.flush()
            if durable:
                os.fsync(f.fileno())
        self._n, self._idx_bytes = n + len(paths), ib + len(text)
        return self._n

    # -- read -------------------------------------------------------------------------------------------
    def _lines(self) -> List[str]:
        if not os.path.exists(self.index_path):
            return []
        raw = open(self.index_path, "rb").read()
        end = raw.rfind(b"\n") + 1                              # drop an incomplete last line
        return raw[:end].decode("utf-8").split("\n")[:-1] if end else []

    def __len__(self) -> int:
        return self._state()[0]

    def paths(self) -> List[str]:
        return self._lines()[:len(self)]

    def rows(self) -> np.ndarray:
        """(N,dim) float32 view of the file (np.memmap; zero rows -> empty array)."""
        n = len(self)
        if n == 0:
            return np.zeros((0, self.dim), np.float32)
        return np.memmap(self.data_path, dtype=np.float32, mode="r", shape=(n, self.dim))

    # -- reference format ---------------------------------------------------------------------------------
    def import_pickle(self, pkl_path: str = None) -> int:
        """Append every entry of a reference `embeddings.pkl` that is not stored yet; returns how many were added."""
        pkl_path = pkl_path or os.path.join(self.folder, PICKLE)
        with open(pkl_path, "rb") as f:
            emb = pickle.load(f).get("embeddings", {})
        have = set(self.paths())
        new = [(p, e) for p, e in emb.items() if p not in have and e is not None]
        if new:
            self.append([p for p, _ in new], np.stack([np.asarray(e, np.float32).reshape(-1) for _, e in new]))
        return len(new)

    def as_reference_dict(self) -> Dict[str, np.ndarray]:
        """{"embeddings": {path: (1,dim) float32}} exactly as clearcam.py:1286 stores it (later rows win on duplicate paths)."""
        rows = self.rows()
        return {"embeddings": {p: np.array(rows[i:i + 1]) for i, p in enumerate(self.paths())}}

    def export_pickle(self, pkl_path: str = None) -> None:
        with open(pkl_path or os.path.join(self.folder, PICKLE), "wb") as f:
            pickle.dump(self.as_reference_dict(), f)


def day_folders(base_path: str, face: bool = False) -> Iterable[str]:
    """data/cameras/<camera>/<objects|faces>/<day> folders, the walk of models/objects.py:398-407."""
    if not os.path.isdir(base_path):
        return
    for cam in sorted(os.listdir(base_path)):
        objects = os.path.join(base_path, cam, "faces" if face else "objects")
        if not os.path.isdir(objects):
            continue
        for day in sorted(os.listdir(objects)):
            d = os.path.join(objects, day)
            if os.path.isdir(d):
                yield d


def load_all(base_path: str, dim: int = 768, face: bool = False) -> Tuple[List[str], np.ndarray]:
    """Every stored crop under base_path: (paths, (N,dim) float32).  Folders that only hold the reference's pickle are read
    through it; a path present in both is taken once (the store row).  Later duplicates of a path win, like dict.update."""
    paths: List[str] = []
    blocks: List[np.ndarray] = []
    for d in day_folders(base_path, face):
        st = EmbeddingStore(d, dim)
        p = st.paths()
        if p:
            paths.extend(p)
            blocks.append(np.asarray(st.rows()))
        pkl = os.path.join(d, PICKLE)
        if os.path.exists(pkl):
            with open(pkl, "rb") as f:
                emb = pickle.load(f).get("embeddings", {})
            have = set(p)
            extra = [(k, v) for k, v in emb.items() if k not in have and v is not None]
            if extra:
                paths.extend(k for k, _ in extra)
                blocks.append(np.stack([np.asarray(v, np.float32).reshape(-1) for _, v in extra]))
    if not paths:
        return [], np.zeros((0, dim), np.float32)
    rows = np.concatenate(blocks)
    last = {}
    for i, p in enumerate(paths):
        last[p] = i
    if len(last) != len(paths):
        keep = sorted(last.values())
        paths, rows = [paths[i] for i in keep], rows[keep]
    return paths, rows

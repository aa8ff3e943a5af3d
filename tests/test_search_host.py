"""Host logic of ``ObjectFinder.search`` / ``_load_all_embeddings`` / ``add_embedding`` (clearcam_amd/objects.py) on the CPU.

The device matrix is replaced by a numpy double with the C ABI's contract (scores = f32 dot products, top-k descending with
ties to the lower row id, per-group filter, -1/-inf padding), so that what is tested here is the part the GPU does not do:
grouping rows by directory, the O(candidates) ranking tail and its exactness argument, the separation of the attached store,
the crop dict and the face dict, and the store files.  The expected answers come from `reference_search`, a line-by-line
restatement of models/objects.py:356-390 (the GPU tests hold the real index to the reference-run fixture)."""
import os
import pickle

import numpy as np
import pytest

from clearcam_amd import objects as O


class FakeIndex:
    def __init__(self, dim=768, capacity=1024, device=0, storage="f32"):
        self.dim, self.rows, self.groups, self.closed = dim, np.zeros((0, dim), np.float32), np.zeros(0, np.int32), False
        self.searches = []

    def __len__(self):
        return len(self.rows)

    def add(self, emb, groups=None):
        a = np.asarray(emb, np.float32)
        if a.shape[-1] != self.dim:
            raise ValueError("width")
        a = a.reshape(-1, self.dim)
        self.rows = np.concatenate([self.rows, a])
        self.groups = np.concatenate([self.groups, np.zeros(len(a), np.int32) if groups is None else np.asarray(groups, np.int32)])

    def scores(self, q):
        return (self.rows @ np.asarray(q, np.float32).reshape(-1, self.dim).T).T.astype(np.float32)

    def search(self, q, k, allowed=None):
        s = self.scores(q)
        self.searches.append(k)
        idx = np.full((s.shape[0], k), -1, np.int32); sc = np.full((s.shape[0], k), -np.inf, np.float32)
        for i in range(s.shape[0]):
            ok = np.ones(len(self.rows), bool) if allowed is None else np.array([g < len(allowed) and allowed[g] != 0 for g in self.groups], bool)
            cand = np.flatnonzero(ok)
            order = cand[np.argsort(-s[i][cand], kind="stable")][:k]
            idx[i, :len(order)] = order; sc[i, :len(order)] = s[i][order]
        return idx, sc

    def close(self):
        self.closed = True


@pytest.fixture(autouse=True)
def fake_index(monkeypatch):
    monkeypatch.setattr(O, "EmbeddingIndex", FakeIndex)


def reference_search(embeddings, text_embedding, top_k=10, cam_name=None, timestamp=None):
    """models/objects.py:364-390, statement by statement."""
    sims = []
    for path, emb in embeddings.items():
        if emb is None:
            continue
        norm = path.replace("\\", "/")
        if cam_name and f"/cameras/{cam_name}/" not in norm:
            continue
        if timestamp and f"/objects/{timestamp}/" not in norm and "/objects/video/" not in norm:
            continue
        sim = float((np.asarray(emb, np.float32).reshape(1, -1) @ text_embedding.reshape(-1, 1)).item())
        fn = os.path.basename(path)
        if fn.lower().endswith(".jpg"):
            oid = O.event_img_info(fn.split(".jpg")[0])["object_id"] if "_" in fn else None
            sims.append((path, sim, oid))
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


def make_world(rng, n, dim=32, cams=("front", "back", "yard"), days=("2026-01-01", "2026-01-02", "video"), n_ids=40, dup=0.05, neg_ids=False):
    emb = {}
    base = "data/cameras"
    vecs = rng.standard_normal((n, dim)).astype(np.float32)
    vecs /= np.linalg.norm(vecs, axis=1, keepdims=True)
    for i in range(n):
        cam, day = cams[rng.integers(len(cams))], days[rng.integers(len(days))]
        kind = rng.random()
        if kind < 0.80:
            name = f"{1700000000 + i}.5_{int(rng.integers(-n_ids if neg_ids else 0, n_ids))}_{int(rng.integers(0, 80))}.jpg"     # track ids include 0 (falsy)
        elif kind < 0.88:
            name = f"snapshot{i}.jpg"                                                                   # no '_' -> no track id
        elif kind < 0.94:
            name = f"{1700000000 + i}.5_{int(rng.integers(0, n_ids))}_3.JPG"                           # case-insensitive suffix... but split('.jpg') keeps the stem whole
        else:
            name = f"{1700000000 + i}_1_2.png"                                                          # never returned
        path = f"{base}/{cam}/objects/{day}/{name}"             # ('\\' separators only work in the reference on Windows: os.path.basename)
        v = vecs[i] if rng.random() > dup or i == 0 else vecs[rng.integers(0, i)]                       # exact score ties
        emb[path] = v.reshape(1, dim).copy()
    return emb


def finder_with(emb, dim):
    f = O.ObjectFinder("nowhere")
    f.image_embeddings = emb
    return f


@pytest.mark.parametrize("seed", range(6))
def test_search_equals_reference_loop(seed):
    rng = np.random.default_rng(seed)
    dim = 32
    emb = make_world(rng, 1500, dim)
    # '.JPG' names: the reference's fn.split(".jpg")[0] leaves the whole name, event_img_info then parses "3.JPG" -> ValueError.
    # Keep only the worlds the reference itself can search; the raising case has its own test below.
    emb = {p: v for p, v in emb.items() if not p.endswith(".JPG")}
    f = finder_with(emb, dim)
    for _ in range(12):
        q = rng.standard_normal(dim).astype(np.float32)
        cam = [None, "front", "back", "nope"][rng.integers(4)]
        ts = [None, "2026-01-01", "2026-01-02", "1999-01-01"][rng.integers(4)]
        k = int(rng.choice([1, 3, 10, 50, 400]))
        want = reference_search(emb, q, k, cam, ts)
        got = f.search(top_k=k, cam_name=cam, timestamp=ts, text_embedding=q)
        assert [p for p, _ in got] == [p for p, _ in want]
        assert np.allclose([s for _, s in got], [s for _, s in want], rtol=0, atol=1e-6)


@pytest.mark.parametrize("seed", range(8))
def test_duplicate_crops_and_negative_ids_rank_like_the_reference(seed):
    """(1) bit-equal scores (half the crops are duplicates of earlier ones): the reference orders tied answers by the
    first appearance of their track id over ALL rows, so a candidate list that shows ties among the answers must hand over to the
    full-vector path; (2) negative track ids are ids like any other (`object_id is not None`), not id-less crops."""
    rng = np.random.default_rng(100 + seed)
    dim = 16
    emb = make_world(rng, 1200, dim, n_ids=25, dup=0.5, neg_ids=True)
    emb = {p: v for p, v in emb.items() if not p.endswith(".JPG")}
    f = finder_with(emb, dim)
    for _ in range(20):
        q = rng.standard_normal(dim).astype(np.float32)
        cam = [None, "front", "back"][rng.integers(3)]
        ts = [None, "2026-01-01", "2026-01-02"][rng.integers(3)]
        k = int(rng.choice([1, 2, 5, 10, 30, 200]))
        want = reference_search(emb, q, k, cam, ts)
        got = f.search(top_k=k, cam_name=cam, timestamp=ts, text_embedding=q)
        assert [p for p, _ in got] == [p for p, _ in want], (seed, k, cam, ts)


def test_candidate_growth_and_full_fallback():
    """Thousands of crops of ONE track outrank everything else: the first candidate lists hold a single track, so the search
    must widen K' and finally rank the whole score vector - and still return exactly the reference's answer."""
    rng = np.random.default_rng(11)
    dim = 16
    q = np.zeros(dim, np.float32); q[0] = 1.0
    emb = {}
    for i in range(2500):                                            # one dominant track, scores 0.9 .. 0.99
        v = np.zeros(dim, np.float32); v[0] = 0.9 + 0.09 * rng.random(); v[1] = rng.random() * 0.1
        emb[f"data/cameras/front/objects/2026-01-01/{i}.0_7_2.jpg"] = v.reshape(1, -1)
    for i in range(300):                                             # everybody else: weaker, distinct tracks
        v = np.zeros(dim, np.float32); v[0] = 0.5 * rng.random(); v[2] = 0.3
        emb[f"data/cameras/front/objects/2026-01-01/{9000 + i}.0_{100 + i}_2.jpg"] = v.reshape(1, -1)
    f = finder_with(emb, dim)
    got = f.search(top_k=10, text_embedding=q)
    assert got == reference_search(emb, q, 10) or [p for p, _ in got] == [p for p, _ in reference_search(emb, q, 10)]
    ks = f._dev["image"][0].searches
    assert ks == [160, 640, 1024]                                    # widened twice, then the full vector
    # a plain query is answered from the first candidate list
    q2 = rng.standard_normal(dim).astype(np.float32)
    f._dev["image"][0].searches.clear()
    assert [p for p, _ in f.search(top_k=5, text_embedding=q2)] == [p for p, _ in reference_search(emb, q2, 5)]
    assert len(f._dev["image"][0].searches) <= 2


def test_unparsable_names_raise_like_the_reference():
    dim = 8
    emb = {"data/cameras/a/objects/d/12.0_x_1.jpg": np.ones((1, dim), np.float32), "data/cameras/b/objects/d/1.0_2_3.jpg": np.ones((1, dim), np.float32)}
    f = finder_with(emb, dim)
    q = np.ones(dim, np.float32)
    with pytest.raises(ValueError):
        reference_search(emb, q)
    with pytest.raises(ValueError):
        f.search(text_embedding=q)
    assert [p for p, _ in f.search(text_embedding=q, cam_name="b")] == [p for p, _ in reference_search(emb, q, cam_name="b")]   # the bad row is filtered out


def test_sources_do_not_share_state(tmp_path):
    """a face search after attach_store() must not replace the attached crop matrix, and a 768-d row must never
    be appended to a 512-d matrix."""
    rng = np.random.default_rng(3)
    base = tmp_path / "cameras"
    day = base / "front" / "objects" / "2026-01-01"
    os.makedirs(day)
    crops = {f"{day}/{i}.0_{i}_2.jpg": (rng.standard_normal((1, 768)).astype(np.float32)) for i in range(20)}
    with open(day / "embeddings.pkl", "wb") as fh:
        pickle.dump({"embeddings": crops}, fh)
    f = O.ObjectFinder(str(base))
    assert f.attach_store() == 20
    store_index = f._dev["store"][0]
    f.face_embeddings = {f"{base}/front/faces/x/{i}.0_{i}_0.jpg": rng.standard_normal((1, 512)).astype(np.float32) for i in range(5)}
    qf, qi = rng.standard_normal(512).astype(np.float32), rng.standard_normal(768).astype(np.float32)
    faces = f.search(top_k=3, text_embedding=qf, is_face=True)
    assert len(faces) == 3 and all("/faces/" in p for p, _ in faces)
    assert f._dev["store"][0] is store_index and not store_index.closed and store_index.dim == 768
    hits = f.search(top_k=5, text_embedding=qi)
    assert [p for p, _ in hits] == [p for p, _ in reference_search(crops, qi, 5)]
    new = str(day / "99.0_99_2.jpg")
    f.add_embedding(new, qi * 10)                                  # outscores every stored crop
    assert len(store_index) == 21 and f.search(top_k=1, text_embedding=qi)[0][0] == new
    with pytest.raises(ValueError):
        f.add_embedding(str(day / "100.0_100_2.jpg"), np.zeros(512, np.float32))
    with pytest.raises(ValueError):
        f.search(top_k=1, text_embedding=qf)                         # a 512-d query against the crop matrix
    # the reference-style reload (clearcam.py:1104-1105) sees the row add_embedding stored in the append-only files
    f._load_all_embeddings()
    assert new in f.image_embeddings and len(f.image_embeddings) == 21


def test_dict_index_follows_in_place_changes(tmp_path):
    """one stale key removed + one new key added keeps id() and len() of the dict; the device matrix must still
    be rebuilt (version counter bumped by _load_all_embeddings / add_embedding)."""
    rng = np.random.default_rng(4)
    base = tmp_path / "cameras"
    d1 = base / "front" / "objects" / "2026-01-01"
    os.makedirs(d1)
    e = {f"{d1}/{i}.0_{i}_2.jpg": rng.standard_normal((1, 768)).astype(np.float32) for i in range(4)}
    with open(d1 / "embeddings.pkl", "wb") as fh:
        pickle.dump({"embeddings": dict(list(e.items())[:3])}, fh)
    f = O.ObjectFinder(str(base))
    f._load_all_embeddings()
    q = np.asarray(list(e.values())[3]).reshape(-1)
    first = f.search(top_k=3, text_embedding=q)
    idx_before = f._dev["image"][0]
    f._load_all_embeddings()                                         # nothing changed on disk: no reload, same matrix
    f.search(top_k=3, text_embedding=q)
    assert f._dev["image"][0] is idx_before
    items = list(e.items())
    with open(d1 / "embeddings.pkl", "wb") as fh:                    # drop crop 0, add crop 3: same number of keys
        pickle.dump({"embeddings": dict(items[1:])}, fh)
    os.utime(d1 / "embeddings.pkl", ns=(1, 2_000_000_000_000_000_000))
    f._load_all_embeddings()
    again = f.search(top_k=3, text_embedding=q)
    assert again[0][0] == items[3][0] and items[0][0] not in [p for p, _ in again] and again != first


def test_jit_infer_passes_the_callable_every_time():
    """CLIP switched off and on again (clearcam.py:1250-1253) with the same global jit_cache must call the NEW
    model; two callables over one input shape must not be confused."""
    from clearcam_amd.helpers import Tensor, jit_infer
    cache = {}
    x = Tensor(np.zeros((1, 3, 224, 224), np.float32))

    class Model:
        def __init__(self, tag):
            self.tag, self.open = tag, True

        def embed(self, t):
            if not self.open:
                raise RuntimeError("closed handle")
            return self.tag
    old = Model("old")
    assert jit_infer(old.embed, x, cache) == "old"
    old.open = False                                                 # turn_off_clip() closed it
    new = Model("new")
    assert jit_infer(new.embed, x, cache) == "new"
    assert jit_infer(lambda t: "other", x, cache) == "other" and list(cache) == [(1, 3, 224, 224)]

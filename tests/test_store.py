"""Append-only embedding store (clearcam_amd/store.py) and its compatibility with the reference's embeddings.pkl
(clearcam.py:1282-1287 writer, models/objects.py:392-422 reader).  Host-only."""
import os
import pickle

import numpy as np
import pytest

from clearcam_amd.store import DATA, INDEX, EmbeddingStore, load_all


def _emb(rng, n, dim=768):
    e = rng.standard_normal((n, dim)).astype(np.float32)
    return e / np.linalg.norm(e, axis=1, keepdims=True)


def test_append_read_and_reference_dict(tmp_path):
    rng = np.random.default_rng(0)
    d = str(tmp_path / "cameras" / "front" / "objects" / "2026-01-01")
    st = EmbeddingStore(d)
    assert len(st) == 0 and st.paths() == [] and st.rows().shape == (0, 768)
    e = _emb(rng, 5)
    p = [f"{d}/{1000 + i}.0_{i}_2.jpg" for i in range(5)]
    assert st.append(p[:2], e[:2]) == 2
    assert st.append(p[2:], e[2:, None, :]) == 5                 # (n,1,dim) as precompute_embedding returns per crop
    assert st.paths() == p and np.array_equal(np.asarray(st.rows()), e)
    ref = st.as_reference_dict()                                # exactly what clearcam.py:1286 pickles
    assert list(ref) == ["embeddings"] and list(ref["embeddings"]) == p
    assert all(v.shape == (1, 768) and v.dtype == np.float32 for v in ref["embeddings"].values())
    st.export_pickle()
    with open(os.path.join(d, "embeddings.pkl"), "rb") as f:
        back = pickle.load(f)
    assert np.array_equal(back["embeddings"][p[3]], e[3:4])
    with pytest.raises(ValueError):
        st.append(["x"], np.zeros((1, 5), np.float32))
    with pytest.raises(ValueError):
        st.append(["a\nb"], e[:1])


def test_torn_writes_never_mispair(tmp_path):
    rng = np.random.default_rng(1)
    st = EmbeddingStore(str(tmp_path), dim=8)
    e = _emb(rng, 4, 8)
    st.append(["a", "b", "c"], e[:3])
    with open(os.path.join(str(tmp_path), DATA), "ab") as f:      # crash after a row (and a half) but before the index line
        f.write(e[3].tobytes() + b"\x00" * 7)
    assert len(st) == 3 and st.paths() == ["a", "b", "c"]
    st.append(["d"], e[3:4])                                     # the next append overwrites the orphan bytes
    assert st.paths() == ["a", "b", "c", "d"] and np.array_equal(np.asarray(st.rows()), e)
    with open(os.path.join(str(tmp_path), INDEX), "ab") as f:     # crash in the middle of an index line
        f.write(b"half-written-pa")
    assert len(st) == 4
    st.append(["e"], e[:1])
    assert st.paths()[-1] == "e" and len(st) == 5 and np.array_equal(np.asarray(st.rows())[4], e[0])


def test_load_all_merges_stores_and_reference_pickles(tmp_path):
    rng = np.random.default_rng(2)
    base = tmp_path / "cameras"
    d1, d2 = base / "front" / "objects" / "2026-01-01", base / "back" / "objects" / "2026-01-02"
    os.makedirs(d1); os.makedirs(d2)
    e = _emb(rng, 6)
    # folder 1: only the reference's pickle (written exactly like clearcam.py:1282-1287)
    data = {"embeddings": {f"{d1}/1.0_1_0.jpg": e[0:1], f"{d1}/2.0_2_0.jpg": e[1:2]}}
    with open(d1 / "embeddings.pkl", "wb") as f:
        pickle.dump(data, f)
    # folder 2: a store, plus a stale pickle holding one path of the store and one extra
    st = EmbeddingStore(str(d2))
    st.append([f"{d2}/3.0_3_0.jpg", f"{d2}/4.0_4_0.jpg"], e[2:4])
    with open(d2 / "embeddings.pkl", "wb") as f:
        pickle.dump({"embeddings": {f"{d2}/3.0_3_0.jpg": e[5:6], f"{d2}/5.0_5_0.jpg": e[4:5]}}, f)
    paths, rows = load_all(str(base))
    got = dict(zip(paths, rows))
    assert len(paths) == 5 and rows.shape == (5, 768)
    assert np.array_equal(got[f"{d1}/2.0_2_0.jpg"], e[1]) and np.array_equal(got[f"{d2}/5.0_5_0.jpg"], e[4])
    assert np.array_equal(got[f"{d2}/3.0_3_0.jpg"], e[2])        # the store row wins over the stale pickle entry
    assert EmbeddingStore(str(d1)).import_pickle() == 2 and EmbeddingStore(str(d1)).import_pickle() == 0
    assert load_all(str(tmp_path / "nowhere")) == ([], pytest.approx(np.zeros((0, 768))))

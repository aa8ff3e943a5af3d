"""GPU parity tests of the CLIP towers and the embedding index through the C ABI.

Tolerance (BASELINE.json north_star): cosine(HIP, oracle) >= 1 - 1e-4 for image and text embeddings, in every
storage dtype; f32 mode additionally within 1e-5 absolute per component.
"""
import os

import numpy as np
import pytest

from clearcam_amd.arch import CLIP_B32, CLIP_L14, CLIP_TINY
from clearcam_amd.weights import synthetic_clip_state_dict
from oracle.clip_oracle import OpenCLIPOracle, pad_tokens, search_reference

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tiny():
    sd = synthetic_clip_state_dict(CLIP_TINY, 4321)
    return sd, OpenCLIPOracle(sd, CLIP_TINY)


def _toks(arch):
    sot, eot = arch.t_vocab - 2, arch.t_vocab - 1
    rows = [pad_tokens([5, 9, 44], arch.t_ctx, sot, eot), pad_tokens(list(range(1, 40)), arch.t_ctx, sot, eot),
            pad_tokens([], arch.t_ctx, sot, eot), pad_tokens(list(range(1, 76)), arch.t_ctx, sot, eot)]   # empty and full-length prompts
    return np.concatenate(rows)


@pytest.mark.parametrize("dtype", ["f32", "f16", "bf16"])
def test_clip_tiny_matches_oracle(dtype, tiny):
    from clearcam_amd.objects import OpenCLIP
    sd, o = tiny
    m = OpenCLIP(state_dict=sd, arch=CLIP_TINY, dtype=dtype)
    x = np.random.default_rng(2).random((5, 3, 56, 56), dtype=np.float32) * 2 - 1
    ref, got = o.precompute_embedding(x), m.precompute_embedding(x).numpy()
    assert got.shape == (5, 64) and np.allclose(np.linalg.norm(got, axis=1), 1, atol=1e-5)
    assert ((ref * got).sum(1) >= 1 - 1e-4).all()
    toks = _toks(CLIP_TINY)
    rt, gt = o.encode_tokens(toks), m.encode_tokens(toks)
    assert ((rt * gt).sum(1) >= 1 - 1e-4).all()
    if dtype == "f32":
        assert np.abs(ref - got).max() < 1e-5 and np.abs(rt - gt).max() < 1e-5
    one = m.precompute_embedding(x[3:4]).numpy()[0]                       # batch invariance
    assert np.abs(one - got[3]).max() < (1e-6 if dtype == "f32" else 5e-3)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_clip_l14_matches_oracle(dtype):
    """The real ViT-L/14 shapes (24+12 layers, 257/77 tokens, 768-d output), seeded weights."""
    from clearcam_amd.objects import OpenCLIP
    sd = synthetic_clip_state_dict(CLIP_L14, 4321)
    o = OpenCLIPOracle(sd, CLIP_L14)
    m = OpenCLIP(state_dict=sd, arch=CLIP_L14, dtype=dtype)
    x = np.random.default_rng(2).random((2, 3, 224, 224), dtype=np.float32) * 2 - 1      # test_clip_speed.py:11 style input
    ref, got = o.precompute_embedding(x), m.precompute_embedding(x).numpy()
    assert got.shape == (2, 768)
    assert ((ref * got).sum(1) >= 1 - 1e-4).all(), (ref * got).sum(1)
    toks = np.concatenate([pad_tokens([9606, 325, 275, 271]), pad_tokens([4160, 763])])   # "ferrari f40", "text here"
    rt, gt = o.encode_tokens(toks), m.encode_tokens(toks)
    assert ((rt * gt).sum(1) >= 1 - 1e-4).all(), (rt * gt).sum(1)
    # the quantity the reference's own test pins (test_clip.py:9-12): text-image cosine, HIP vs oracle
    assert abs(float(gt[0] @ got[0]) - float(rt[0] @ ref[0])) < (1e-5 if dtype == "f32" else 2e-3)


def test_text_surface_and_tokenizer_roundtrip(tiny):
    """`OpenCLIP._encode_text(str)` (models/objects.py:135-143) end to end.  With clearcam's vocabulary file when it is around,
    otherwise with the committed subset of its merge table that covers the KAT strings (tests/conftest.py sparse_tokenizer)."""
    from clearcam_amd.clip_tokenizer import find_vocab
    from conftest import sparse_tokenizer
    try:
        find_vocab()
        tok = None                                                    # the model builds its own from the full table
    except FileNotFoundError:
        tok = sparse_tokenizer()
    from clearcam_amd.objects import OpenCLIP
    sd = synthetic_clip_state_dict(CLIP_L14, 4321)
    sd = {k: v for k, v in sd.items() if not k.startswith(("resblocks_img", "visual", "ln_p", "class_", "positional_embedding", "proj")) or k == "positional_embedding_text"}
    m = OpenCLIP(state_dict=sd, arch=CLIP_L14, dtype="f32", tokenizer=tok)           # text tower only
    assert m.tokenizer.tokens_for_model("ferrari f40")[0, :6].tolist() == [49406, 9606, 325, 275, 271, 49407]   # the reference's ids
    e = m._encode_text("ferrari f40")
    assert e.numpy().shape == (768,)
    assert np.array_equal(m._encode_text("ferrari f40", realize=True), e.numpy())
    assert np.array_equal(m._encode_text("  Ferrari   F40 ").numpy(), e.numpy())     # clean(): lower + whitespace collapse
    with pytest.raises(Exception):
        m.precompute_embedding(np.zeros((1, 3, 224, 224), np.float32))               # image tower not loaded -> loud error


def test_index_scores_and_topk_exact():
    from clearcam_amd.objects import EmbeddingIndex
    rng = np.random.default_rng(3)
    E = rng.standard_normal((70000, 768)).astype(np.float32); E /= np.linalg.norm(E, axis=1, keepdims=True)
    E[60000] = E[123]                                                     # exact duplicate -> tie broken by row id
    q = rng.standard_normal((6, 768)).astype(np.float32); q /= np.linalg.norm(q, axis=1, keepdims=True)
    q[5] = E[123]
    ix = EmbeddingIndex(768, 80000)
    assert len(ix) == 0
    i0, s0 = ix.search(q, 7)
    assert (i0 == -1).all() and np.isneginf(s0).all()                     # empty index
    ix.add(E[:16384]); ix.add(E[16384:])                                  # ragged appends across a chunk boundary
    assert len(ix) == 70000
    sc = ix.scores(q)
    assert np.abs(sc - q @ E.T).max() < 1e-6
    for k in (1, 100, 1024):
        idx, s = ix.search(q, k)
        order = np.argsort(-sc, axis=1, kind="stable")[:, :k]
        assert np.array_equal(idx, order) and np.array_equal(s, np.take_along_axis(sc, order, 1))
    assert ix.search(q, 2)[0][5].tolist() == [123, 60000]
    small = EmbeddingIndex(768, 10); small.add(E[:3])
    i3, s3 = small.search(q[:1], 5)
    assert (i3[0, 3:] == -1).all() and sorted(i3[0, :3].tolist()) == [0, 1, 2]
    small.add(E[:100])                                                    # past the initial capacity: the matrix grows, nothing is lost
    assert len(small) == 103 and small.capacity >= 103
    i4, _ = small.search(E[50:51], 1)
    assert i4[0, 0] in (50, 53)                                           # E[50] is stored as rows 50 and 53 (ties -> the lower row id)


def test_object_finder_search_matches_reference_loop(tiny):
    """ObjectFinder.search (device scan) == the reference's Python loop (objects.py:365-390) on the same store."""
    from clearcam_amd.objects import ObjectFinder
    rng = np.random.default_rng(5)
    store = {}
    for i in range(500):
        v = rng.standard_normal(768).astype(np.float32); v /= np.linalg.norm(v)
        cam = "front" if i % 2 else "back"
        day = "2026-01-01" if i % 3 else "2026-01-02"
        store[f"data/cameras/{cam}/objects/{day}/{1000 + i}.0_{i % 40}_{i % 5}.jpg"] = v[None]
    store["data/cameras/front/objects/2026-01-01/readme.txt"] = store[next(iter(store))]
    q = rng.standard_normal(768).astype(np.float32); q /= np.linalg.norm(q)
    f = ObjectFinder(); f.image_embeddings = store
    for kw in ({}, {"cam_name": "front"}, {"timestamp": "2026-01-02"}, {"top_k": 3}):
        got = f.search(text_embedding=q, **kw)
        ref = search_reference(store, q, **kw)
        assert [p for p, _ in got] == [p for p, _ in ref]
        assert np.allclose([s for _, s in got], [s for _, s in ref], atol=1e-6)
    assert ObjectFinder().search(text_embedding=q) == []


def _crops(rng):
    shapes = [(1, 1), (2, 3), (3, 5), (57, 131), (224, 224), (225, 223), (500, 300), (97, 1200), (1080, 64)]
    return [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for h, w in shapes]


def test_crop_preprocess_bit_exact_vs_opencv_restatement():
    """cc_crop_preprocess == oracle/cv_resize_oracle (OpenCV 4.10 8-bit INTER_CUBIC restatement), bit for bit,
    on ragged crop sizes incl. 1x1, up- and down-scaling and non-square aspect (models/objects.py:237-242)."""
    from clearcam_amd.objects import ObjectFinder, preprocess_crops
    from oracle.cv_resize_oracle import preprocess
    crops = _crops(np.random.default_rng(7))
    got = preprocess_crops(crops).cpu().numpy()
    assert got.shape == (len(crops), 3, 224, 224) and got.dtype == np.float32
    for c, g in zip(crops, got):
        assert np.array_equal(preprocess(c), g)
    one = ObjectFinder().preprocess(crops[3])                            # the reference's single-image surface
    assert one.shape == (3, 224, 224) and np.array_equal(one, got[3])
    smooth = np.clip(np.add.outer(np.arange(300), np.arange(200))[:, :, None] * np.array([0.5, 0.3, 0.7]), 0, 255).astype(np.uint8)
    assert np.array_equal(preprocess_crops([smooth], 56).cpu().numpy()[0], preprocess(smooth, 56))
    with pytest.raises(ValueError):
        preprocess_crops([np.zeros((0, 4, 3), np.uint8)])


def test_crops_to_embeddings_chain(tiny):
    """crop -> cubic preprocess -> precompute_embedding entirely on the device == oracle chain (clearcam.py:1280-1285)."""
    import torch
    from clearcam_amd.objects import OpenCLIP, preprocess_crops
    from oracle.cv_resize_oracle import preprocess
    sd, o = tiny
    crops = _crops(np.random.default_rng(8))
    m = OpenCLIP(state_dict=sd, arch=CLIP_TINY, dtype="f32")
    x = preprocess_crops(crops, CLIP_TINY.image_size)
    emb = torch.empty(len(crops), CLIP_TINY.embed, device=x.device)
    m.precompute_embedding_device(x, emb)
    ref = o.precompute_embedding(np.stack([preprocess(c, CLIP_TINY.image_size) for c in crops]))
    assert np.abs(emb.cpu().numpy() - ref).max() < 1e-5


def test_store_backed_search_matches_reference_loop(tmp_path):
    """ObjectFinder over the append-only store (attach_store + add_embedding) == the reference's dict + Python loop."""
    import pickle
    from clearcam_amd.objects import ObjectFinder
    rng = np.random.default_rng(9)
    base = tmp_path / "cameras"
    ref_store = {}
    f = ObjectFinder(base_path=str(base))
    def emb():
        v = rng.standard_normal(768).astype(np.float32)
        return (v / np.linalg.norm(v))[None]
    # half of the history exists as reference pickles, the rest is appended crop by crop
    d0 = base / "front" / "objects" / "2026-01-01"
    d0.mkdir(parents=True)
    old = {f"{d0}/{100 + i}.0_{i % 7}_{i % 3}.jpg": emb() for i in range(60)}
    with open(d0 / "embeddings.pkl", "wb") as fh:
        pickle.dump({"embeddings": old}, fh)
    ref_store.update(old)
    assert f.attach_store() == 60
    for i in range(80):
        cam, day = ("back", "2026-01-02") if i % 2 else ("front", "2026-01-01")
        p = str(base / cam / "objects" / day / f"{500 + i}.0_{i % 11}_{i % 4}.jpg")
        e = emb()
        f.add_embedding(p, e)
        ref_store[p] = e
    q = emb()[0]
    for kw in ({}, {"cam_name": "back"}, {"timestamp": "2026-01-01"}, {"top_k": 5}):
        got, ref = f.search(text_embedding=q, **kw), search_reference(ref_store, q, **kw)
        assert [p for p, _ in got] == [p for p, _ in ref]
        assert np.allclose([s for _, s in got], [s for _, s in ref], atol=1e-6)
    g = ObjectFinder(base_path=str(base))                        # a fresh process finds everything on disk
    assert g.attach_store() == 140
    assert [p for p, _ in g.search(text_embedding=q)] == [p for p, _ in search_reference(ref_store, q)]


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_clip_b32_matches_oracle(dtype):
    """ViT-B/32 (the model BASELINE.json's north star names) through the same engine: 50 tokens, 12 heads, 512-d space."""
    from clearcam_amd.objects import OpenCLIP
    sd = synthetic_clip_state_dict(CLIP_B32, 99)
    o = OpenCLIPOracle(sd, CLIP_B32)
    m = OpenCLIP(state_dict=sd, arch=CLIP_B32, dtype=dtype)
    x = np.random.default_rng(3).random((3, 3, 224, 224), dtype=np.float32) * 2 - 1
    ref, got = o.precompute_embedding(x), m.precompute_embedding(x).numpy()
    assert got.shape == (3, 512) and ((ref * got).sum(1) >= 1 - 1e-4).all()
    toks = _toks(CLIP_B32)
    assert ((o.encode_tokens(toks) * m.encode_tokens(toks)).sum(1) >= 1 - 1e-4).all()
    if dtype == "f32":
        assert np.abs(ref - got).max() < 1e-5


@pytest.mark.parametrize("image_size,patch,t_ctx", [(112, 14, 40),      # 65 image tokens: five query tiles, the last one split over the keys; 40 causal text tokens
                                                   (224, 16, 65),      # 197 tokens (ViT-B/16's count): thirteen tiles + split tail on the 18-fragment instantiation; 65 causal
                                                   (168, 14, 96),      # 145 tokens: ten tiles, no split; 96 text tokens = six whole key fragments
                                                   (224, 14, 77)])     # 257 / 77: the straight-line instantiations, next to the run-time form on the same weights' shapes
@pytest.mark.parametrize("dtype", ["f16", "bf16"])
def test_attention_token_counts(dtype, image_size, patch, t_ctx):
    """attn_mfma_kernel beyond the three token counts the reference runs: other lengths take the run-time form of the score / softmax section
    (key-padding and causal masks decided per fragment at run time), with and without the split last query tile - against the oracle."""
    from dataclasses import replace
    from clearcam_amd.objects import OpenCLIP
    arch = replace(CLIP_TINY, image_size=image_size, patch=patch, t_ctx=t_ctx, v_width=128, v_heads=2, t_width=64, t_heads=1)
    sd = synthetic_clip_state_dict(arch, 17)
    o = OpenCLIPOracle(sd, arch)
    m = OpenCLIP(state_dict=sd, arch=arch, dtype=dtype)
    x = np.random.default_rng(5).random((3, 3, image_size, image_size), dtype=np.float32) * 2 - 1
    ref, got = o.precompute_embedding(x), m.precompute_embedding(x).numpy()
    assert ((ref * got).sum(1) >= 1 - 1e-4).all(), (ref * got).sum(1)
    sot, eot = arch.t_vocab - 2, arch.t_vocab - 1
    toks = np.concatenate([pad_tokens([5, 9, 44], t_ctx, sot, eot), pad_tokens(list(range(1, t_ctx - 1)), t_ctx, sot, eot), pad_tokens([], t_ctx, sot, eot)])
    assert ((o.encode_tokens(toks) * m.encode_tokens(toks)).sum(1) >= 1 - 1e-4).all()


def test_multi_query_scan_matches_single_query_path():
    """Q > 4 queries go through one GEMM pass over the index (exact-f32 MFMA) instead of ceil(Q/4) GEMV passes: same
    scores to f32 rounding, same top-k semantics (stable descending order of the returned scores)."""
    from clearcam_amd.objects import EmbeddingIndex
    rng = np.random.default_rng(11)
    E = rng.standard_normal((50000, 768)).astype(np.float32); E /= np.linalg.norm(E, axis=1, keepdims=True)
    Q = rng.standard_normal((64, 768)).astype(np.float32); Q /= np.linalg.norm(Q, axis=1, keepdims=True)
    ix = EmbeddingIndex(768, 50000); ix.add(E)
    many = ix.scores(Q)                                                    # GEMM path
    single = np.concatenate([ix.scores(Q[i:i + 1]) for i in range(0, 64, 16)])       # GEMV path on a few rows
    assert np.abs(many - Q @ E.T).max() < 1e-6
    assert np.abs(many[::16] - single).max() < 1e-6
    idx, s = ix.search(Q, 100)
    order = np.argsort(-many, axis=1, kind="stable")[:, :100]
    assert np.array_equal(idx, order) and np.array_equal(s, np.take_along_axis(many, order, 1))
    odd = EmbeddingIndex(768, 50001); odd.add(E); odd.add(E[:1])             # N % 4 != 0 -> GEMV passes, same answers
    assert np.abs(odd.scores(Q[:8])[:, :50000] - many[:8]).max() < 1e-6


def test_plan_cache_is_bounded_lru():
    """One plan (buffers + hipGraph) per input shape, at most CLEARCAM_MAX_PLANS alive: with a cap of 2, cycling through
    three batch sizes keeps evicting and rebuilding, results stay right and device memory does not grow."""
    import subprocess
    import sys
    code = r'''
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from clearcam_amd.arch import CLIP_TINY
from clearcam_amd.weights import synthetic_clip_state_dict, synthetic_yolov9_state_dict
from clearcam_amd.objects import OpenCLIP
from clearcam_amd.yolov9 import YOLOv9
from clearcam_amd._lib import CCError
m = OpenCLIP(state_dict=synthetic_clip_state_dict(CLIP_TINY, 4321), arch=CLIP_TINY, dtype="f32")
x = np.random.default_rng(0).random((3, 3, 56, 56), dtype=np.float32) * 2 - 1
ref = {b: m.precompute_embedding(x[:b]).numpy() for b in (1, 2, 3)}
torch.cuda.synchronize(); free0 = torch.cuda.mem_get_info()[0]
for _ in range(5):
    for b in (1, 2, 3):
        assert np.array_equal(m.precompute_embedding(x[:b]).numpy(), ref[b])
torch.cuda.synchronize(); assert abs(torch.cuda.mem_get_info()[0] - free0) < 64 << 20
y = YOLOv9("t", 320, state_dict=synthetic_yolov9_state_dict("t", 1234), dtype="f32")
f = np.random.default_rng(1).integers(0, 256, (1, 160, 320, 3), dtype=np.uint8)
a = y.detect_batch(f)
y.detect_batch(np.repeat(f, 2, 0)); y.detect_batch(np.repeat(f, 3, 0))        # two more shapes: the first plan is evicted
try:
    y.get_tensor("p3")                                                           # the tap's plan is the last one run: still fine
except CCError as e:
    raise SystemExit("unexpected: " + str(e))
assert np.array_equal(y.detect_batch(f), a)                                     # rebuilt, same answer
print("OK")
'''
    env = dict(os.environ, CLEARCAM_MAX_PLANS="2")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), timeout=600)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_index_grows_filters_and_keeps_results_on_device():
    """cc_index_add* past the initial capacity (geometric growth, rows and group ids preserved), the per-group filter of
    cc_index_search_groups, rows of the wrong width refused, and search_device (result stays on the GPU)."""
    import torch
    from clearcam_amd.objects import EmbeddingIndex
    rng = np.random.default_rng(21)
    E = rng.standard_normal((5000, 768)).astype(np.float32); E /= np.linalg.norm(E, axis=1, keepdims=True)
    grp = rng.integers(0, 7, 5000).astype(np.int32)
    ix = EmbeddingIndex(768, 16)                                          # far too small: grows 16 -> 8192 while appending
    for i in range(0, 5000, 617):
        ix.add(E[i:i + 617], grp[i:i + 617])
    assert len(ix) == 5000 and ix.capacity >= 5000
    q = E[1234:1235] + 0.01 * rng.standard_normal((1, 768)).astype(np.float32)
    full = (E @ q[0]).astype(np.float32)
    idx, sc = ix.search(q, 50)
    order = np.argsort(-full, kind="stable")[:50]
    assert np.array_equal(idx[0], order) and np.allclose(sc[0], full[order], atol=1e-6)
    allowed = np.array([0, 1, 0, 1, 0, 0, 0], np.uint8)                  # groups 1 and 3 only
    idx, sc = ix.search(q, 50, allowed)
    keep = np.flatnonzero(allowed[grp] != 0)
    want = keep[np.argsort(-full[keep], kind="stable")][:50]
    assert np.array_equal(idx[0], want)
    none = ix.search(q, 8, np.zeros(7, np.uint8))                        # nothing allowed -> padding only
    assert (none[0] == -1).all() and np.isinf(none[1]).all()
    few = ix.search(q, 1024, (np.arange(7) == 5).astype(np.uint8))      # fewer allowed rows than k -> the rest is padding
    n5 = int((grp == 5).sum())
    assert n5 < 1024 and (few[0][0][:n5] >= 0).all() and (few[0][0][n5:] == -1).all() and (grp[few[0][0][:n5]] == 5).all()
    short = ix.search(q, 64, np.array([0, 1], np.uint8))                 # bitmap shorter than the group ids in use: those rows are out
    assert set(grp[short[0][0][short[0][0] >= 0]]) <= {1}
    with pytest.raises(ValueError):
        ix.add(np.zeros((2, 512), np.float32))
    di, ds = ix.search_device(q, 50)
    assert di.is_cuda and ds.is_cuda and np.array_equal(di.cpu().numpy()[0], order)
    qd = torch.from_numpy(q).cuda()
    di2, _ = ix.search_device(qd, 50)                                     # device-resident query as well
    assert torch.equal(di, di2)


def test_bf16_index_scores_within_1e_3():
    """storage="bf16": rows rounded to bf16 (half the bytes per scan); scores within 1e-3 of the exact f32 index for unit
    vectors (measured ~2e-4), top-k equal to the f32 ranking except among rows closer than that."""
    from clearcam_amd.objects import EmbeddingIndex
    rng = np.random.default_rng(22)
    E = rng.standard_normal((40000, 768)).astype(np.float32); E /= np.linalg.norm(E, axis=1, keepdims=True)
    Q = rng.standard_normal((11, 768)).astype(np.float32); Q /= np.linalg.norm(Q, axis=1, keepdims=True)
    a, b = EmbeddingIndex(768, 40000), EmbeddingIndex(768, 1000, storage="bf16")
    a.add(E); b.add(E[:25000]); b.add(E[25000:])
    sa, sb = a.scores(Q), b.scores(Q)
    assert np.abs(sa - sb).max() <= 1e-3
    import torch
    Eb = torch.from_numpy(E).to(torch.bfloat16).float().numpy()           # the rounding the index applied
    assert np.abs(sb - Q @ Eb.T).max() <= 2e-6                            # ... and nothing else: f32 accumulation of exact products
    ia, _ = a.search(Q, 20); ib, scb = b.search(Q, 20)
    for r in range(11):
        assert len(set(ia[r]) & set(ib[r])) >= 17
        assert (np.diff(scb[r]) <= 0).all()


def test_bf16_index_add_is_ordered_behind_an_asynchronous_producer():
    """Rows handed over as a CUDA tensor that torch's stream is still writing (randn -> normalise, no sync) must be converted
    AFTER the producer finished: the conversion kernel runs on the index's own non-blocking stream.  A host query through
    search_device is uploaded on the caller's stream first (the C ABI's on_device flag covers query AND outputs)."""
    import torch
    from clearcam_amd.objects import EmbeddingIndex
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(5)
    b = EmbeddingIndex(768, 1000, storage="bf16")
    chunks = []
    for _ in range(4):
        big = torch.randn(8192, 8192, device=dev, generator=g)
        junk = (big @ big).sum()                                           # ~10 ms of queued work ahead of the producer
        e = torch.randn(60000, 768, device=dev, generator=g) + junk * 0
        e = e / e.norm(dim=1, keepdim=True)
        b.add(e)                                                           # no synchronize: the library must order itself
        chunks.append(e)
    E = torch.cat(chunks)
    q = torch.randn(3, 768, generator=torch.Generator().manual_seed(1)); q /= q.norm(dim=1, keepdim=True)
    want = q.to(dev) @ E.to(torch.bfloat16).float().T
    got = torch.from_numpy(b.scores(q.numpy())).to(dev)
    assert float((got - want).abs().max()) <= 2e-5
    idx_d, sc_d = b.search_device(q.numpy(), 10)                            # host query, device results
    idx_h, sc_h = b.search(q.numpy(), 10)
    assert np.array_equal(idx_d.cpu().numpy(), idx_h) and np.allclose(sc_d.cpu().numpy(), sc_h, atol=1e-6)
    b.close()


def test_sharded_index_over_rccl_world_size_1():
    """The RCCL plumbing of the N>1 search path on real hardware (one rank): nccl init, shard offsets, the device-resident
    all-gather + merge of ShardedIndex and the padded all-gather of ReplicatedIndex, against the local HIP index."""
    import subprocess
    import sys
    code = r'''
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.getcwd())
from clearcam_amd.dist import ReplicatedIndex, ShardedIndex
from clearcam_amd.objects import EmbeddingIndex
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29571")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
rng = np.random.default_rng(5)
E = rng.standard_normal((30000, 768)).astype(np.float32); E /= np.linalg.norm(E, axis=1, keepdims=True)
q = rng.standard_normal((3, 768)).astype(np.float32)
ix = EmbeddingIndex(768, 30000); ix.add(E)
sh = ShardedIndex(ix, device=dev)
assert (sh.row_offset, sh.total) == (0, 30000)
gi, gs = sh.search_device(q, 100)
assert gi.is_cuda and gs.is_cuda and gi.dtype == torch.int64
full = q @ E.T
order = np.argsort(-full, axis=1, kind="stable")[:, :100]
assert np.array_equal(gi.cpu().numpy(), order)
hi, hs = sh.search(q, 100)
assert np.array_equal(hi, order) and np.allclose(hs, np.take_along_axis(full, order, 1), atol=1e-6)
rep = ReplicatedIndex(EmbeddingIndex(768, 8))
new = torch.from_numpy(E[:300]).to(dev)
assert list(rep.add_local(new)) == [300] and len(rep.index) == 300
assert list(rep.add_local(new[:0])) == [0] and len(rep.index) == 300
ri, _ = rep.search(q, 5)
assert np.array_equal(ri, np.argsort(-(q @ E[:300].T), axis=1, kind="stable")[:, :5])
dist.barrier(); dist.destroy_process_group()
print("OK")
'''
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), timeout=600)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_clip_batches_in_flight_equal_encode():
    """cc_clip_set_in_flight / cc_clip_submit_image / cc_clip_wait: image batches queued round robin on the handle's slots give
    exactly the embeddings precompute_embedding gives - every depth, batch sizes changing between submissions, device and pinned
    host tensors."""
    import torch
    from clearcam_amd.arch import CLIP_B32
    from clearcam_amd.objects import OpenCLIP
    from clearcam_amd.weights import synthetic_clip_state_dict
    m = OpenCLIP(state_dict=synthetic_clip_state_dict(CLIP_B32, 7), arch=CLIP_B32, dtype="bf16", device=0)
    rng = np.random.default_rng(3)
    xs = [(rng.random((b, 3, 224, 224), dtype=np.float32) * 2 - 1) for b in (1, 3, 1, 8, 2, 3, 1)]
    ref = [m.precompute_embedding(x).numpy() for x in xs]
    assert all(np.isfinite(r).all() and abs(np.linalg.norm(r, axis=1) - 1).max() < 1e-4 for r in ref)
    for depth in (1, 2, 3):
        m.set_in_flight(depth)
        for host in (False, True):
            src = [torch.from_numpy(x).pin_memory() if host else torch.from_numpy(x).cuda() for x in xs]
            outs = [torch.full((len(x), 512), -2.0) for x in xs]
            outs = [o.pin_memory() if host else o.cuda() for o in outs]
            tickets = [m.submit_image(x, o) for x, o in zip(src, outs)]
            for t in tickets:
                m.wait(t, host=True)
            for i in range(len(xs)):
                assert np.array_equal(outs[i].cpu().numpy(), ref[i]), (depth, host, i)
    with pytest.raises(RuntimeError):
        m.wait(10 ** 9)
    m.set_in_flight(1)
    assert np.array_equal(m.precompute_embedding(xs[3]).numpy(), ref[3])
    m.close()

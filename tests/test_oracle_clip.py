"""CPU tests of the CLIP oracle, the tokenizer (against ids from the reference's own tokenizer) and search logic."""
import json
import os

import numpy as np
import pytest

from clearcam_amd.arch import CLIP_L14, CLIP_TINY
from clearcam_amd.weights import clip_shapes, synthetic_clip_state_dict
from oracle.clip_oracle import OpenCLIPOracle, pad_tokens, search_reference

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _vocab_available():
    from clearcam_amd.clip_tokenizer import find_vocab
    try:
        find_vocab()
        return True
    except FileNotFoundError:
        return False


def test_clip_structure_pins():
    s = clip_shapes(CLIP_L14)
    assert s["visual_conv1.weight"] == (1024, 3, 14, 14) and s["positional_embedding"] == (257, 1024)
    assert s["resblocks_img.23.in_proj_weight"] == (3072, 1024) and s["resblocks.11.attn_out_proj_weight"] == (768, 768)
    assert s["proj"] == (1024, 768) and s["text_projection"] == (768, 768) and s["token_embedding.weight"] == (49408, 768)
    n = sum(int(np.prod(v)) for v in s.values())
    assert abs(n / 1e6 - 427.6) < 0.5                                     # ViT-L/14 + text tower
    # MACs per image: 81.0 G = patch embed + 24 x (qkv + scores + pv + out + mlp) + proj
    L, D, M = 257, 1024, 4096
    macs = 256 * 588 * D + 24 * (L * D * 3 * D + 2 * 16 * L * L * 64 + L * D * D + 2 * L * D * M) + D * 768
    assert abs(macs / 1e9 - 81.013) < 0.01


def test_reference_golden_embeddings_are_consistent():
    pins = json.load(open(os.path.join(GOLD, "reference_pins.json")))
    e = {k: np.asarray(v["values"], np.float32) for k, v in pins["embeddings"].items()}
    assert set(e) == {"f40.jpg", "micra.jpg"} and all(v.shape == (768,) for v in e.values())
    for v in e.values():
        assert abs(np.linalg.norm(v) - 1.0) < 1e-5                         # L2-normalised (objects.py:132)
    assert abs(float(e["f40.jpg"] @ e["micra.jpg"]) - 0.5501478) < 1e-6
    assert np.allclose(e["f40.jpg"][:5], [-0.01891246, -0.13603427, -0.02724903, 0.01994121, -0.01596762], atol=1e-7)
    assert pins["text_image_cosine_ferrari_f40"] == 0.330654


@pytest.mark.skipif(not _vocab_available(), reason="bpe_simple_vocab_16e6.txt.gz not available")
def test_tokenizer_matches_reference_ids():
    from clearcam_amd.clip_tokenizer import SimpleTokenizer
    k = json.load(open(os.path.join(GOLD, "tokenizer_kats.json")))
    t = SimpleTokenizer()
    assert (t.vocab_size, t.sot_token_id, t.eot_token_id) == (k["vocab_size"], k["sot"], k["eot"]) == (49408, 49406, 49407)
    for c in k["cases"]:
        assert t.encode(c["text"]) == c["ids"], c["text"]
    m = t.tokens_for_model("ferrari f40")
    assert m.shape == (1, 77) and m[0, :6].tolist() == [49406, 9606, 325, 275, 271, 49407] and m[0, 6:].sum() == 0


def test_tokenizer_on_the_committed_merge_subset():
    """The same KATs (ids produced by the reference's own tokenizer) without clearcam's vocabulary file: the committed subset
    of its merge table carries every merge those strings can look up (tools/make_vocab_subset.py)."""
    from conftest import sparse_tokenizer
    k = json.load(open(os.path.join(GOLD, "tokenizer_kats.json")))
    t = sparse_tokenizer()
    assert (t.vocab_size, t.sot_token_id, t.eot_token_id) == (49408, 49406, 49407)
    for c in k["cases"]:
        assert t.encode(c["text"]) == c["ids"], c["text"]
    assert t.tokens_for_model("  Ferrari   F40 ")[0, :6].tolist() == [49406, 9606, 325, 275, 271, 49407]


def test_split_words_pattern():
    from clearcam_amd.clip_tokenizer import split_words
    assert split_words("it's 42nd st.!!") == ["it", "'s", "4", "2", "nd", "st", ".!!"]
    assert split_words("a<end_of_text>b") == ["a", "<end_of_text>", "b"]


def test_oracle_embeddings_are_unit_and_batch_invariant():
    sd = synthetic_clip_state_dict(CLIP_TINY, 4321)
    o = OpenCLIPOracle(sd, CLIP_TINY)
    x = np.random.default_rng(2).random((3, 3, 56, 56), dtype=np.float32) * 2 - 1
    e = o.precompute_embedding(x)
    assert e.shape == (3, 64) and np.allclose(np.linalg.norm(e, axis=1), 1, atol=1e-5)
    assert np.allclose(o.precompute_embedding(x[1:2])[0], e[1], atol=1e-5)
    toks = np.concatenate([pad_tokens([5, 9, 44], 77, 510, 511), pad_tokens(list(range(1, 40)), 77, 510, 511)])
    t = o.encode_tokens(toks)
    assert t.shape == (2, 64) and np.allclose(np.linalg.norm(t, axis=1), 1, atol=1e-5)
    assert np.allclose(o.encode_tokens(toks[1:])[0], t[1], atol=1e-5)
    # causal mask: tokens after EOT must not influence the pooled EOT row
    toks2 = toks.copy(); toks2[0, 10:] = 7
    assert np.allclose(o.encode_tokens(toks2)[0], t[0], atol=1e-6)


def test_search_reference_semantics():
    """objects.py:356-390: best score per track id, id-less crops kept, filters by camera / day."""
    rng = np.random.default_rng(0)
    q = rng.standard_normal(8).astype(np.float32); q /= np.linalg.norm(q)

    def emb(scale):
        v = (q * scale + rng.standard_normal(8).astype(np.float32) * (1.05 - scale) * 0.4)    # cosine grows with `scale`
        return (v / np.linalg.norm(v))[None]
    d = {
        "data/cameras/front/objects/2026-01-01/100.0_7_2.jpg": emb(1.0),
        "data/cameras/front/objects/2026-01-01/101.0_7_2.jpg": emb(0.5),      # same track id 7 -> only the best survives
        "data/cameras/back/objects/2026-01-01/102.0_9_0.jpg": emb(0.8),
        "data/cameras/back/objects/2026-01-02/103.0_11_0.jpg": emb(0.9),
        "data/cameras/back/objects/2026-01-02/notes.txt": emb(1.0),            # not a .jpg -> ignored
    }
    r = search_reference(d, q, top_k=10)
    assert [os.path.basename(p) for p, _ in r][0] == "100.0_7_2.jpg" and len(r) == 3
    assert all(r[i][1] >= r[i + 1][1] for i in range(len(r) - 1))
    assert len(search_reference(d, q, cam_name="back")) == 2
    assert len(search_reference(d, q, timestamp="2026-01-02")) == 1
    assert len(search_reference(d, q, top_k=1)) == 1


def test_cubic_resize_oracle_properties():
    """oracle/cv_resize_oracle.py: identity at equal size, constants preserved, and agreement (<= 1 LSB, fixed point vs
    float) with an independent float implementation of the same kernel (A=-0.75, replicate border): torch bicubic."""
    import torch
    from oracle.cv_resize_oracle import preprocess, resize_cubic_u8
    rng = np.random.default_rng(0)
    a = rng.integers(0, 256, (224, 224, 3), dtype=np.uint8)
    assert np.array_equal(resize_cubic_u8(a), a)
    assert np.unique(resize_cubic_u8(np.full((40, 30, 3), 200, np.uint8))).tolist() == [200]
    for h, w in [(57, 131), (3, 5), (400, 90)]:
        b = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        r = resize_cubic_u8(b)
        t = torch.from_numpy(b.astype(np.float32)).permute(2, 0, 1)[None]
        f = torch.nn.functional.interpolate(t, size=(224, 224), mode="bicubic", align_corners=False)[0].permute(1, 2, 0).numpy()
        assert np.abs(np.clip(np.rint(f), 0, 255) - r.astype(np.float32)).max() <= 1
    p = preprocess(a)
    assert p.shape == (3, 224, 224) and p.dtype == np.float32 and p.min() >= -1 and p.max() <= 1
    assert p[0, 0, 0] == np.float32((np.float32(a[0, 0, 0]) / np.float32(255.0) - np.float32(0.5)) / np.float32(0.5))


def test_adaface_oracle_structure():
    """IR-50 as models/adaface.py builds it: 24 blocks (3, 4, 14, 3 per stage), 43.6 M parameters, 7x7x512 features,
    unit-norm 512-d output that depends on the input."""
    from clearcam_amd.weights import ADAFACE_BLOCKS, synthetic_adaface_state_dict
    from oracle.adaface_oracle import BLOCKS, AdaFaceOracle
    assert BLOCKS == ADAFACE_BLOCKS and len(BLOCKS) == 24 and [b[2] for b in BLOCKS].count(2) == 4
    sd = synthetic_adaface_state_dict(777)
    assert abs(sum(v.size for v in sd.values()) / 1e6 - 43.62) < 0.05
    o = AdaFaceOracle(sd)
    rng = np.random.default_rng(0)
    a, b = (rng.integers(0, 256, (112, 112, 3), dtype=np.uint8) for _ in range(2))
    ea, eb = o(a), o(b)
    assert ea.shape == (1, 512) and abs(np.linalg.norm(ea) - 1) < 1e-5 and float((ea @ eb.T)[0, 0]) < 0.9
    assert np.array_equal(o(a), ea) and tuple(o.features(a).shape) == (1, 512, 7, 7)
    assert np.abs(o(a.astype(np.float32)) - ea).max() < 1e-6               # uint8 and float inputs agree


def test_blazeface_oracle_structure():
    """BlazeFace as models/blazeface.py builds it: 31 blocks, 896 anchors (512 on the 16x16 map, 384 on the 8x8 map), the
    'a row dies if a BETTER-ranked row overlaps it' rule (the (1,N,N) triu mask summed over axis 1, models/blazeface.py:232-234;
    pinned by tests/test_reference_run.py), zero rows mapped through the back-map like every other row."""
    import torch
    from clearcam_amd.weights import BLAZE_BLOCKS, blazeface_anchors, synthetic_blazeface_state_dict
    from oracle.blazeface_oracle import BLOCKS, BlazeFaceOracle
    assert BLOCKS == BLAZE_BLOCKS and len(BLOCKS) == 31 and [b[2] for b in BLOCKS].count(2) == 3
    a = blazeface_anchors()
    assert a.shape == (896, 4) and np.allclose(a[0], [1 / 32, 1 / 32, 1, 1]) and np.allclose(a[512], [1 / 16, 1 / 16, 1, 1])
    o = BlazeFaceOracle(synthetic_blazeface_state_dict(555))
    det = torch.zeros(896, 17)
    det[0] = torch.tensor([0.10, 0.10, 0.30, 0.30] + [0.0] * 12 + [0.95])      # best score ...
    det[1] = torch.tensor([0.11, 0.11, 0.31, 0.31] + [0.0] * 12 + [0.90])      # ... overlaps this lower-ranked row, which dies
    det[2] = torch.tensor([0.60, 0.60, 0.80, 0.80] + [0.0] * 12 + [0.92])
    post = o.postprocess(det)
    assert post[:, 16].tolist()[:4] == [pytest.approx(0.95), pytest.approx(0.92), 0.0, 0.0]      # sorted by score, third row zeroed
    img = np.random.default_rng(1).integers(0, 256, (360, 480, 3), dtype=np.uint8)
    out = o(img)
    scale, pad_top = min(256 / 480, 256 / 360), (256 - int(360 * min(256 / 480, 256 / 360))) // 2
    dead = out[out[:, 16] == 0]
    assert out.shape == (896, 17) and len(dead) > 800 and np.allclose(dead[:, 0], -pad_top / scale) and (out[:, 16] != 0).sum() > 10


def test_cv_warp_oracle_properties():
    """oracle/cv_warp_oracle.py: identity / integer shifts are exact, the border is constant 0, the linear resize agrees with an
    independent float bilinear (torch) to 1 LSB and with a hand 2x2 average on exact decimation."""
    import torch
    from oracle import cv_warp_oracle as cvo
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (90, 140, 3), dtype=np.uint8)
    eye = np.array([[1, 0, 0], [0, 1, 0]], float)
    assert np.array_equal(cvo.warp_affine_u8(img, eye, (140, 90)), img)
    sh = cvo.warp_affine_u8(img, np.array([[1, 0, 5], [0, 1, -3]], float), (140, 90))
    assert np.array_equal(sh[:87, 5:], img[3:, :135]) and not sh[88:].any() and not sh[:, :4].any()
    r = cvo.resize_linear_u8(img, (300, 200))
    t = torch.from_numpy(img.astype(np.float32)).permute(2, 0, 1)[None]
    f = torch.nn.functional.interpolate(t, size=(200, 300), mode="bilinear", align_corners=False)[0].permute(1, 2, 0).numpy()
    assert np.abs(np.rint(f) - r).max() <= 1
    half = cvo.resize_linear_u8(img, (70, 45)).astype(int)
    assert np.array_equal(half, (img[0::2, 0::2].astype(int) + img[0::2, 1::2] + img[1::2, 0::2] + img[1::2, 1::2] + 2) >> 2)
    R = cvo.get_rotation_matrix_2d((70, 45), 90.0, 1.0)
    assert np.allclose(R, [[0, 1, 25], [-1, 0, 115]], atol=1e-9)


def test_face_path_oracles_reproduce_golden():
    """tests/golden/face_path.npz (tools/make_golden.py) pins the face-path oracles against drift: AdaFace, BlazeFace and the
    OpenCV restatements must reproduce their committed outputs on this machine."""
    from clearcam_amd.weights import synthetic_adaface_state_dict, synthetic_blazeface_state_dict
    from oracle import cv_resize_oracle, cv_warp_oracle
    from oracle.adaface_oracle import AdaFaceOracle
    from oracle.blazeface_oracle import BlazeFaceOracle
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "face_path.npz"))
    assert np.abs(AdaFaceOracle(synthetic_adaface_state_dict(777))(g["face"]) - g["adaface"]).max() < 1e-5
    b = BlazeFaceOracle(synthetic_blazeface_state_dict(555))(g["img"])
    assert np.array_equal(b[:, 16] != 0, g["blazeface"][:, 16] != 0) and np.abs(b - g["blazeface"]).max() < 2e-2
    assert np.array_equal(cv_resize_oracle.resize_cubic_u8(g["crop"], 224), g["crop_cubic_224"])
    assert np.array_equal(cv_warp_oracle.resize_linear_u8(g["crop"], (200, 90)), g["crop_linear_200x90"])
    assert np.array_equal(cv_warp_oracle.warp_affine_u8(g["crop"], g["warp_M"], (150, 80)), g["crop_warp_150x80"])

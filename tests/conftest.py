import os
import sys

import numpy as np
import pytest

# the BPE vocabulary is clearcam's own data file; in the build container the reference checkout has it (test infrastructure only)
_REF_VOCAB = "/root/reference/utils/bpe_simple_vocab_16e6.txt.gz"
if os.path.exists(_REF_VOCAB):
    os.environ.setdefault("CLEARCAM_BPE_VOCAB", _REF_VOCAB)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_count() -> int:
    try:
        import torch
        return torch.cuda.device_count()
    except Exception:
        return 0


def pytest_collection_modifyitems(config, items):
    if _gpu_count() > 0:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def lib_path():
    """In-tree libclearcam_hip.so (built on demand with hipcc; cross-compiles without a GPU)."""
    from clearcam_amd import build
    return build.build()


@pytest.fixture(scope="session")
def sd_t():
    from clearcam_amd.weights import synthetic_yolov9_state_dict
    return synthetic_yolov9_state_dict("t", 1234)


@pytest.fixture(scope="session")
def sd_c():
    from clearcam_amd.weights import synthetic_yolov9_state_dict
    return synthetic_yolov9_state_dict("c", 1234)


def noise_frames(seed, b, h, w, dtype=np.uint8):
    f = np.random.default_rng(seed).integers(0, 256, (b, h, w, 3), dtype=np.uint8)
    return f if dtype == np.uint8 else f.astype(dtype)


def sparse_tokenizer():
    """SimpleTokenizer over tests/golden/bpe_merges_subset.json: the merges (published ranks) that the strings of
    tokenizer_kats.json can look up, so the string surface is testable where clearcam's vocabulary file is absent."""
    import json
    from clearcam_amd.clip_tokenizer import SimpleTokenizer
    sub = json.load(open(os.path.join(ROOT, "tests", "golden", "bpe_merges_subset.json")))["merges"]
    return SimpleTokenizer(sparse_merges={tuple(k.split(" ")): r for k, r in sub.items()})

import numpy as np, time, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from clearcam_amd.arch import CLIP_TINY, CLIP_L14
from clearcam_amd.weights import synthetic_clip_state_dict
from clearcam_amd.objects import OpenCLIP, EmbeddingIndex
from oracle.clip_oracle import OpenCLIPOracle, pad_tokens
which = sys.argv[1] if len(sys.argv) > 1 else "tiny"
arch = CLIP_TINY if which == "tiny" else CLIP_L14
sd = synthetic_clip_state_dict(arch, 4321)
o = OpenCLIPOracle(sd, arch)
B = 3 if which == "tiny" else 2
x = (np.random.default_rng(2).random((B, 3, arch.image_size, arch.image_size), dtype=np.float32) * 2 - 1)
toks = np.concatenate([pad_tokens([5, 9, 300 % arch.t_vocab], arch.t_ctx, arch.t_vocab - 2, arch.t_vocab - 1),
                       pad_tokens(list(range(1, 30)), arch.t_ctx, arch.t_vocab - 2, arch.t_vocab - 1)])
t = time.time(); ri = o.precompute_embedding(x); rt = o.encode_tokens(toks); print("oracle s", time.time() - t)
for dt in ("f32", "f16", "bf16"):
    m = OpenCLIP(state_dict=sd, arch=arch, dtype=dt)
    gi = m.precompute_embedding(x).numpy(); gt = m.encode_tokens(toks)
    ci = (gi * ri).sum(1); ct = (gt * rt).sum(1)
    print(dt, "img cos", ci, "max abs", np.abs(gi - ri).max(), "| txt cos", ct, "max abs", np.abs(gt - rt).max(), "norms", np.linalg.norm(gi, axis=1)[:2])
    t = time.time(); m.precompute_embedding(x); print("   img ms", (time.time() - t) * 1e3, "gpu", m.last_gpu_ms())
# index
rng = np.random.default_rng(3)
E = rng.standard_normal((50000, 768)).astype(np.float32); E /= np.linalg.norm(E, axis=1, keepdims=True)
q = rng.standard_normal((5, 768)).astype(np.float32); q /= np.linalg.norm(q, axis=1, keepdims=True)
ix = EmbeddingIndex(768, 60000); ix.add(E[:20000]); ix.add(E[20000:])
sc = ix.scores(q); ref = q @ E.T
print("index scores err", np.abs(sc - ref).max(), len(ix))
idx, s = ix.search(q, 100)
order = np.argsort(-sc, axis=1, kind="stable")[:, :100]
print("topk idx equal", np.array_equal(idx, order), "scores equal", np.array_equal(s, np.take_along_axis(sc, order, 1)))

"""ORACLE (test infrastructure, not product code): CPU fp32 restatement of the reference CLIP towers + search.

Restates ``models/objects.py`` of roryclear/clearcam — ``OpenCLIP.precompute_embedding`` (:94-133),
``OpenCLIP._encode_text`` / ``encode_text`` (:135-186) and the scoring/dedup logic of
``ObjectFinder.search`` (:356-390) — in PyTorch-CPU float32 / numpy.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module.

PARITY PINNING: the reference carries two golden 768-d image embeddings
(``test/clip_images/embeddings.pkl``) and one scalar (``test/test_clip.py:12``: cos = 0.330654), but both
need the real ViT-L/14 checkpoint (fetched from HuggingFace at construction, :91) and cv2 — neither exists
offline, so this oracle is unpinned against the TRAINED model's outputs ("parity unpinned" for those three numbers).
What pins it: the reference's own ``OpenCLIP`` class and ``encode_text`` executed unchanged over a PyTorch stand-in for
tinygrad (tools/refshim) on the seeded ViT-L/14 checkpoint, and the reference's own ``ObjectFinder.search`` /
``_load_all_embeddings`` run on pickles in its format (tests/golden/refrun_clip_l14.npz, refrun_search.npz;
tests/test_reference_run.py: embeddings within 2e-6 (measured 1e-7), search results identical);
the tokenizer against ids produced by the reference's own tokenizer run in this container
(tests/golden/tokenizer_kats.json), the golden pickle's self-consistency (unit norm, cos(f40,micra)),
and structural pins (parameter count, 81.0 GMAC/image).  The reference's pins are restated in
tests/golden/reference_pins.json so the scalar check can run the day a checkpoint is available.

tinygrad semantics: ``gelu()`` is the tanh approximation (B-5), LayerNorm eps 1e-5
biased variance (B-6), ``masked_fill(-inf)`` causal mask (B-8), image norm eps 1e-8 / text norm no eps.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

from clearcam_amd.arch import CLIP_L14, ClipArch


def _gelu_tanh(x: torch.Tensor) -> torch.Tensor:
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * x * x * x)))


class OpenCLIPOracle:
    def __init__(self, state_dict: Dict[str, np.ndarray], arch: ClipArch = CLIP_L14):
        self.a = arch
        self.sd = {k: torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)) for k, v in state_dict.items()}

    def _ln(self, x, name):
        return F.layer_norm(x, (x.shape[-1],), self.sd[name + ".weight"], self.sd[name + ".bias"], 1e-5)

    def _block(self, x, p, heads, out_w, out_b, mask: Optional[torch.Tensor]):
        """One residual block (:105-127 vision, :152-180 text)."""
        sd = self.sd
        h = self._ln(x, p + "ln_1")
        B, L, D = h.shape
        dh = D // heads
        qkv = h @ sd[p + "in_proj_weight"].T + sd[p + "in_proj_bias"]
        q, k, v = qkv.split(D, dim=-1)
        q, k, v = (t.reshape(B, L, heads, dh).transpose(1, 2) for t in (q, k, v))
        s = q @ k.transpose(-2, -1) * (1.0 / dh ** 0.5)
        if mask is not None:
            s = s.masked_fill(mask, float("-inf"))
        ctx = (s.softmax(-1) @ v).transpose(1, 2).reshape(B, L, D)
        x = x + (ctx @ sd[p + out_w].T + sd[p + out_b])
        h = self._ln(x, p + "ln_2")
        h = _gelu_tanh(h @ sd[p + "mlp_c_fc.weight"].T + sd[p + "mlp_c_fc.bias"])
        return x + (h @ sd[p + "mlp_c_proj.weight"].T + sd[p + "mlp_c_proj.bias"])

    @torch.no_grad()
    def precompute_embedding(self, x: np.ndarray) -> np.ndarray:
        """``precompute_embedding`` :94-133: (B,3,S,S) f32 -> (B,embed) L2-normalised (eps 1e-8)."""
        sd, a = self.sd, self.a
        x = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
        x = F.conv2d(x, sd["visual_conv1.weight"], None, stride=a.patch)
        x = x.reshape(x.shape[0], x.shape[1], -1).permute(0, 2, 1)
        cls = sd["class_embedding"].reshape(1, 1, -1).expand(x.shape[0], -1, -1)
        x = torch.cat((cls, x), 1) + sd["positional_embedding"]
        x = self._ln(x, "ln_pre")
        for i in range(a.v_layers):
            x = self._block(x, f"resblocks_img.{i}.", a.v_heads, "out_proj_weight", "out_proj_bias", None)
        x = self._ln(x, "ln_post")[:, 0, :] @ sd["proj"]
        return (x / (x.pow(2).sum(-1, keepdim=True).sqrt() + 1e-8)).numpy()

    @torch.no_grad()
    def encode_tokens(self, tokens: np.ndarray) -> np.ndarray:
        """``encode_text`` :145-186 over a batch: (B,ctx) int -> (B,embed); row b pools position argmax(tokens[b])
        (the reference has B=1 and pools ``text.argmax()`` of batch 0 — identical for B=1)."""
        sd, a = self.sd, self.a
        t = torch.from_numpy(np.asarray(tokens, dtype=np.int64))
        x = sd["token_embedding.weight"][t] + sd["positional_embedding_text"]
        mask = torch.ones(a.t_ctx, a.t_ctx).tril() == 0          # attn_mask < 0 (:76,:167)
        for i in range(a.t_layers):
            x = self._block(x, f"resblocks.{i}.", a.t_heads, "attn_out_proj_weight", "attn_out_proj_bias", mask)
        x = self._ln(x, "ln_final")
        x = x[torch.arange(x.shape[0]), t.argmax(-1)] @ sd["text_projection"]
        return (x / (x * x).sum(-1, keepdim=True).sqrt()).numpy()


def pad_tokens(ids: Sequence[int], ctx: int = 77, sot: int = 49406, eot: int = 49407) -> np.ndarray:
    """``_encode_text`` :136-140: [SOT] + ids + [EOT], zero padded to ctx (no truncation in the reference)."""
    t = [sot] + list(ids) + [eot]
    if len(t) < ctx:
        t += [0] * (ctx - len(t))
    return np.asarray([t], dtype=np.int32)


# ----------------------------------------------------------------------------------------------
# search  (models/objects.py:356-390, clearcam.py:1193 event_img_info)
# ----------------------------------------------------------------------------------------------

def event_img_info(stem: str):
    p = stem.split("_")
    return {"ts": int(float(p[0])), "object_id": int(p[1]), "class_id": int(p[2])}


def search_reference(embeddings: Dict[str, np.ndarray], text_embedding: np.ndarray, top_k: int = 10,
                     cam_name: Optional[str] = None, timestamp: Optional[str] = None):
    """``ObjectFinder.search`` :356-390 with an explicit embedding dict and query vector."""
    import os
    sims = []
    for path, emb in embeddings.items():
        if emb is None:
            continue
        norm = path.replace("\\", "/")
        if cam_name and f"/cameras/{cam_name}/" not in norm:
            continue
        if timestamp and f"/objects/{timestamp}/" not in norm and "/objects/video/" not in norm:
            continue
        sim = (emb @ text_embedding.T).item()
        fn = os.path.basename(path)
        if fn.lower().endswith(".jpg"):
            oid = event_img_info(fn.split(".jpg")[0])["object_id"] if "_" in fn else None
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

"""Multi-GPU layout of the hot path: one process per GPU, ``torch.distributed`` (backend "nccl" = RCCL over xGMI).

* detect / CLIP-encode: cameras (and their crops) are independent -> ``camera_rank`` pins each camera to one
  GPU; weights are replicated; there is NO data-path collective.
* search: the embedding matrix is row-sharded (a camera's crops live on its GPU).  A query is scanned locally
  on every rank (``EmbeddingIndex.search``) and the per-rank top-k lists — k x (f32 score, i64 global row id),
  800 B per rank at k=100 — are exchanged with ONE all-gather and merged on every rank.  Embeddings never move:
  replicating 1 M x 768 f32 would cost ~3 GB through a per-link-bound ring, the top-k exchange ~20 us.
"""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist


def camera_rank(camera_index: int, world_size: int) -> int:
    """Static camera -> GPU map (round robin keeps per-GPU stream counts within one of each other)."""
    return camera_index % world_size


def shard_offsets(local_rows: int, group=None, device: Optional[torch.device] = None) -> Tuple[int, int]:
    """(first global row id of this rank's shard, total rows) via an all-gather of shard sizes."""
    world = dist.get_world_size(group)
    t = torch.tensor([local_rows], dtype=torch.int64, device=device)
    sizes = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(sizes, t, group=group)
    sizes = [int(s.item()) for s in sizes]
    r = dist.get_rank(group)
    return sum(sizes[:r]), sum(sizes)


def merge_topk(all_idx: torch.Tensor, all_score: torch.Tensor, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """(Q, n) candidate (global id, score) lists -> top-k by score desc, ties by lower global id; id -1 sorts last."""
    big = torch.iinfo(torch.int64).max
    key_id = torch.where(all_idx < 0, torch.full_like(all_idx, big), all_idx)
    o1 = torch.argsort(key_id, dim=1, stable=True)
    s1, i1 = torch.gather(all_score, 1, o1), torch.gather(all_idx, 1, o1)
    o2 = torch.argsort(s1, dim=1, descending=True, stable=True)[:, :k]
    return torch.gather(i1, 1, o2), torch.gather(s1, 1, o2)


def allgather_topk_device(local_idx: torch.Tensor, local_score: torch.Tensor, row_offset: int, k: int,
                          group=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """The one exchange step of sharded search, tensors in, tensors out, on whatever device they live on.

    local_idx (int32/int64) / local_score (f32): (Q,k) per-rank top-k with LOCAL row ids (-1 / -inf padding).  Returns the
    merged global (Q,k) int64 ids and f32 scores, identical on every rank.  With CUDA tensors and the "nccl" backend
    (RCCL) nothing touches the host: one all-gather of (Q, 2, k) int64 per rank, then two sorts on the GPU."""
    idx = local_idx.to(torch.int64)
    idx = torch.where(idx >= 0, idx + row_offset, idx)
    # one message per rank: (Q, 2, k) int64 = [global ids | float32 score bits]
    packed = torch.stack([idx, local_score.contiguous().view(torch.int32).to(torch.int64)], dim=1).contiguous()
    world = dist.get_world_size(group)
    parts = [torch.empty_like(packed) for _ in range(world)]
    dist.all_gather(parts, packed, group=group)
    gi = torch.cat([t[:, 0] for t in parts], 1)
    gs = torch.cat([t[:, 1].to(torch.int32).view(torch.float32) for t in parts], 1)
    return merge_topk(gi, gs, k)


def allgather_topk(local_idx, local_score, row_offset: int, k: int, group=None,
                   device: Optional[torch.device] = None) -> Tuple[np.ndarray, np.ndarray]:
    """Host-array form of `allgather_topk_device` (numpy in, numpy out; the exchange itself runs on `device`)."""
    idx = torch.as_tensor(np.asarray(local_idx), dtype=torch.int64, device=device)
    sc = torch.as_tensor(np.asarray(local_score), dtype=torch.float32, device=device)
    mi, ms = allgather_topk_device(idx, sc, row_offset, k, group)
    return mi.cpu().numpy(), ms.cpu().numpy()


class ShardedIndex:
    """Row-sharded embedding index: local HBM scan + one RCCL all-gather of k candidates per rank.  When the local index
    can leave its result on the GPU (`search_device`, the HIP EmbeddingIndex) the candidate lists go scan -> all-gather ->
    merge without visiting the host; `search` copies the merged (Q,k) result out once, `search_device` not at all."""

    def __init__(self, index, group=None, device: Optional[torch.device] = None):
        self.index, self.group, self.device = index, group, device
        self.row_offset, self.total = shard_offsets(len(index), group, device)

    def search_device(self, q, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
        if hasattr(self.index, "search_device") and self.device is not None and torch.device(self.device).type == "cuda":
            idx, sc = self.index.search_device(q, k)
        else:                                                # CPU / gloo (tests) or an index without a device-side result
            i, s = self.index.search(q, k)
            idx = torch.as_tensor(np.asarray(i), dtype=torch.int64, device=self.device)
            sc = torch.as_tensor(np.asarray(s), dtype=torch.float32, device=self.device)
        return allgather_topk_device(idx, sc, self.row_offset, k, self.group)

    def search(self, q, k: int):
        mi, ms = self.search_device(q, k)
        return mi.cpu().numpy(), ms.cpu().numpy()


def allgather_rows(rows: torch.Tensor, group=None) -> Tuple[torch.Tensor, Sequence[int]]:
    """Pool newly computed embeddings: every rank contributes its (n_r, dim) rows (n_r may differ, may be 0) and receives
    the concatenation in rank order plus the per-rank counts.  One size exchange + one padded all-gather (RCCL over xGMI
    with CUDA tensors).  This is the "replicated index" alternative of SURVEY.md §8e: a camera's crops are encoded on its
    own GPU, then every GPU appends everybody's rows, so any rank can answer a cross-camera query alone.  At 768 f32 a
    batch of 256 crops per rank is 0.8 MB per rank per exchange; replicating a whole 1 M-row index this way costs 3 GB
    through a per-link-bound ring, which is why steady-state search uses ShardedIndex (top-k exchange) instead."""
    world = dist.get_world_size(group)
    n = torch.tensor([rows.shape[0]], dtype=torch.int64, device=rows.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    counts = [int(c.item()) for c in counts]
    cap = max(max(counts), 1)
    padded = torch.zeros((cap, rows.shape[1]), dtype=rows.dtype, device=rows.device)
    padded[:rows.shape[0]] = rows
    parts = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded, group=group)
    return torch.cat([p[:c] for p, c in zip(parts, counts)], 0), counts


class ReplicatedIndex:
    """Every rank holds every row: `add_local(rows)` pools this step's new embeddings from all ranks (allgather_rows) and
    appends them in rank order, so row ids agree everywhere; `search` is purely local."""

    def __init__(self, index, group=None):
        self.index, self.group = index, group

    def add_local(self, rows: torch.Tensor) -> Sequence[int]:
        allrows, counts = allgather_rows(rows, self.group)
        if allrows.shape[0]:
            self.index.add(allrows)
        return counts

    def search(self, q, k: int):
        return self.index.search(q, k)

"""world_size-2 gloo test of the one exchange step of sharded search (all-gather of per-rank top-k + merge)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from clearcam_amd.dist import ReplicatedIndex, allgather_rows, allgather_topk, camera_rank, merge_topk, shard_offsets


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, ret, backend="gloo"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # the pool's driver only supports dmabuf IPC (RCCL across processes)
    if backend == "nccl":
        torch.cuda.set_device(rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(3)
        E = rng.standard_normal((1000, 16)).astype(np.float32)
        E[700] = E[10]                                       # an exact tie across shards: lower global id must win
        q = rng.standard_normal((3, 16)).astype(np.float32)
        bounds = [0, 600, 1000]                              # ragged shards
        lo, hi = bounds[rank], bounds[rank + 1]
        dev = torch.device("cuda", rank) if backend == "nccl" else None      # RCCL moves device tensors
        off, total = shard_offsets(hi - lo, None, dev)
        assert (off, total) == (lo, 1000)
        k = 20
        sc = q @ E[lo:hi].T                                  # test scaffolding: the local scan runs in HIP in production
        order = np.argsort(-sc, axis=1, kind="stable")[:, :k]
        gi, gs = allgather_topk(order, np.take_along_axis(sc, order, 1), off, k, None, dev)
        full = q @ E.T
        ref = np.argsort(-full, axis=1, kind="stable")[:, :k]
        ok = np.array_equal(gi, ref) and np.allclose(gs, np.take_along_axis(full, ref, 1))
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_sharded_topk_allgather_gloo_world2():
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    assert ret.get(0) is True and ret.get(1) is True


@pytest.mark.gpu
def test_sharded_topk_allgather_nccl_world2():
    """The same exchange over RCCL, one rank per GPU: runs on the first box that has two GPUs (the pool's boxes have one: skipped there),
    so that N > 1 over xGMI is exercised without further work."""
    if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), ret, "nccl"), nprocs=2, join=True)
    assert ret.get(0) is True and ret.get(1) is True


class _HostIndex:                                              # stands in for the HBM matrix: gloo workers have no GPU
    def __init__(self):
        self.rows = np.zeros((0, 8), np.float32)

    def add(self, r):
        self.rows = np.concatenate([self.rows, np.asarray(r, np.float32)])

    def search(self, q, k):
        sc = q @ self.rows.T
        o = np.argsort(-sc, axis=1, kind="stable")[:, :k]
        return o, np.take_along_axis(sc, o, 1)


def _worker_rows(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ri = ReplicatedIndex(_HostIndex())
        ok = True
        for step, sizes in enumerate([(3, 5), (0, 2), (4, 0), (0, 0)]):          # ragged and empty contributions
            mine = torch.full((sizes[rank], 8), float(10 * step + rank))
            counts = ri.add_local(mine)
            ok &= list(counts) == list(sizes)
        expect = np.concatenate([np.full((n, 8), 10.0 * step + r, np.float32)
                                 for step, sizes in enumerate([(3, 5), (0, 2), (4, 0), (0, 0)]) for r, n in enumerate(sizes)])
        ok &= np.array_equal(ri.index.rows, expect)                              # same rows, same order on every rank
        rows, counts = allgather_rows(torch.arange(rank * 6, rank * 6 + 6, dtype=torch.float32).reshape(3, 2))
        ok &= rows.flatten().tolist() == list(map(float, range(12))) and list(counts) == [3, 3]
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_pooled_embeddings_replicated_index_gloo_world2():
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_rows, args=(2, _free_port(), ret), nprocs=2, join=True)
    assert ret.get(0) is True and ret.get(1) is True


def test_merge_topk_padding_and_ties():
    idx = torch.tensor([[5, -1, 2, 9, -1, 0]])
    sc = torch.tensor([[0.5, float("-inf"), 0.9, 0.5, float("-inf"), 0.1]])
    i, s = merge_topk(idx, sc, 5)
    assert i.tolist() == [[2, 5, 9, 0, -1]] and np.allclose(s[0, :4].numpy(), [0.9, 0.5, 0.5, 0.1])


def test_camera_rank_balanced():
    counts = np.bincount([camera_rank(c, 8) for c in range(64)], minlength=8)
    assert counts.tolist() == [8] * 8

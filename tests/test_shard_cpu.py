"""CPU: the N>1 path (row shards + ONE all-gather + merge) with world_size 2 over gloo.
Local search / merge are the oracle (injected); what is under test is the orchestration."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ragmeup_amd.shard import ShardedSearcher, shard_bounds


def test_shard_bounds_cover_exactly():
    for n in (0, 1, 7, 10_000_000, 10_000_003):
        for w in (1, 2, 3, 8):
            b = [shard_bounds(n, w, r) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            assert max(hi - lo for lo, hi in b) - min(hi - lo for lo, hi in b) <= 1


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker_l2(rank, world, port, n, nq, k, q_out):
    """Distance lists (RMU_METRIC_L2SQ: smaller = better): the cross-shard merge must keep the SMALLEST distances."""
    from oracle import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(5)
    x = (rng.standard_normal((n, 48)) * rng.uniform(0.5, 2.0, (n, 1))).astype(np.float32)     # un-normalised rows
    q = x[rng.permutation(n)[:nq]] + 0.05 * rng.standard_normal((nq, 48)).astype(np.float32)
    lo, hi = shard_bounds(n, world, rank)

    def local_search(qq, kk):
        s, r = O.flat_search(np.asarray(qq), x[lo:hi], kk, metric=O.METRIC_L2SQ)               # similarity = -distance
        return torch.from_numpy((-s).astype(np.float32)), torch.from_numpy(np.where(r >= 0, r + lo, -1))

    def merge(ps, pr, smaller_better=False):
        assert smaller_better
        s, r = O.merge_topk(-ps.numpy(), pr.numpy(), ps.shape[2])
        return torch.from_numpy((-s).astype(np.float32)), torch.from_numpy(r)

    ss = ShardedSearcher(local_search=local_search, merge=merge, smaller_better=True)
    s, r = ss.search(torch.from_numpy(q), k)
    gs, gr = O.flat_search(q, x, k, metric=O.METRIC_L2SQ)
    ok = bool(np.array_equal(r.numpy(), gr) and np.allclose(s.numpy(), -gs, atol=1e-4) and (np.diff(s.numpy(), axis=1) >= 0).all())
    q_out.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


def _worker(rank, world, port, n, nq, k, q_out):
    from oracle import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    x = O.make_corpus(n)
    q, _ = O.make_queries(x, nq)
    lo, hi = shard_bounds(n, world, rank)

    def local_search(qq, kk):
        s, r = O.flat_search(np.asarray(qq), x[lo:hi], kk)
        return torch.from_numpy(s.astype(np.float32)), torch.from_numpy(np.where(r >= 0, r + lo, -1))

    def merge(ps, pr):
        s, r = O.merge_topk(ps.numpy(), pr.numpy(), ps.shape[2])
        return torch.from_numpy(s.astype(np.float32)), torch.from_numpy(r)

    ss = ShardedSearcher(local_search=local_search, merge=merge)
    s, r = ss.search(torch.from_numpy(q), k)
    gs, gr = O.flat_search(q, x, k)
    ok = bool(np.array_equal(r.numpy(), gr) and np.allclose(s.numpy(), gs, atol=1e-6))
    q_out.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n,nq,k", [(5000, 9, 10), (33, 4, 20)])
def test_two_rank_sharded_search_equals_global(n, nq, k):
    ctx = mp.get_context("spawn")
    q_out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, nq, k, q_out)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q_out.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True), (1, True)]


def test_two_rank_sharded_l2_search_keeps_the_smallest_distances():
    ctx = mp.get_context("spawn")
    q_out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_l2, args=(r, 2, port, 3000, 7, 10, q_out)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q_out.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True), (1, True)]


def test_sharded_searcher_takes_the_merge_direction_from_the_index_metric():
    from ragmeup_amd import _native as N

    class Idx:
        metric = N.METRIC_L2SQ

        def search(self, q, k, row_base=0):
            raise AssertionError

    assert ShardedSearcher(index=Idx(), merge=lambda *a, **k: None).smaller_better is True
    Idx.metric = N.METRIC_IP
    assert ShardedSearcher(index=Idx(), merge=lambda *a, **k: None).smaller_better is False

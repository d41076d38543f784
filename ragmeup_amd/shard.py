"""Row-sharded flat search across the GPUs of one node.

Rank r owns corpus rows [r*N/W, (r+1)*N/W); queries are replicated; every rank runs the identical
local scan (global row ids via row_base) and ONE all-gather (RCCL over xGMI; the message is
B*k*12 bytes per rank, latency-bound) hands every rank all per-shard top-k lists, which
rmu_topk_merge folds into the final top-k.  No reference counterpart (the reference is
single-process); the local step serves server/RAGHelper.py:497-499.

`local_search` / `merge` are injectable so the orchestration (offsets, packing, ordering) is
testable on CPU with gloo; the defaults are the HIP paths and raise if librmu.so is missing.
"""
from __future__ import annotations

from typing import Callable

import torch
import torch.distributed as dist


def shard_bounds(n_rows: int, world: int, rank: int) -> tuple[int, int]:
    """Contiguous row range of `rank` (first ranks take the remainder)."""
    base, rem = divmod(n_rows, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class ShardedSearcher:
    def __init__(self, index=None, row_base: int = 0, group=None,
                 local_search: Callable | None = None, merge: Callable | None = None, force_collective: bool = False):
        self.index = index
        self.row_base = int(row_base)
        self.group = group
        if local_search is None:
            if index is None:
                raise ValueError("need an index or a local_search callable")
            local_search = lambda q, k: index.search(q, k, row_base=self.row_base)
        if merge is None:
            from .index import topk_merge
            merge = topk_merge
        self._search = local_search
        self._merge = merge
        self.force_collective = bool(force_collective)   # run the all-gather + merge even at world size 1 (tests)

    @property
    def world(self) -> int:
        return dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1

    def search(self, q, k: int):
        """q: [B, d] (torch tensor on this rank's device, identical on every rank).
        Returns (scores [B,k] f32, rows [B,k] i64) -- identical on every rank."""
        s, r = self._search(q, k)
        w = self.world
        if w == 1 and not (self.force_collective and dist.is_initialized()):
            return s, r
        s = torch.as_tensor(s).contiguous()
        r = torch.as_tensor(r).contiguous()
        nq = s.shape[0]
        nb_s, nb_r = nq * k * 4, nq * k * 8
        # one struct-of-arrays byte buffer -> a single collective
        send = torch.empty(nb_s + nb_r, dtype=torch.uint8, device=s.device)
        send[:nb_s] = s.view(torch.uint8).reshape(-1)
        send[nb_s:] = r.view(torch.uint8).reshape(-1)
        recv = torch.empty(w * (nb_s + nb_r), dtype=torch.uint8, device=s.device)
        dist.all_gather_into_tensor(recv, send, group=self.group)
        recv = recv.view(w, nb_s + nb_r)
        ps = recv[:, :nb_s].contiguous().view(torch.float32).view(w, nq, k)
        pr = recv[:, nb_s:].contiguous().view(torch.int64).view(w, nq, k)
        return self._merge(ps, pr)

"""Row-sharded flat search across the GPUs of one node.

Rank r owns corpus rows [r*N/W, (r+1)*N/W); queries are replicated; every rank runs the identical
local scan (global row ids via row_base) and ONE all-gather (RCCL over xGMI; the message is
B*k*12 bytes per rank, latency-bound) hands every rank all per-shard top-k lists, which
rmu_topk_merge folds into the final top-k.  No reference counterpart (the reference is
single-process); the local step serves server/RAGHelper.py:497-499.

Two transports for the exchange:
  * `NativeComm` -- the C-ABI path (`rmu_comm_*`, `rmu_shard_allgather_topk`): RCCL called from librmu.so itself, pack +
    all-gather + merge in one call on one stream; any host language can drive it.  The 128-byte RCCL unique id travels
    through whatever side channel the host has (here: torch.distributed's broadcast, or a file).
  * torch.distributed `all_gather_into_tensor` + `rmu_topk_merge` (the default when no NativeComm is given; with the
    gloo backend this is also what the CPU tests exercise).
`local_search` / `merge` are injectable so the orchestration (offsets, packing, ordering) is
testable on CPU with gloo; the defaults are the HIP paths and raise if librmu.so is missing.
"""
from __future__ import annotations

import ctypes
from typing import Callable

import torch
import torch.distributed as dist


class NativeComm:
    """RCCL communicator owned by librmu.so (include/rmu.h: rmu_comm_*).  `NativeComm.from_torch_dist()` bootstraps it from
    an initialised torch.distributed group (rank 0 creates the unique id, a broadcast distributes it)."""

    def __init__(self, unique_id: bytes, world: int, rank: int, device: int | None = None):
        from . import _native as N
        self._N, self._lib = N, N.lib()
        if device is not None:
            N.check(self._lib.rmu_init(int(device)), "rmu_init")
        if len(unique_id) != N.COMM_ID_BYTES:
            raise ValueError(f"unique id must be {N.COMM_ID_BYTES} bytes")
        h = ctypes.c_void_p()
        buf = ctypes.create_string_buffer(unique_id, N.COMM_ID_BYTES)
        N.check(self._lib.rmu_comm_init(ctypes.byref(h), buf, int(world), int(rank)), "rmu_comm_init")
        self._h = h
        # what the communicator itself reports (rmu_comm_world), not what the caller asked for
        w, r = ctypes.c_int(), ctypes.c_int()
        N.check(self._lib.rmu_comm_world(h, ctypes.byref(w), ctypes.byref(r)), "rmu_comm_world")
        self.world, self.rank = int(w.value), int(r.value)
        if (self.world, self.rank) != (int(world), int(rank)):
            raise RuntimeError(f"rmu_comm_init: asked for rank {rank} of {world}, the communicator reports {self.rank} of {self.world}")

    @staticmethod
    def unique_id() -> bytes:
        from . import _native as N
        buf = ctypes.create_string_buffer(N.COMM_ID_BYTES)
        N.check(N.lib().rmu_comm_unique_id(buf), "rmu_comm_unique_id")
        return buf.raw

    @classmethod
    def from_torch_dist(cls, device: int, group=None) -> "NativeComm":
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        ids = [cls.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0, group=group)
        return cls(ids[0], world, rank, device=device)

    def allgather_topk(self, s, r, smaller_better: bool = False, stream: int | None = None):
        """Local [nq, k] torch CUDA lists -> merged global [nq, k] (identical on every rank).
        `stream`: a non-zero hipStream_t handle -> pack, all-gather and merge are ORDERED on it and the call returns without a
        host synchronisation (the inputs must have been produced on that stream or be ordered before it); default: the inputs'
        torch stream is drained first and the result is complete on return."""
        N = self._N
        s = s.contiguous()
        r = r.contiguous()
        nq, k = s.shape
        out_s, out_r = torch.empty_like(s), torch.empty_like(r)
        if not stream:
            torch.cuda.current_stream(s.device).synchronize()
        N.check(self._lib.rmu_shard_allgather_topk(self._h, s.data_ptr(), r.data_ptr(), nq, k,
                                                   N.F_Q_DEVICE | N.F_OUT_DEVICE | (N.F_SMALLER_BETTER if smaller_better else 0),
                                                   out_s.data_ptr(), out_r.data_ptr(), int(stream or 0)), "rmu_shard_allgather_topk")
        return out_s, out_r

    def close(self):
        if getattr(self, "_h", None):
            self._lib.rmu_comm_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def shard_bounds(n_rows: int, world: int, rank: int) -> tuple[int, int]:
    """Contiguous row range of `rank` (first ranks take the remainder)."""
    base, rem = divmod(n_rows, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class ShardedSearcher:
    def __init__(self, index=None, row_base: int = 0, group=None,
                 local_search: Callable | None = None, merge: Callable | None = None, force_collective: bool = False,
                 comm: NativeComm | None = None, smaller_better: bool | None = None):
        self.index = index
        # distance lists (an RMU_METRIC_L2SQ index: smaller = better) must be merged the other way round; by default the
        # index's own metric decides
        if smaller_better is None:
            from . import _native as N
            smaller_better = getattr(index, "metric", None) == N.METRIC_L2SQ
        self.smaller_better = bool(smaller_better)
        self.row_base = int(row_base)
        self.group = group
        self.comm = comm                       # the C-ABI RCCL path when given (else torch.distributed)
        self._native_local = local_search is None
        self._stream = None                    # the searcher's own side stream (stream-ordered steps, torch CUDA queries)
        self._local = {}                       # (batch, k) -> the local (scores, rows) buffers of the stream-ordered step
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

    def _search_stream_ordered(self, q, k: int):
        """The sharded step with ZERO host synchronisations: local scan -> pack -> ONE RCCL all-gather -> merge, all ordered on
        the searcher's own side stream (torch's default stream is the NULL stream, which the C-ABI reads as "internal + blocking");
        the side stream waits for the caller's stream first (the queries) and the caller's stream waits for it afterwards (the
        results), both by device-side events."""
        cur = torch.cuda.current_stream(q.device)
        if self._stream is None:
            self._stream = torch.cuda.Stream(q.device)
        side = self._stream
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            # the local lists live in buffers the searcher keeps per (batch, k): every step is ordered on `side`, so the next local scan
            # overwrites them only after this step's pack has read them (no allocator traffic per step)
            key = (int(q.shape[0]), int(k))
            loc = self._local.get(key)
            if loc is None or loc[0].device != q.device:
                loc = (torch.empty(key, dtype=torch.float32, device=q.device), torch.empty(key, dtype=torch.int64, device=q.device))
                if len(self._local) > 8:
                    self._local.clear()
                self._local[key] = loc
            s, r = self.index.search(q, k, row_base=self.row_base, stream=side.cuda_stream, out=loc)
            out = self.comm.allgather_topk(s, r, smaller_better=self.smaller_better, stream=side.cuda_stream)
        cur.wait_stream(side)
        # Allocator bookkeeping.  q was allocated on the caller's stream and is READ on `side`: q.record_stream(side).  The outputs were
        # allocated under `side` (their home stream) and are used on the CALLER's stream from here on: recording `side` on them is a no-op
        # -- once the caller dropped them the allocator could hand the blocks to the next side-stream search while the caller's stream is
        # still reading -- so the stream to record is `cur`.  (s, r never leave the side stream: the searcher keeps them.)
        q.record_stream(side)
        for t in out:
            t.record_stream(cur)
        return out

    @property
    def world(self) -> int:
        return dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1

    def search(self, q, k: int, out=None):
        """q: [B, d] (torch tensor on this rank's device, identical on every rank).
        Returns (scores [B,k] f32, rows [B,k] i64) -- identical on every rank.  `out`: caller-owned result tensors for the single-process
        native path (FlatIndex.search's `out`)."""
        if out is not None and self._native_local and self.comm is None and self.world == 1 and not self.force_collective:
            return self.index.search(q, k, row_base=self.row_base, out=out)
        if (self._native_local and self.comm is not None and (self.comm.world > 1 or self.force_collective)
                and torch.is_tensor(q) and q.is_cuda):
            return self._search_stream_ordered(q, k)
        s, r = self._search(q, k)
        if self.comm is not None and (self.comm.world > 1 or self.force_collective):
            return self.comm.allgather_topk(torch.as_tensor(s), torch.as_tensor(r), smaller_better=self.smaller_better)
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
        return self._merge(ps, pr, smaller_better=True) if self.smaller_better else self._merge(ps, pr)

"""FlatIndex -- Python handle on the HBM-resident flat index behind librmu.so.

Host-side mirror of what the reference reaches through langchain_milvus.Milvus /
langchain_postgres.PGVector (server/RAGHelper.py:385-434, 497-499): insert, exact top-k,
row fetch for MMR, delete.  Arrays are numpy on the host or torch CUDA tensors (device pointers
cross the C-ABI as integers); nothing here computes a score on the CPU.
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import _native as N


def _is_torch_cuda(t) -> bool:
    return hasattr(t, "data_ptr") and hasattr(t, "is_cuda") and bool(t.is_cuda)


class FlatIndex:
    def __init__(self, dim: int, metric: int = N.METRIC_IP, capacity_hint: int = 0, device: int | None = None):
        self._lib = N.lib()
        if device is not None:
            N.check(self._lib.rmu_init(int(device)), "rmu_init")
        self.dim = int(dim)
        self.metric = int(metric)
        h = ctypes.c_void_p()
        N.check(self._lib.rmu_index_create(ctypes.byref(h), self.dim, self.metric, int(capacity_hint)),
                "rmu_index_create")
        self._h = h

    # -- persistence ---------------------------------------------------------------------
    def save(self, path: str):
        N.check(self._lib.rmu_index_save(self._h, str(path).encode()), "rmu_index_save")

    @classmethod
    def load(cls, path: str, device: int | None = None) -> "FlatIndex":
        self = cls.__new__(cls)
        self._lib = N.lib()
        if device is not None:
            N.check(self._lib.rmu_init(int(device)), "rmu_init")
        h = ctypes.c_void_p()
        N.check(self._lib.rmu_index_load(ctypes.byref(h), str(path).encode()), "rmu_index_load")
        self._h = h
        d = ctypes.c_int()
        N.check(self._lib.rmu_index_dim(h, ctypes.byref(d)), "rmu_index_dim")
        self.dim = int(d.value)
        m = ctypes.c_int()
        N.check(self._lib.rmu_index_metric(h, ctypes.byref(m)), "rmu_index_metric")
        self.metric = int(m.value)                 # what the file header says, not an assumption
        return self

    # -- lifecycle ---------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self._lib.rmu_index_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self) -> int:
        n = ctypes.c_int64()
        N.check(self._lib.rmu_index_size(self._h, ctypes.byref(n)), "rmu_index_size")
        return int(n.value)

    def stats(self) -> dict:
        """rmu_index_stat: allocated capacity, re-allocations by add() and their wall time, live rows."""
        out = {}
        for name, what in (("capacity", N.STAT_CAPACITY), ("grow_count", N.STAT_GROW_COUNT), ("grow_ms", N.STAT_GROW_MS),
                           ("live_rows", N.STAT_LIVE_ROWS)):
            v = ctypes.c_double()
            N.check(self._lib.rmu_index_stat(self._h, what, ctypes.byref(v)), "rmu_index_stat")
            out[name] = float(v.value) if name == "grow_ms" else int(v.value)
        return out

    def reserve(self, rows: int):
        """Make room for `rows` rows in all (grow-only; see rmu_index_reserve): a re-allocation waits for the device, so a caller that is
        about to leave work in flight does it first."""
        N.check(self._lib.rmu_index_reserve(self._h, int(rows)), "rmu_index_reserve")

    # -- mutation ----------------------------------------------------------------------------------
    def add(self, vecs) -> int:
        """Append rows; returns the row id of the first one."""
        first = ctypes.c_int64()
        if _is_torch_cuda(vecs):
            import torch
            v = vecs.detach().to(torch.float32).contiguous()
            if v.ndim != 2 or v.shape[1] != self.dim:
                raise ValueError(f"expected [n, {self.dim}] got {tuple(v.shape)}")
            torch.cuda.current_stream().synchronize()
            N.check(self._lib.rmu_index_add(self._h, v.data_ptr(), v.shape[0], 1, ctypes.byref(first)), "rmu_index_add")
        else:
            v = np.ascontiguousarray(vecs, dtype=np.float32)
            if v.ndim != 2 or v.shape[1] != self.dim:
                raise ValueError(f"expected [n, {self.dim}] got {v.shape}")
            N.check(self._lib.rmu_index_add(self._h, v.ctypes.data, v.shape[0], 0, ctypes.byref(first)), "rmu_index_add")
        return int(first.value)

    def remove_rows(self, rows) -> int:
        r = np.ascontiguousarray(rows, dtype=np.int64)
        cnt = ctypes.c_int64()
        N.check(self._lib.rmu_index_remove_rows(self._h, r.ctypes.data, r.shape[0], ctypes.byref(cnt)),
                "rmu_index_remove_rows")
        return int(cnt.value)

    def get_rows(self, rows) -> np.ndarray:
        r = np.ascontiguousarray(rows, dtype=np.int64)
        out = np.empty((r.shape[0], self.dim), dtype=np.float32)
        N.check(self._lib.rmu_index_get_rows(self._h, r.ctypes.data, r.shape[0], out.ctypes.data), "rmu_index_get_rows")
        return out

    def mmr(self, q, rows, k: int, lambda_mult: float = 0.5) -> np.ndarray:
        """Batched greedy MMR on the device: q [nq, dim] fp32, rows [nq, fetch_k] int64 candidate row ids (-1 = absent, as
        `search` pads them; without a row_base) -> positions [nq, k] int32 into each candidate list (-1 = none)."""
        qq = np.ascontiguousarray(q, dtype=np.float32)
        if qq.ndim == 1:
            qq = qq[None]
        r = np.ascontiguousarray(rows, dtype=np.int64)
        if r.ndim == 1:
            r = r[None]
        if r.shape[0] != qq.shape[0] or qq.shape[1] != self.dim:
            raise ValueError("mmr: q must be [nq, dim] and rows [nq, fetch_k]")
        out = np.empty((qq.shape[0], int(k)), dtype=np.int32)
        N.check(self._lib.rmu_index_mmr(self._h, qq.ctypes.data, qq.shape[0], r.ctypes.data, r.shape[1], int(k),
                                        float(lambda_mult), 0, out.ctypes.data), "rmu_index_mmr")
        return out

    def search_mmr(self, q, fetch_k: int, k: int, lambda_mult: float = 0.5, row_base: int = 0):
        """`search(q, fetch_k)` + `mmr(...)` in one call with one host round trip (rmu_index_search_mmr; the reference's per-request
        retriever call): q [nq, dim] host fp32 -> (rows [nq, k] int64 in pick order, -1 padded; scores [nq, k] fp32)."""
        qq = np.ascontiguousarray(q, dtype=np.float32)
        if qq.ndim == 1:
            qq = qq[None]
        if qq.shape[1] != self.dim:
            raise ValueError(f"expected [nq, {self.dim}] got {qq.shape}")
        nq = qq.shape[0]
        rows = np.empty((nq, int(k)), dtype=np.int64)
        scores = np.empty((nq, int(k)), dtype=np.float32)
        N.check(self._lib.rmu_index_search_mmr(self._h, qq.ctypes.data, nq, int(fetch_k), int(k), float(lambda_mult), int(row_base),
                                               rows.ctypes.data, scores.ctypes.data), "rmu_index_search_mmr")
        return rows, scores

    # -- search ------------------------------------------------------------------------------------
    def search(self, q, k: int, row_base: int = 0, stream: int | None = None, out=None):
        """Exact top-k.  numpy in -> numpy out; torch CUDA in -> torch CUDA out (same device).
        `stream` (torch CUDA input only): a non-zero hipStream_t handle (`torch.cuda.Stream.cuda_stream`) the search is ORDERED
        on -- the call returns without any host synchronisation (include/rmu.h stream contract) and the outputs are valid for
        work queued behind it on that stream.  Default: complete on return."""
        if _is_torch_cuda(q):
            import torch
            qq = q.detach().to(torch.float32).contiguous()
            if qq.ndim == 1:
                qq = qq[None]
            nq = qq.shape[0]
            if out is not None:                  # caller-owned result tensors (a serving loop's)
                out_s, out_r = out
                if (out_s.shape != (nq, k) or out_r.shape != (nq, k) or out_s.dtype != torch.float32 or out_r.dtype != torch.int64
                        or not out_s.is_contiguous() or not out_r.is_contiguous() or out_s.device != qq.device or out_r.device != qq.device):
                    raise ValueError("out must be (float32 [nq, k], int64 [nq, k]) contiguous tensors on the queries' device")
            else:
                out_s = torch.empty((nq, k), dtype=torch.float32, device=qq.device)
                out_r = torch.empty((nq, k), dtype=torch.int64, device=qq.device)
            if not stream:
                torch.cuda.current_stream().synchronize()
            N.check(self._lib.rmu_index_search(self._h, qq.data_ptr(), nq, int(k), N.F_Q_DEVICE | N.F_OUT_DEVICE,
                                               int(row_base), out_s.data_ptr(), out_r.data_ptr(), int(stream or 0)),
                    "rmu_index_search")
            return out_s, out_r
        qq = np.ascontiguousarray(q, dtype=np.float32)
        if qq.ndim == 1:
            qq = qq[None]
        if qq.shape[1] != self.dim:
            raise ValueError(f"expected [nq, {self.dim}] got {qq.shape}")
        nq = qq.shape[0]
        out_s = np.empty((nq, k), dtype=np.float32)
        out_r = np.empty((nq, k), dtype=np.int64)
        N.check(self._lib.rmu_index_search(self._h, qq.ctypes.data, nq, int(k), 0, int(row_base),
                                           out_s.ctypes.data, out_r.ctypes.data, 0), "rmu_index_search")
        return out_s, out_r

    def set_screening(self, on: bool = True):
        """RMU_OPT_SCREEN: allow (default) or forbid the fp16 screening path; results are identical either way."""
        N.check(self._lib.rmu_index_set_option(self._h, N.OPT_SCREEN, 1 if on else 0), "rmu_index_set_option")

    def set_screen_min_batch(self, n: int = 0):
        """RMU_OPT_SCREEN_MIN_NQ: n > 0 sends every batch of >= n queries through the screening path whatever the corpus size
        (0 restores the default heuristics); results are identical either way."""
        N.check(self._lib.rmu_index_set_option(self._h, N.OPT_SCREEN_MIN_NQ, int(n)), "rmu_index_set_option")

    def set_ladder(self, ratio: int = 0, first: int = 0):
        """Tuning (RMU_OPT_LADDER_RATIO / RMU_OPT_LADDER_FIRST): geometry of the screening path's threshold ladder -- growth ratio above
        64k rows and the size of the first range; 0 = the defaults.  Results are identical for every value (tools/ladder_sweep.py)."""
        N.check(self._lib.rmu_index_set_option(self._h, N.OPT_LADDER_RATIO, int(ratio)), "rmu_index_set_option")
        N.check(self._lib.rmu_index_set_option(self._h, N.OPT_LADDER_FIRST, int(first)), "rmu_index_set_option")

    def screen_candidates(self, q):
        """Test hook (rmu_index_screen_candidates): per query the screening pass's 32 candidates ->
        (approx scores [nq,32], rows [nq,32], exact fp32 scores of the same rows [nq,32], EPS [nq])."""
        qq = np.ascontiguousarray(q, dtype=np.float32)
        if qq.ndim == 1:
            qq = qq[None]
        nq = qq.shape[0]
        ap = np.empty((nq, 32), np.float32)
        ro = np.empty((nq, 32), np.int64)
        ex = np.empty((nq, 32), np.float32)
        eps = np.empty((nq,), np.float32)
        N.check(self._lib.rmu_index_screen_candidates(self._h, qq.ctypes.data, nq, ap.ctypes.data, ro.ctypes.data, ex.ctypes.data,
                                                      eps.ctypes.data), "rmu_index_screen_candidates")
        return ap, ro, ex, eps

    # -- measurement hooks (bench.py) ----------------------------------------------------------------
    def set_timing(self, on: bool = True):
        self._lib.rmu_set_timing(1 if on else 0)

    def last_scan_ms(self) -> float:
        return float(self._lib.rmu_last_scan_ms())

    def last_search_ms(self) -> float:
        return float(self._lib.rmu_last_search_ms())

    def last_screened(self) -> int:
        """>0: answered by the fp16 screening pass + exact re-score; 0: exact scan; <0: screened, with that many queries
        re-run on the exact scan."""
        return int(self._lib.rmu_last_screened())

    def last_geometry(self) -> dict:
        g, b, l, p = (ctypes.c_int() for _ in range(4))
        self._lib.rmu_last_scan_geometry(ctypes.byref(g), ctypes.byref(b), ctypes.byref(l), ctypes.byref(p))
        return {"grid": g.value, "block": b.value, "lds_bytes": l.value, "launches": p.value}


def topk_merge(part_scores, part_rows, k: int | None = None, smaller_better: bool = False, stream: int | None = None):
    """Merge [parts, nq, k] shard lists (numpy or torch CUDA) -> [nq, k].  smaller_better: the scores are distances
    (lists of an RMU_METRIC_L2SQ index).  `stream` as in FlatIndex.search (torch CUDA lists only)."""
    lib = N.lib()
    fl = N.F_SMALLER_BETTER if smaller_better else 0
    if _is_torch_cuda(part_scores):
        import torch
        s = part_scores.detach().to(torch.float32).contiguous()
        r = part_rows.detach().to(torch.int64).contiguous()
        parts, nq, kk = s.shape
        out_s = torch.empty((nq, kk), dtype=torch.float32, device=s.device)
        out_r = torch.empty((nq, kk), dtype=torch.int64, device=s.device)
        if not stream:
            torch.cuda.current_stream().synchronize()
        N.check(lib.rmu_topk_merge(s.data_ptr(), r.data_ptr(), parts, nq, kk, N.F_Q_DEVICE | N.F_OUT_DEVICE | fl,
                                   out_s.data_ptr(), out_r.data_ptr(), int(stream or 0)), "rmu_topk_merge")
        return out_s, out_r
    s = np.ascontiguousarray(part_scores, dtype=np.float32)
    r = np.ascontiguousarray(part_rows, dtype=np.int64)
    parts, nq, kk = s.shape
    out_s = np.empty((nq, kk), dtype=np.float32)
    out_r = np.empty((nq, kk), dtype=np.int64)
    N.check(lib.rmu_topk_merge(s.ctypes.data, r.ctypes.data, parts, nq, kk, fl, out_s.ctypes.data, out_r.ctypes.data, 0),
            "rmu_topk_merge")
    return out_s, out_r

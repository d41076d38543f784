"""Test-side driver: runs bench.py's N > 1 control flow (sharding, broadcast of the queries, exchange, barrier + max-over-ranks
timing, rank-0 JSON line) on CPU with gloo.  The local index is an ORACLE-backed stand-in injected through bench.main's `hooks`
(test infrastructure: neither bench.py nor anything under ragmeup_amd/ imports the oracle for this).  Started plainly
(`python tests/bench_world2_driver.py --gpus 2 ...`) it exercises bench.py's self-launch as well."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

import bench  # noqa: E402
from oracle import oracle as O  # noqa: E402


class OracleIndex:
    """FlatIndex's call surface as bench.py uses it, computed by the numpy oracle."""

    def __init__(self, dim, metric=0, capacity_hint=0, device=None):
        self.dim, self.x = dim, np.zeros((0, dim), np.float32)

    def add(self, v):
        first = self.x.shape[0]
        self.x = np.concatenate([self.x, np.asarray(torch.as_tensor(v).cpu().numpy(), np.float32)])
        return first

    def search(self, q, k, row_base=0, stream=None):
        s, r = O.flat_search(torch.as_tensor(q).cpu().numpy().reshape(-1, self.dim), self.x, k)
        return torch.from_numpy(s.astype(np.float32)), torch.from_numpy(np.where(r >= 0, r + row_base, -1))

    def set_screening(self, on=True):
        pass

    def set_timing(self, on=True):
        pass

    def last_scan_ms(self):
        return 1.0

    def last_screened(self):
        return 0

    def last_geometry(self):
        return {"grid": 0, "block": 0, "lds_bytes": 0, "launches": 1}

    def close(self):
        pass


def merge(ps, pr, smaller_better=False):
    s, r = O.merge_topk(ps.numpy(), pr.numpy(), ps.shape[2])
    return torch.from_numpy(s.astype(np.float32)), torch.from_numpy(r)


if __name__ == "__main__":
    bench.main(sys.argv[1:], hooks={"device": "cpu", "backend": "gloo", "index_cls": OracleIndex, "merge": merge,
                                    "script": os.path.abspath(__file__)})

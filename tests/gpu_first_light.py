"""First-light GPU check (run by hand through gpurun): HIP scan vs numpy oracle on a few shapes."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from ragmeup_amd import FlatIndex
from oracle import oracle as O

def check(n, nq, k, d=384):
    x = O.make_corpus(n, d); q, perm = O.make_queries(x, nq)
    idx = FlatIndex(d)
    idx.add(x)
    idx.set_timing(True)
    t = time.time(); s, r = idx.search(q, k); dt = time.time() - t
    os_, or_ = O.flat_search(q, x, k)
    same = (r == or_)
    # tie rule: mismatching positions must be near-ties in fp64
    bad = 0
    for qi, pi in zip(*np.nonzero(~same)):
        if abs(os_[qi, pi] - s[qi, pi]) > 1e-6: bad += 1
    print(f"n={n} nq={nq} k={k}: id-match {same.mean():.6f} hard-mismatch {bad} max|ds| {np.abs(s-os_).max():.2e} "
          f"scan {idx.last_scan_ms():.3f} ms search {idx.last_search_ms():.3f} ms wall {dt*1e3:.1f} ms geom {idx.last_geometry()}", flush=True)
    return bad == 0

ok = True
for n, nq, k in [(1000, 1, 10), (5000, 7, 10), (20000, 33, 10), (20000, 64, 20), (50000, 200, 10), (50000, 130, 100), (33, 5, 10), (7, 3, 10), (100000, 1024, 10)]:
    ok &= check(n, nq, k)
print("ALL OK" if ok else "FAILED")

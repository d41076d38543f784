"""Writes tests/golden/search_golden.npz: seeded synthetic inputs (regenerated, not stored) and the fp64
oracle's top-k ids/scores.  Not reference-pinned (the reference has no goldens for this path)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

n, d, nq, k, seed_x, seed_q = 30000, 384, 48, 20, 1234, 4321
x = O.make_corpus(n, d, seed_x)
q, planted = O.make_queries(x, nq, seed_q)
s, r = O.flat_search(q, x, k)
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "search_golden.npz"),
                    n=n, d=d, nq=nq, k=k, seed_x=seed_x, seed_q=seed_q, rows=r, scores=s, planted=planted)
print("ok", r.shape)

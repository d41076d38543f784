"""Runs the screening search with ONE form of the screening kernel (selected by the environment, read once per process by librmu)
on a fixed synthetic corpus and prints a digest of the answers next to the exact fp32 scan's.  Executed by tests/test_search_gpu.py in a
fresh interpreter per form: the switchable forms of the product library must all return the exact scan's answers, bit for bit."""
import hashlib
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ragmeup_amd import FlatIndex  # noqa: E402

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(77)
n = 400_003                                        # ragged last tile; several ladder ranges
x = torch.randn((n, 384), generator=g, device=dev, dtype=torch.float32)
x /= x.norm(dim=1, keepdim=True)
idx = FlatIndex(384, capacity_hint=n, device=0)
idx.add(x)
idx.set_screen_min_batch(1)                         # every batch size goes through the screening path on this small corpus
out = {}
for nq in (1024, 300, 96, 7):                      # full query tiles (8 waves), a ragged one, one query tile (4 waves), a lone wave
    pick = torch.randperm(n, generator=g, device=dev)[:nq]
    q = x[pick] + 0.1 * torch.randn((nq, 384), generator=g, device=dev, dtype=torch.float32)
    q /= q.norm(dim=1, keepdim=True)
    idx.set_screening(True)
    s, r = idx.search(q, 10)
    screened = idx.last_screened()
    idx.set_screening(False)
    se, re_ = idx.search(q, 10)
    out[str(nq)] = {"screened": int(screened), "same": bool(torch.equal(s, se) and torch.equal(r, re_)),
                    "digest": hashlib.sha1(r.cpu().numpy().tobytes() + s.cpu().numpy().tobytes()).hexdigest()}
print("RESULT " + json.dumps(out))

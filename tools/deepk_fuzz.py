import os, sys, random
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch
from ragmeup_amd import FlatIndex, _native as N
rnd = random.Random(5)
dev = torch.device("cuda", 0)
bad = 0
for trial in range(18):
    n = rnd.choice([262_144, 300_001, 700_003, 1_000_000])
    nq = rnd.choice([1, 31, 64, 129, 500, 1024])
    k = rnd.choice([33, 41, 50, 64, 77, 100, 104])
    metric = rnd.choice([N.METRIC_IP, N.METRIC_COSINE, N.METRIC_L2SQ])
    g = torch.Generator(device=dev); g.manual_seed(trial)
    x = torch.randn((n, 384), generator=g, device=dev)
    if metric != N.METRIC_L2SQ or rnd.random() < 0.5:
        x /= x.norm(dim=1, keepdim=True)
    if rnd.random() < 0.5:                      # clustered near-duplicates: crowded candidate slots
        base = rnd.randrange(n - 300)
        x[base:base + 200] = x[base:base + 1] + 1e-3 * torch.randn((200, 384), generator=g, device=dev)
    q = x[torch.randint(0, n, (nq,), generator=g, device=dev)] + 0.05 * torch.randn((nq, 384), generator=g, device=dev)
    idx = FlatIndex(384, metric, capacity_hint=n, device=0)
    idx.add(x)
    if rnd.random() < 0.4:
        idx.remove_rows(np.arange(0, n, 7))
    s, r = idx.search(q, k)
    scr = idx.last_screened()
    idx.set_screening(False)
    s2, r2 = idx.search(q, k)
    same = bool(torch.equal(r, r2) and torch.equal(s, s2))
    print(f"trial {trial}: n {n} nq {nq} k {k} metric {metric} screened {scr} identical {same}", flush=True)
    bad += 0 if same else 1
    idx.close(); del x
print("FAILED" if bad else "ALL IDENTICAL")

"""Threshold-ladder geometry sweep in ONE process (the geometry is a per-index option: RMU_OPT_LADDER_RATIO / RMU_OPT_LADDER_FIRST):
step time, scan-kernel time and launches per search for a list of (ratio, first range) pairs, per batch size, and bit-identity of the
answers with the default geometry's.   python tools/ladder_sweep.py [--rows 10000000] [--batches 1,32,128] [--cfgs 0:0,8:16384,16:16384,...]"""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_shard
from ragmeup_amd import FlatIndex

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=10_000_000)
ap.add_argument("--batches", default="1,32,128")
ap.add_argument("--cfgs", default="0:0,8:2048,8:16384,16:16384,16:65536,32:16384,32:65536,64:131072")
ap.add_argument("--steps", type=int, default=30)
ap.add_argument("--k", type=int, default=10)
a = ap.parse_args()
dev = torch.device("cuda", 0)
idx = FlatIndex(384, capacity_hint=a.rows, device=0)
x = make_shard(a.rows, 384, 1234, dev)
idx.add(x)
g = torch.Generator(device=dev); g.manual_seed(4321)
bmax = max(int(b) for b in a.batches.split(","))
pick = torch.randperm(a.rows, generator=g, device=dev)[:bmax]
qall = x[pick] + 0.1 * torch.randn((bmax, 384), generator=g, dtype=torch.float32, device=dev)
qall /= qall.norm(dim=1, keepdim=True)
del x
for b in [int(v) for v in a.batches.split(",")]:
    q = qall[:b].contiguous()
    out = (torch.empty((b, a.k), dtype=torch.float32, device=dev), torch.empty((b, a.k), dtype=torch.int64, device=dev))
    ref = None
    for cfg in a.cfgs.split(","):
        ratio, first = (int(v) for v in cfg.split(":"))
        idx.set_ladder(ratio, first)
        for _ in range(3):
            s, r = idx.search(q, a.k, out=out)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            s, r = idx.search(q, a.k, out=out)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / a.steps
        idx.set_timing(True)
        ks = []
        for _ in range(5):
            idx.search(q, a.k); ks.append(idx.last_scan_ms())
        geo = idx.last_geometry()
        idx.set_timing(False)
        if ref is None:
            ref = (s.clone(), r.clone())
        same = bool(torch.equal(s, ref[0]) and torch.equal(r, ref[1]))
        print(f"rows {a.rows} batch {b:5d} ratio {ratio:3d} first {first:7d}: step {ms:.4f} ms  scan kernels {sum(ks)/len(ks):.4f} ms  launches {geo.get('launches')}  "
              f"{b / ms * 1e3:.0f} qps  hbm_frac_step {a.rows * 768 / (ms * 1e-3) / 8e12:.4f}  identical={same} screened={idx.last_screened()}", flush=True)
    idx.set_ladder(0, 0)

"""Native L2 index against the inner-product index on the same rows, one process: step time and scan-kernel time per batch size with the
screening path and with the exact scan, and the identity of the screened L2 answers with the exact L2 scan's.
python tools/l2_ab.py [--rows 2000000] [--batches 1024,128,32] [--k 10]"""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_shard
from ragmeup_amd import FlatIndex
from ragmeup_amd import _native as N

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=2_000_000)
ap.add_argument("--batches", default="1024,128,32")
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--k", type=int, default=10)
a = ap.parse_args()
dev = torch.device("cuda", 0)
x = make_shard(a.rows, 384, 1234, dev)
g = torch.Generator(device=dev); g.manual_seed(4321)
x *= 0.5 + torch.rand((a.rows, 1), generator=g, device=dev)             # norms 0.5 .. 1.5: the two metrics rank differently
bmax = max(int(b) for b in a.batches.split(","))
pick = torch.randperm(a.rows, generator=g, device=dev)[:bmax]
qall = x[pick] + 0.05 * torch.randn((bmax, 384), generator=g, dtype=torch.float32, device=dev)
for name, metric in (("ip", N.METRIC_IP), ("l2", N.METRIC_L2SQ)):
    idx = FlatIndex(384, metric=metric, capacity_hint=a.rows, device=0)
    for lo in range(0, a.rows, 1_000_000):
        idx.add(x[lo:lo + 1_000_000])
    for b in [int(v) for v in a.batches.split(",")]:
        q = qall[:b].contiguous()
        out = (torch.empty((b, a.k), dtype=torch.float32, device=dev), torch.empty((b, a.k), dtype=torch.int64, device=dev))
        res = {}
        for screen in (True, False):
            idx.set_screening(screen)
            for _ in range(3):
                s, r = idx.search(q, a.k, out=out)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.steps if screen else max(3, a.steps // 4)):
                s, r = idx.search(q, a.k, out=out)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) * 1e3 / (a.steps if screen else max(3, a.steps // 4))
            idx.set_timing(True)
            ks = []
            for _ in range(3):
                idx.search(q, a.k); ks.append(idx.last_scan_ms())
            idx.set_timing(False)
            res[screen] = (s.clone(), r.clone())
            print(f"{name} rows {a.rows} batch {b:5d} {'screened' if screen else 'exact   '}: step {ms:.4f} ms  scan kernels {sum(ks) / len(ks):.4f} ms  "
                  f"{b / ms * 1e3:.0f} qps  last_screened={idx.last_screened()}", flush=True)
        same = bool(torch.equal(res[True][0], res[False][0]) and torch.equal(res[True][1], res[False][1]))
        print(f"{name} batch {b}: screened == exact bit for bit: {same}; top-1 is the source row: {float((res[True][1][:, 0] == pick[:b]).float().mean()):.4f}", flush=True)
    idx.close()

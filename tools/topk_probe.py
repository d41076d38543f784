"""Dense top-k latency probe (ON THE GPU BOX): ms per search for (rows, queries, k) triples, default path.
python tools/topk_probe.py 1000000:64:100 1000000:64:10 10000000:64:100"""
import sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_shard, timed
from ragmeup_amd import FlatIndex, _native
cur = None
for spec in sys.argv[1:]:
    n, nq, k = (int(v) for v in spec.split(":"))
    if cur is None or cur[0] != n:
        if cur: cur[1].close()
        x = make_shard(n, 384, 1234, torch.device("cuda"))
        idx = FlatIndex(384, capacity_hint=n); idx.add(x)
        cur = (n, idx, x)
    q = cur[2][:nq] + 0.1 * torch.randn((nq, 384), device="cuda"); q /= q.norm(dim=1, keepdim=True); q = q.contiguous()
    ms = timed(lambda: cur[1].search(q, k), steps=10, warmup=3)
    cur[1].set_timing(True); cur[1].search(q, k); kms = cur[1].last_scan_ms(); cur[1].set_timing(False)
    print(f"PROBE rows={n} nq={nq} k={k}: {ms:.3f} ms/search (scan kernels {kms:.3f} ms), passes {cur[1].last_geometry()['launches']}, "
          f"{n*1536/ms/1e6:.0f} GB/s of fp32 rows")

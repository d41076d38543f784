"""Per-range debug counters of the screening ladder (debug-kernels build, RMU_SCAN_EXP=7): slow tiles, compactions, appends, cycles in the
slow path / at the pair barrier / in the ring wait, per launch of ONE search.
  RMU_TUNING=1 RMU_LIB=ragmeup_amd/lib/librmu_dbg.so RMU_SCAN_EXP=7 python tools/screen_dbg_counters.py [rows] [queries]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_shard
from ragmeup_amd import FlatIndex
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
b = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
dev = torch.device("cuda", 0)
idx = FlatIndex(384, capacity_hint=n, device=0)
x = make_shard(n, 384, 1234, dev); idx.add(x)
q = (x[:b] + 0.1 * torch.randn((b, 384), device=dev)); q /= q.norm(dim=1, keepdim=True)
out = (torch.empty((b, 10), dtype=torch.float32, device=dev), torch.empty((b, 10), dtype=torch.int64, device=dev))
for i in range(2):
    print(f"---- search {i}", file=sys.stderr, flush=True)
    idx.search(q, 10, out=out)
    torch.cuda.synchronize()

"""Sibling-pacing probe for the headline screening scan (10M x 384, batch 1024): step time, scan-kernel time and bit-identity of
the answers for a list of pacing windows (RMU_SCREEN_PACE, 0 = off), one index build.  Under rocprofv3 give ONE window and few steps.
  python tools/pace_probe.py [--pace 0,4,8,16,32] [--steps 20] [--rows 10000000] [--batch 1024]"""
import argparse, os, sys, time
os.environ.setdefault("RMU_TUNING", "1")     # librmu honours its RMU_* switches only with this set
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_shard
from ragmeup_amd import FlatIndex

ap = argparse.ArgumentParser()
ap.add_argument("--pace", default="0,1,0,1")
ap.add_argument("--env", default="RMU_SCREEN_PP", help="switch that is read per launch (RMU_SCREEN_PP while the ping-pong form is measured)")
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--rows", type=int, default=10_000_000)
ap.add_argument("--batch", type=int, default=1024)
ap.add_argument("--save", default="", help="write the answers (scores, rows) of the last window here")
ap.add_argument("--ref", default="", help="compare every window's answers with the ones saved here (another process / another kernel)")
a = ap.parse_args()
dev = torch.device("cuda", 0)
idx = FlatIndex(384, capacity_hint=a.rows, device=0)
x = make_shard(a.rows, 384, 1234, dev)
idx.add(x)
g = torch.Generator(device=dev); g.manual_seed(4321)
pick = torch.randperm(a.rows, generator=g, device=dev)[:a.batch]
q = x[pick] + 0.1 * torch.randn((a.batch, 384), generator=g, dtype=torch.float32, device=dev)
q /= q.norm(dim=1, keepdim=True)
del x
ref = None
if a.ref:
    ref = tuple(t.to(dev) for t in torch.load(a.ref, weights_only=True))
for w in [int(v) for v in a.pace.split(",")]:
    os.environ[a.env] = str(w)   # NOTE: read once per process since the tuning ended: run one window per process
    for _ in range(3):
        s, r = idx.search(q, 10)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        s, r = idx.search(q, 10)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / a.steps
    idx.set_timing(True)
    k = []
    for _ in range(min(a.steps, 5)):
        idx.search(q, 10); k.append(idx.last_scan_ms())
    idx.set_timing(False)
    if ref is None:
        ref = (s.clone(), r.clone())
    same = bool(torch.equal(s, ref[0]) and torch.equal(r, ref[1]))
    if a.save:
        torch.save((s.cpu(), r.cpu()), a.save)
    print(f"{a.env}={w:3d}: step {ms:.3f} ms  scan kernels {sum(k)/len(k):.3f} ms  {a.batch/ms*1e3:.0f} qps  identical_to_first={same} screened={idx.last_screened()}", flush=True)

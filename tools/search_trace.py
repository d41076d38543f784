"""One small-batch search's timeline on the GPU box (the north-star regime: batch 16 / 32 over 10M rows): run under `rocprofv3 --kernel-trace`
and print the kernels of the LAST search with their durations and the gaps between them.
  rocprofv3 --kernel-trace --output-format csv -d /tmp/st -o s -- python tools/search_trace.py run 32 ; python tools/search_trace.py show /tmp/st"""
import glob, os, sys
if sys.argv[1] == "run":
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import make_shard
    from ragmeup_amd import FlatIndex
    b = int(sys.argv[2]); n = int(sys.argv[3]) if len(sys.argv) > 3 else 10_000_000
    dev = torch.device("cuda", 0)
    idx = FlatIndex(384, capacity_hint=n, device=0)
    x = make_shard(n, 384, 1234, dev); idx.add(x)
    q = (x[:b] + 0.1 * torch.randn((b, 384), device=dev)); q /= q.norm(dim=1, keepdim=True)
    out = (torch.empty((b, 10), dtype=torch.float32, device=dev), torch.empty((b, 10), dtype=torch.int64, device=dev))
    for _ in range(12):
        idx.search(q, 10, out=out)
    torch.cuda.synchronize()
else:
    import csv
    f = glob.glob(sys.argv[2] + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    last = max(i for i, r in enumerate(rows) if "k_split_rows" in r["Kernel_Name"])     # the query conversion opens a search
    seq = rows[last:]
    t0 = int(seq[0]["Start_Timestamp"]); prev_end = t0
    tot_k = 0
    for r in seq:
        st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:60]
        print(f"{(st - t0) / 1e3:8.1f} us  gap {(st - prev_end) / 1e3:5.1f}  dur {(en - st) / 1e3:7.1f}  grid {r.get('Grid_Size', r.get('Grid_Size_X', '?')):>8}  {name}")
        prev_end = en; tot_k += en - st
    print(f"kernels {len(seq)}  span {(prev_end - t0) / 1e3:.1f} us  sum of durations {tot_k / 1e3:.1f} us")

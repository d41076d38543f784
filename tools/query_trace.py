"""One-query timeline on the GPU box: run under `rocprofv3 --kernel-trace` and print the kernels of the last fused query call
(rmu_bert_search_mmr: graph-replayed batch-1 forward + search + MMR) with their durations and the gaps between them.
  rocprofv3 --kernel-trace --output-format csv -d /tmp/qt -o q -- python tools/query_trace.py run ; python tools/query_trace.py show /tmp/qt"""
import glob, os, sys
if sys.argv[1] == "run":
    import numpy as np, torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import bert_weights, synth_tokens
    from ragmeup_amd import FlatIndex
    from ragmeup_amd.bert import BertEncoder
    enc = BertEncoder(bert_weights(0, False), layers=6)
    x = torch.nn.functional.normalize(torch.randn((10000, 384), device="cuda"), dim=1)
    idx = FlatIndex(384, capacity_hint=10000); idx.add(x)
    qids, _, qlens = synth_tokens(1, seed=22, lmin=16, lmax=16, mean=16, std=1)
    for _ in range(20):
        enc.search_host(idx, qids, qlens, 0, 20, 10, 0.5)
    torch.cuda.synchronize()
else:
    import csv
    f = glob.glob(sys.argv[2] + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    # the last call = everything from the last k_cu_seqlens on
    last = max(i for i, r in enumerate(rows) if "k_cu_seqlens" in r["Kernel_Name"])
    seq = rows[last:]
    t0 = int(seq[0]["Start_Timestamp"]); prev_end = t0
    tot_k = 0
    for r in seq:
        st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:44]
        print(f"{(st - t0) / 1e3:8.1f} us  gap {(st - prev_end) / 1e3:5.1f}  dur {(en - st) / 1e3:5.1f}  grid {r.get('Grid_Size', r.get('Grid_Size_X', '?')):>8}  {name}")
        prev_end = en; tot_k += en - st
    print(f"kernels {len(seq)}  span {(prev_end - t0) / 1e3:.1f} us  sum of durations {tot_k / 1e3:.1f} us")

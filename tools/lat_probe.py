"""Single-query latency split on the GPU box: encoder forward at batch 1, dense top-10 over a 10k / 1M-row index, both."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import bert_weights, synth_tokens, timed
from ragmeup_amd import FlatIndex
from ragmeup_amd.bert import BertEncoder
enc = BertEncoder(bert_weights(0, False), layers=6)
qids, _, qlens = synth_tokens(1, seed=22, lmin=16, lmax=16, mean=16, std=1)
qi, ql = torch.as_tensor(qids).cuda(), torch.as_tensor(qlens).cuda()
out = torch.empty((1, 384), dtype=torch.float32, device="cuda")
for n in (10_000, 1_000_000):
    x = torch.nn.functional.normalize(torch.randn((n, 384), device="cuda"), dim=1)
    idx = FlatIndex(384, capacity_hint=n); idx.add(x)
    v = enc.encode_ids(qi, ql, None, 0)
    e = timed(lambda: enc.encode_ids(qi, ql, None, 0, out=out), 200, 20)
    s = timed(lambda: idx.search(v, 10), 200, 20)
    b = timed(lambda: idx.search(enc.encode_ids(qi, ql, None, 0, out=out), 10), 200, 20)
    g = timed(lambda: enc.encode_host(qids, qlens, None, 0), 200, 20)                      # host ids -> host vector, one hipGraph replay
    gb = timed(lambda: idx.search(enc.encode_host(qids, qlens, None, 0), 10), 200, 20)
    print(f"rows {n}: encode(batch 1, 16 tokens) {e:.3f} ms   search(top-10) {s:.3f} ms   both {b:.3f} ms   |   "
          f"graph-replayed host encode {g:.3f} ms   + search {gb:.3f} ms")
    idx.close()
ids, tt, lens = synth_tokens(30, seed=9, lmin=100, lmax=190, mean=147, std=20, pair=True)
ce = BertEncoder(bert_weights(1, True), layers=6)
a = [torch.as_tensor(t).cuda() for t in (ids, lens, tt)]
print(f"cross-encoder 30 pairs: {timed(lambda: ce.encode_ids(a[0], a[1], a[2], mode=1), 100, 10):.3f} ms")

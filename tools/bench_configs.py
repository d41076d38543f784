#!/usr/bin/env python
"""Secondary benchmarks (BASELINE.json configs 3 and 5); bench.py stays the driver's headline benchmark.

  python tools/bench_configs.py embed  [--chunks 65536] [--batch 8192]
  python tools/bench_configs.py rerank [--rows 1000000] [--queries 64]
Synthetic token ids, architecture-exact random-init weights (no checkpoints exist offline).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.helpers import bert_weights_numpy, make_bert, synth_tokens  # noqa: E402  (weights/ids generators only)


def flops_per_token(L):
    return 2 * 6 * (4 * 384 * 384 + 2 * 384 * 1536) + 6 * 4 * L * 384


def bench_embed(args):
    from ragmeup_amd.bert import BertEncoder
    enc = BertEncoder(bert_weights_numpy(make_bert(0, 6)), layers=6)
    ids, _, lens = synth_tokens(args.batch, seed=7)
    ids_t = torch.as_tensor(ids).cuda(); lens_t = torch.as_tensor(lens).cuda()
    out = torch.empty((args.batch, 384), dtype=torch.float32, device="cuda")
    for _ in range(2):
        enc.encode_ids(ids_t, lens_t, None, 0, out=out)
    torch.cuda.synchronize()
    reps = max(1, args.chunks // args.batch)
    t0 = time.perf_counter()
    for _ in range(reps):
        enc.encode_ids(ids_t, lens_t, None, 0, out=out)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tokens = int(lens.sum()) * reps
    fl = sum(flops_per_token(int(l)) * int(l) for l in lens) * reps
    print(json.dumps({"config": "embed", "chunks": reps * args.batch, "chunks_per_s": round(reps * args.batch / dt, 1),
                      "tokens_per_s": round(tokens / dt, 1), "tflops": round(fl / dt / 1e12, 2),
                      "mean_len": float(lens.mean()), "batch": args.batch, "seconds": round(dt, 3)}))


def bench_rerank(args):
    from ragmeup_amd import FlatIndex
    from ragmeup_amd.bert import BertEncoder
    ce = BertEncoder(bert_weights_numpy(make_bert(1, 6, head=True)), layers=6)
    g = torch.Generator(device="cuda").manual_seed(1234)
    x = torch.randn((args.rows, 384), generator=g, device="cuda"); x /= x.norm(dim=1, keepdim=True)
    idx = FlatIndex(384, capacity_hint=args.rows); idx.add(x)
    q = x[torch.randperm(args.rows, generator=g, device="cuda")[:args.queries]].clone()
    # synthetic pairs: 16 query tokens + ~128 passage tokens; passage tokens keyed by the row id
    ids, tt, lens = synth_tokens(args.queries * 100, seed=9, lmin=100, lmax=190, mean=147, std=20, pair=True)
    ids_t, tt_t, lens_t = (torch.as_tensor(a).cuda() for a in (ids, tt, lens))
    def step():
        s, r = idx.search(q, 100)                                   # dense top-100
        logits = ce.encode_ids(ids_t, lens_t, tt_t, mode=1)         # 100 pairs per query
        top = torch.topk(logits.view(args.queries, 100), 10, dim=1) # final top-10 (stable order applied on host side)
        return r.gather(1, top.indices)
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"config": "retrieve100_rerank10", "rows": args.rows, "queries_per_step": args.queries,
                      "qps": round(args.queries * args.steps / dt, 1), "pairs_per_s": round(args.queries * 100 * args.steps / dt, 1),
                      "seconds": round(dt, 3)}))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("what", choices=["embed", "rerank"])
    ap.add_argument("--chunks", type=int, default=65536)
    ap.add_argument("--batch", type=int, default=8192)
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--queries", type=int, default=64)
    ap.add_argument("--steps", type=int, default=5)
    a = ap.parse_args()
    bench_embed(a) if a.what == "embed" else bench_rerank(a)

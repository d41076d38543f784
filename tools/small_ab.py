"""A/B of the interactive path's launch fusions on the GPU box (round 5): per-call latency of (1) a graph-replayed 16-token query forward,
(2) the same + dense top-10 over a 10k-row index in one call (rmu_bert_search_mmr: bench.py's C1 leg), (3) a 48-token query, (4) the
reference's rerank call (14 pairs, ~1.5k tokens) -- and digests of every output, so that runs under different switches (read once per
process: RMU_QKV_ATTN_TOKENS, RMU_SMALL_FUSE; set RMU_TUNING=1) can be compared bit for bit:  python tools/small_ab.py [tag]"""
import hashlib, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import bert_weights, spread_embeddings, synth_tokens, timed, topic_tokens
from ragmeup_amd import FlatIndex
from ragmeup_amd.bert import BertEncoder

tag = sys.argv[1] if len(sys.argv) > 1 else "run"
dg = lambda a: hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()[:12]
enc = BertEncoder(spread_embeddings(bert_weights(0, False)), layers=6)
ids, lens, qids, qlens, _ = topic_tokens(10_000, 8, seed=21)
idx = FlatIndex(384, capacity_hint=10_000)
idx.add(enc.encode_ids(torch.as_tensor(ids).cuda(), torch.as_tensor(lens).cuda(), None, 0))
q1, l1 = np.ascontiguousarray(qids[:1]), qlens[:1].copy()
v = enc.encode_host(q1, l1, None, 0)
rows, sc = enc.search_host(idx, q1, l1, 0, 10, 10, None)
rows_m, sc_m = enc.search_host(idx, q1, l1, 0, 20, 10, 0.5)
q48, _, l48 = synth_tokens(1, seed=5, lmin=48, lmax=48, mean=48, std=1)
v48 = enc.encode_host(q48, l48, None, 0)
b4, _, l4 = synth_tokens(4, seed=6, lmin=20, lmax=120, mean=70, std=30)
v4 = enc.encode_host(b4, l4, None, 0)
dev = enc.encode_ids(b4, l4, None, 0).cpu().numpy()
t_q = timed(lambda: enc.encode_host(q1, l1, None, 0), 300, 30)
t_s = timed(lambda: enc.search_host(idx, q1, l1, 0, 10, 10, None), 300, 30)
t_m = timed(lambda: enc.search_host(idx, q1, l1, 0, 20, 10, 0.5), 300, 30)
t_48 = timed(lambda: enc.encode_host(q48, l48, None, 0), 300, 30)
ce = BertEncoder(bert_weights(1, True), layers=6)
pi, pt, pl = synth_tokens(14, seed=9, lmin=60, lmax=160, mean=110, std=20, pair=True)
lg = ce.encode_host(pi, pl, pt, 1)
t_r = timed(lambda: ce.encode_host(pi, pl, pt, 1), 200, 20)
print(f"AB {tag}: query16 {t_q:.4f} ms | query16+top10 {t_s:.4f} ms | query16+top20+mmr {t_m:.4f} ms | query48 {t_48:.4f} ms | rerank14 {t_r:.4f} ms || digests "
      f"q16 {dg(v)} rows {dg(rows)} sc {dg(sc)} mmr {dg(rows_m)} q48 {dg(v48)} b4 {dg(v4)} b4==device {bool(np.array_equal(v4, dev))} ce {dg(lg)}", flush=True)

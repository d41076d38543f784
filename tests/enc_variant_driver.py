"""Runs ONE encoder kernel variant (selected by the environment, read once per process by librmu) on a fixed synthetic batch
and prints its parity against the fp64 oracle as JSON.  Executed by tests/test_encoder_gpu.py in a fresh interpreter."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O  # noqa: E402
from tests.helpers import bert_weights_numpy, centred_cosine, make_bert, synth_tokens, token_rel_error  # noqa: E402
from ragmeup_amd.bert import BertEncoder  # noqa: E402

m = make_bert(seed=0, layers=6)
w = bert_weights_numpy(m)
enc = BertEncoder(w, layers=6)
# 300 sequences ~ 38k tokens: more than one 256-token tile per workgroup of the persistent GEMM, ragged tails everywhere
ids, tt, lens = synth_tokens(300, seed=11, lmin=3, lmax=250, mean=128, std=60)
got = enc.encode_ids(ids, lens, None, mode=0).cpu().numpy()
sel = np.arange(0, 300, 10)                      # the oracle is fp64 numpy: compare a sample of 30 sequences
hid = O.bert_hidden(w, ids[sel], np.zeros_like(ids[sel]), lens[sel])
ref = O.embed_pool(hid, lens[sel])
tok = enc.encode_ids(ids, lens, None, mode=3).cpu().numpy()          # RMU_BERT_TOKENS: every token's final hidden state
cu = np.concatenate([[0], np.cumsum(lens)])
rel = token_rel_error(np.concatenate([tok[cu[b]:cu[b + 1]] for b in sel]), hid, lens[sel])
g = got[sel]
cos = (g * ref).sum(1) / np.linalg.norm(g, axis=1) / np.linalg.norm(ref, axis=1)
# the interactive sizes (k_gemm_small with the LayerNorm folds; round 5: k_qkv_attn_small, offsets inside k_embed_ln, k_pool_ln): one short
# query, a few sequences, a long one -- every token's final hidden state and the pooled vectors against the oracle
small_rel, small_cos = 0.0, 1.0
for n, lmax, mean in ((1, 16, 12), (3, 40, 24), (2, 200, 150), (5, 64, 40)):
    sids, _, slens = synth_tokens(n, seed=40 + n, lmin=2, lmax=lmax, mean=mean, std=max(2, mean // 3))
    shid = O.bert_hidden(w, sids, np.zeros_like(sids), slens)
    stok = enc.encode_ids(sids, slens, None, mode=3).cpu().numpy()
    small_rel = max(small_rel, float(token_rel_error(stok, shid, slens).max()))
    sp, sr = enc.encode_ids(sids, slens, None, mode=0).cpu().numpy(), O.embed_pool(shid, slens)
    small_cos = min(small_cos, float(((sp * sr).sum(1) / np.linalg.norm(sp, axis=1) / np.linalg.norm(sr, axis=1)).min()))
    assert np.array_equal(enc.encode_host(sids, slens, None, 0), sp)          # the graph-replayed host entry point takes the same kernels
import hashlib  # noqa: E402
print("RESULT " + json.dumps({"tok_sha": hashlib.sha256(np.ascontiguousarray(tok).tobytes()).hexdigest(), "pooled_sha": hashlib.sha256(np.ascontiguousarray(got).tobytes()).hexdigest(), "small_max_tok_rel": small_rel, "small_min_cos": small_cos, "min_cos": float(cos.min()), "finite": bool(np.isfinite(got).all()),
                              "max_tok_rel": float(rel.max()), "mean_tok_rel": float(rel.mean()),
                              "min_centred_cos": float(centred_cosine(g, ref).min()),
                              "norm_err": float(np.abs(np.linalg.norm(got, axis=1) - 1).max())}))

"""Latency of the reference's rerank call on the GPU box: the cross-encoder forward over a handful of (query, passage) pairs
(<= 14 per ScoredCrossEncoderReranker.compress_documents call), device-resident ids, for a list of RMU_MID_TOKENS thresholds
(tokens up to which the GEMMs take k_gemm_small; 256 = the round-3 behaviour).  python tools/ce_probe.py [256,4096] [14,30,100]"""
import os as _os
_os.environ.setdefault("RMU_TUNING", "1")
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import bert_weights, synth_tokens, timed
from ragmeup_amd.bert import BertEncoder
mids = [v for v in (sys.argv[1] if len(sys.argv) > 1 else "256,4096").split(",")]      # "MID" or "MID:TB"
pairs = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "14,30,100").split(",")]
ce = BertEncoder(bert_weights(1, True), layers=6)
ref = {}
for n in pairs:
    ids, tt, lens = synth_tokens(n, seed=9, lmin=60, lmax=160, mean=110, std=20, pair=True)
    a = [torch.as_tensor(t).cuda() for t in (ids, lens, tt)]
    for mid in mids:
        # "FOLD[:TB[:CFG[:G3MIN]]]": FOLD = RMU_FOLD_TOKENS (tokens up to which the LN-folded k_gemm_small forward runs; 0 = the tiled kernels);
        # the other three are debug-build switches
        f = mid.split(":")
        os.environ["RMU_FOLD_TOKENS"] = f[0]
        os.environ["RMU_MID_TOKENS"] = "256"
        os.environ["RMU_SMALL_TB"] = f[1] if len(f) > 1 else "128"   # (debug builds only)
        os.environ["RMU_GEMM_CFG"] = f[2] if len(f) > 2 else "0"
        os.environ["RMU_G3_MIN"] = f[3] if len(f) > 3 else "0"
        out = ce.encode_ids(a[0], a[1], a[2], mode=1)
        ms = timed(lambda: ce.encode_ids(a[0], a[1], a[2], mode=1), 100, 10)
        o = out.cpu().numpy()
        d = float(np.abs(o - ref.setdefault(n, o)).max())
        print(f"pairs {n:4d} tokens {int(lens.sum()):6d} cap {ids.size:6d} FOLD {mid:>8s}: {ms:.3f} ms per call   max|dlogit| vs first {d:.2e}", flush=True)

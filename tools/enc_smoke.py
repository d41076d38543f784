"""Tiny encoder run for bisecting kernel variants on the GPU box: encodes n synthetic sequences and prints a checksum."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import bert_weights, synth_tokens
from ragmeup_amd.bert import BertEncoder
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
enc = BertEncoder(bert_weights(0, False), layers=6)
ids, _, lens = synth_tokens(n, seed=7)
out = enc.encode_ids(ids, lens, None, mode=0)
torch.cuda.synchronize()
o = out.cpu().numpy()
print("OK", n, "finite", bool(np.isfinite(o).all()), "norm", float(np.linalg.norm(o, axis=1).mean()), "checksum", float(o[:, :7].sum()))

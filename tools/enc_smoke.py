"""Tiny encoder run for bisecting kernel variants on the GPU box: encodes n synthetic sequences, prints a checksum and the
encode rate; ENC_OUT=<path.npy> saves the embeddings, ENC_REF=<path.npy> compares with a saved run (the kernel-variant
switches are read once per process, so variants are compared across processes)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import bert_weights, synth_tokens
from ragmeup_amd.bert import BertEncoder
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 0
pair = len(sys.argv) > 3 and sys.argv[3] == "pair"      # the rerank workload of bench.py's C5 leg: (query, passage) pairs through the cross-encoder
if pair:
    enc = BertEncoder(bert_weights(1, True), layers=6)
    ids, tt, lens = synth_tokens(n, seed=9, lmin=100, lmax=190, mean=147, std=20, pair=True)
    ids, tt, lens = (torch.as_tensor(a).cuda() for a in (ids, tt, lens))
    out = enc.encode_ids(ids, lens, tt, mode=1)
    torch.cuda.synchronize()
    o = out.cpu().numpy()
    print("OK", n, "pairs finite", bool(np.isfinite(o).all()), "checksum", float(o.sum()))
    sys.exit(0)
enc = BertEncoder(bert_weights(0, False), layers=6)
ids, _, lens = synth_tokens(n, seed=7)
if os.environ.get("ENC_DEVICE_IDS"):                     # ids resident on the device (what bench.py's embed leg times)
    ids, lens = torch.as_tensor(ids).cuda(), torch.as_tensor(lens).cuda()
out = enc.encode_ids(ids, lens, None, mode=0)
torch.cuda.synchronize()
o = out.cpu().numpy()
print("OK", n, "finite", bool(np.isfinite(o).all()), "norm", float(np.linalg.norm(o, axis=1).mean()), "checksum", float(o[:, :7].sum()))
if os.environ.get("ENC_OUT"):
    np.save(os.environ["ENC_OUT"], o)
if os.environ.get("ENC_REF"):
    r = np.load(os.environ["ENC_REF"])
    cos = (o * r).sum(1) / (np.linalg.norm(o, axis=1) * np.linalg.norm(r, axis=1))
    print("vs ref: min cosine", float(cos.min()), "max |diff|", float(np.abs(o - r).max()))
if reps:
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        enc.encode_ids(ids, lens, None, mode=0)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print("RATE %s %.1f chunks/s  %.3f ms" % (os.environ.get("TAG", ""), n / dt, dt * 1e3))

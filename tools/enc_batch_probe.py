"""Encoder throughput vs forward size on the GPU box (device-resident ids): does a forward whose activations fit the 256-MiB
Infinity Cache run faster per chunk than the 8192-chunk forwards?  python tools/enc_batch_probe.py [256,512,1024,2048,4096,8192]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import bert_weights, synth_tokens, encoder_flops
from ragmeup_amd.bert import BertEncoder
sizes = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "256,512,1024,2048,4096,8192").split(",")]
enc = BertEncoder(bert_weights(0, False), layers=6)
ids, _, lens = synth_tokens(8192, seed=7)
ids_t, lens_t = torch.as_tensor(ids).cuda(), torch.as_tensor(lens).cuda()
out = torch.empty((8192, 384), dtype=torch.float32, device="cuda")
fl = encoder_flops(lens)
for n in sizes:
    def run():
        for lo in range(0, 8192, n):
            enc.encode_ids(ids_t[lo:lo + n], lens_t[lo:lo + n], None, 0, out=out[lo:lo + n])
    run(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): run()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    print(f"forward size {n:5d} chunks: {8192 / ms * 1e3:9.0f} chunks/s  {ms:7.2f} ms per 8192 chunks  {fl / ms / 1e9:6.1f} TFLOP/s", flush=True)

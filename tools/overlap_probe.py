"""Do two encoder forwards on two streams overlap usefully?  The bulk forward's kernels are bound by different things (k_ffn3 /
k_gemm3: matrix pipe + LDS, one workgroup per CU; k_attn3: latency; out-proj k_gemm: HBM), and a single stream runs them one after
the other.  Two models (same weights, own workspaces) x half the chunks each, enqueued on two streams with a phase offset, against one
model x all chunks on one stream.  python tools/overlap_probe.py [offsets in us, comma separated]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import bert_weights, synth_tokens
from ragmeup_amd.bert import BertEncoder

offsets = [float(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "0,300,700,1200,2000,3000").split(",")]
NCH = 8192
REP = 6
w = bert_weights(0, False)
encs = [BertEncoder(w, layers=6) for _ in range(2)]
ids, _, lens = synth_tokens(NCH, seed=7)
ids_t, lens_t = torch.as_tensor(ids).cuda().int().contiguous(), torch.as_tensor(lens).cuda().int().contiguous()
out = torch.empty((NCH, 384), dtype=torch.float32, device="cuda")
ref = torch.empty((NCH, 384), dtype=torch.float32, device="cuda")
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
torch.cuda.synchronize()


def one_stream(n_per_call):
    s = streams[0]
    for _ in range(REP):
        for lo in range(0, NCH, n_per_call):
            encs[0].encode_ids(ids_t[lo:lo + n_per_call], lens_t[lo:lo + n_per_call], None, 0, out=ref[lo:lo + n_per_call], stream=s)
    s.synchronize()


def two_streams(off_us, n_per_call):
    # stream 1 starts `off_us` behind stream 0 (a spin kernel of that length in front of its first forward)
    if off_us > 0:
        with torch.cuda.stream(streams[1]):
            torch.cuda._sleep(int(off_us * 100))          # ~100 MHz timer ticks: 100 per us (approximate; the offset is what matters, not its value)
    half = NCH // 2
    for _ in range(REP):
        for lo in range(0, half, n_per_call):
            for r in range(2):
                a = r * half + lo
                encs[r].encode_ids(ids_t[a:a + n_per_call], lens_t[a:a + n_per_call], None, 0, out=out[a:a + n_per_call], stream=streams[r])
    for s in streams:
        s.synchronize()


def timed(fn, *a):
    fn(*a)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(*a)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / REP * 1e3


for n in (8192, 4096):
    ms = timed(one_stream, n)
    print(f"one stream, {n}-chunk calls: {ms:7.2f} ms per {NCH} chunks  {NCH / ms * 1e3:9.0f} chunks/s", flush=True)
for n in (4096, 2048):
    for off in offsets:
        ms = timed(two_streams, off, n)
        same = bool(torch.equal(out, ref))
        print(f"two streams, {n}-chunk calls, offset {off:6.0f} us: {ms:7.2f} ms per {NCH} chunks  {NCH / ms * 1e3:9.0f} chunks/s  identical {same}", flush=True)

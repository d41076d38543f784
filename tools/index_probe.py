"""Where the time of an add_documents call goes (GPU): tokenizer, encoder (one forward in input order vs length-sorted batches),
vector-store bookkeeping; per call size.  python tools/index_probe.py [n_texts]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as Bn                                                   # noqa: E402
from ragmeup_amd import bert as B                                    # noqa: E402
from ragmeup_amd.embeddings import MI355XEmbeddings                  # noqa: E402
from ragmeup_amd.tokenizer import WordPieceTokenizer                 # noqa: E402
from ragmeup_amd.vectorstore import MI355XVectorStore                # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
import tempfile                                                      # noqa: E402
vocab, words = Bn.synth_vocab_and_words()
texts = Bn.synth_texts(words, n, seed=5)
vp = os.path.join(tempfile.mkdtemp(), "vocab.txt")
open(vp, "w", encoding="utf-8").write("\n".join(vocab) + "\n")
tok = WordPieceTokenizer(vp)
enc = B.BertEncoder(Bn.bert_weights(0, False), layers=6)
emb = MI355XEmbeddings(encoder=enc, tokenizer=tok, max_seq_length=256)


def sync():
    torch.cuda.synchronize()


def timed(f, reps=1):
    sync(); t0 = time.perf_counter()
    for _ in range(reps):
        f()
    sync()
    return (time.perf_counter() - t0) / reps


for batch in (1000, 8192):
    blk = texts[:batch]
    emb.embed_documents_device(blk); sync()
    t_tok = timed(lambda: emb._tokenize(blk), 5)
    ids, lens = emb._tokenize(blk)
    out = torch.empty((batch, 384), dtype=torch.float32, device="cuda")
    res = {}
    for one in (True, False):
        emb.one_forward = one
        emb._run_arrays(ids, None, lens, emb._mode, out=out); sync()
        res[one] = timed(lambda: emb._run_arrays(ids, None, lens, emb._mode, out=out), 5)
    # pre-sorted ids through the one-forward path: is it the ORDER that matters to the kernels?
    order = np.argsort(-lens, kind="stable")
    ids_s, lens_s = np.ascontiguousarray(ids[order]), lens[order]
    emb.one_forward = True
    t_sorted = timed(lambda: emb._run_arrays(ids_s, None, lens_s, emb._mode, out=out), 5)
    print(f"block {batch}: tokenizer {t_tok * 1e3:.2f} ms | encoder one forward, input order {res[True] * 1e3:.2f} ms | "
          f"length-sorted batches {res[False] * 1e3:.2f} ms | one forward over pre-sorted rows {t_sorted * 1e3:.2f} ms", flush=True)
    for one in (True, False):
        emb.one_forward = one
        t_dev = timed(lambda: emb.embed_documents_device(blk), 5)
        print(f"   embed_documents_device(one_forward={one}) {t_dev * 1e3:.2f} ms")

ids_all = [f"id{i}" for i in range(n)]
for one in (True, False):
    emb.one_forward = one
    for batch in (1000, n):
        store = MI355XVectorStore(embeddings=emb, collection_name=f"probe{batch}{one}", auto_persist=False)
        store.add_texts(texts[:batch], ids=ids_all[:batch]); sync()                # warm
        store = MI355XVectorStore(embeddings=emb, collection_name=f"probe{batch}{one}b", auto_persist=False)
        sync(); t0 = time.perf_counter()
        for i in range(0, n, batch):
            store.add_texts(texts[i:i + batch], ids=ids_all[i:i + batch])
        sync(); dt = time.perf_counter() - t0
        print(f"add_texts one_forward={one} in {batch}-text calls: {n / dt:.0f} chunks/s ({dt * 1e3:.1f} ms)", flush=True)
        store._index.close()

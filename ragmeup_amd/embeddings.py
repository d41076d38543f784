"""MI355XEmbeddings / MI355XCrossEncoder -- the `Embeddings` and `BaseCrossEncoder` objects the reference
builds at server/RAGHelper_local.py:107-117 (HuggingFaceEmbeddings) and server/RAGHelper.py:483-486
(HuggingFaceCrossEncoder), with the forwards running in librmu.so.

Behaviour restated from the pinned dependencies:
  * embed_documents: "\\n" -> " ", tokenize (truncate to max_seq_length = 256), BERT, masked mean pool,
    L2 normalise, -> list[list[float]];  embed_query(t) = embed_documents([t])[0].
  * score(pairs): tokenize pair ([CLS] q [SEP] p [SEP], token types 0/1, longest-first truncation to 512),
    BertForSequenceClassification logit (num_labels = 1), -> list[float].
Sequences are length-sorted and batched by a token budget (sentence-transformers sorts by length too); the
packed-token kernels spend no FLOPs on padding.

A tokenizer needs a vocabulary, which does not exist offline in the build image: pass `tokenizer=` (any
callable HF-style tokenizer) or a `model_dir` containing tokenizer files.  `embed_ids` / `score_ids` take
pre-tokenised input and are what the benchmarks use.
"""
from __future__ import annotations

from typing import Any, Sequence

import numpy as np

from ._lc import CROSS_ENCODER_BASES, Embeddings
from .bert import BertEncoder


def _pad(seqs: Sequence[Sequence[int]], pad: int = 0):
    L = max(1, max(len(s) for s in seqs))
    ids = np.full((len(seqs), L), pad, dtype=np.int32)
    lens = np.zeros(len(seqs), dtype=np.int32)
    for i, s in enumerate(seqs):
        ids[i, :len(s)] = s
        lens[i] = len(s)
    return ids, lens


class _EncoderBase:
    def __init__(self, encoder: BertEncoder | None = None, model_dir: str | None = None, tokenizer: Any = None,
                 max_seq_length: int = 256, token_budget: int = 262144, device: int = 0):
        if encoder is None:
            if model_dir is None:
                raise ValueError("pass a BertEncoder or a checkpoint directory")
            encoder = BertEncoder.from_pretrained_dir(model_dir, device=device)
        self.encoder = encoder
        if tokenizer is None and model_dir is not None:
            import os
            vocab = os.path.join(model_dir, "vocab.txt")
            if os.path.exists(vocab):       # BERT WordPiece checkpoints: the native host tokenizer (rmu_tok_*)
                from .tokenizer import WordPieceTokenizer
                lower = True
                cfg = os.path.join(model_dir, "tokenizer_config.json")
                if os.path.exists(cfg):
                    import json
                    lower = bool(json.load(open(cfg)).get("do_lower_case", True))
                tokenizer = WordPieceTokenizer(vocab, do_lower_case=lower)
            else:
                from transformers import AutoTokenizer
                tokenizer = AutoTokenizer.from_pretrained(model_dir)
        self.tokenizer = tokenizer
        self.max_seq_length = min(int(max_seq_length), encoder.max_pos)
        self.token_budget = int(token_budget)

    def _batches(self, lens: np.ndarray):
        """Length-sorted batches under a token budget: yields index arrays (into the original order)."""
        order = np.argsort(-lens, kind="stable")
        i = 0
        while i < len(order):
            L = int(lens[order[i]])
            n = max(1, min(len(order) - i, self.token_budget // max(L, 1), 65535))
            yield order[i:i + n]
            i += n

    def _run(self, seqs, types, mode: int):
        import torch
        lens = np.asarray([len(s) for s in seqs], dtype=np.int32)
        out = torch.empty((len(seqs), 384) if mode == 0 else (len(seqs),), dtype=torch.float32, device=self.encoder.device)
        for idx in self._batches(lens):
            ids, ln = _pad([seqs[i] for i in idx])
            tt = None if types is None else _pad([types[i] for i in idx])[0]
            res = self.encoder.encode_ids(ids, ln, tt, mode=mode)
            out[torch.as_tensor(idx, device=out.device, dtype=torch.long)] = res
        return out


class MI355XEmbeddings(_EncoderBase, Embeddings):
    """Drop-in for langchain_huggingface.HuggingFaceEmbeddings on the reference's call sites; an instance of
    langchain_core's `Embeddings` when LangChain is installed (SemanticChunker(self.embeddings, ...),
    server/RAGHelper.py:336-341, and the vector stores type-check it)."""

    #: texts per pipeline block: while the GPU encodes block i the host tokenises block i+1.  Both the
    #: tokenizer (rmu_tok_encode) and the encoder (rmu_bert_encode) run in librmu.so with the GIL released.
    pipeline_block = 8192

    def _tokenize(self, texts: list[str]) -> list[list[int]]:
        if self.tokenizer is None:
            raise RuntimeError("no tokenizer: give model_dir/tokenizer, or call embed_ids with token ids")
        enc = self.tokenizer([t.replace("\n", " ") for t in texts], truncation=True, max_length=self.max_seq_length,
                             padding=False, add_special_tokens=True)
        return [list(x) for x in enc["input_ids"]]

    def embed_ids(self, seqs: Sequence[Sequence[int]]):
        """Pre-tokenised sequences (already carrying [CLS]/[SEP]) -> torch CUDA [n, 384] unit-norm fp32."""
        seqs = [list(s)[:self.max_seq_length] for s in seqs]
        return self._run(seqs, None, mode=0)

    def embed_documents_device(self, texts: list[str]):
        """texts -> torch CUDA [n, 384] unit-norm fp32, rows in text order.  The vector store appends this tensor to
        the HBM-resident corpus device-to-device (`rmu_index_add(is_device=1)`): embeddings never visit the host."""
        import torch
        texts = list(texts)
        blk = max(1, int(self.pipeline_block))
        if len(texts) <= blk:
            return self.embed_ids(self._tokenize(texts))
        # two-stage pipeline over blocks of texts: a worker thread tokenises the next block while this thread encodes
        from concurrent.futures import ThreadPoolExecutor
        out = torch.empty((len(texts), 384), dtype=torch.float32, device=self.encoder.device)
        starts = list(range(0, len(texts), blk))
        with ThreadPoolExecutor(max_workers=1) as pool:
            fut = pool.submit(self._tokenize, texts[starts[0]:starts[0] + blk])
            for i, lo in enumerate(starts):
                seqs = fut.result()
                if i + 1 < len(starts):
                    fut = pool.submit(self._tokenize, texts[starts[i + 1]:starts[i + 1] + blk])
                out[lo:lo + len(seqs)] = self.embed_ids(seqs)
        return out

    def embed_documents_array(self, texts: list[str]) -> np.ndarray:
        return self.embed_documents_device(texts).cpu().numpy()

    def embed_documents(self, texts: list[str]) -> list[list[float]]:
        return self.embed_documents_array(texts).tolist()

    def embed_query(self, text: str) -> list[float]:
        return self.embed_documents([text])[0]


class MI355XCrossEncoder(_EncoderBase, *CROSS_ENCODER_BASES):
    """Drop-in for langchain_community.cross_encoders.HuggingFaceCrossEncoder: `.score(text_pairs)`; an instance of every
    importable `BaseCrossEncoder` (the type of the reranker's `model` field, server/ScoredCrossEncoderReranker.py:15)."""

    def __init__(self, *a, max_seq_length: int = 512, **kw):
        super().__init__(*a, max_seq_length=max_seq_length, **kw)
        if not self.encoder.has_head:
            raise ValueError("checkpoint has no pooler/classifier head")

    def score_ids(self, seqs: Sequence[Sequence[int]], type_ids: Sequence[Sequence[int]]):
        seqs = [list(s)[:self.max_seq_length] for s in seqs]
        types = [list(t)[:self.max_seq_length] for t in type_ids]
        return self._run(seqs, types, mode=1)

    def score(self, text_pairs: list[tuple[str, str]]) -> list[float]:
        if not text_pairs:
            return []
        if self.tokenizer is None:
            raise RuntimeError("no tokenizer: give model_dir/tokenizer, or call score_ids with token ids")
        enc = self.tokenizer([p[0] for p in text_pairs], [p[1] for p in text_pairs], truncation="longest_first",
                             max_length=self.max_seq_length, padding=False, return_token_type_ids=True)
        return self.score_ids(enc["input_ids"], enc["token_type_ids"]).cpu().numpy().astype(np.float64).tolist()

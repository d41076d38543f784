"""MI355XEmbeddings / MI355XCrossEncoder -- the `Embeddings` and `BaseCrossEncoder` objects the reference
builds at server/RAGHelper_local.py:107-117 (HuggingFaceEmbeddings) and server/RAGHelper.py:483-486
(HuggingFaceCrossEncoder), with the forwards running in librmu.so.

Behaviour restated from the pinned dependencies:
  * embed_documents: "\\n" -> " " (langchain-huggingface), tokenize (truncate to the checkpoint's max_seq_length), BERT,
    the checkpoint's Pooling (masked mean | CLS) and Normalize if its modules.json lists one, -> list[list[float]];
    embed_query(t) = embed_documents([t])[0].
  * score(pairs): tokenize pair ([CLS] q [SEP] p [SEP], token types 0/1, longest-first truncation to 512),
    BertForSequenceClassification logit (num_labels = 1), the checkpoint's activation (Identity for the ms-marco
    models, Sigmoid when the config names none), -> list[float].
What the checkpoint directory declares is read by `ragmeup_amd.checkpoint`, never assumed; with a bare `encoder=`
(no directory) the all-MiniLM-L6-v2 settings BASELINE.json names apply: mean pooling, Normalize, max_seq_length 256.

Sequences are length-sorted and batched by a token budget (sentence-transformers sorts by length too; 1M padded tokens per
forward = ~9 GB of bf16 workspace of the 288 GB: an 8192-text pipeline block is ONE forward, which is where the fused
kernels are efficient); the packed-token kernels spend no FLOPs on padding.  Token ids stay in numpy arrays from `rmu_tok_encode` to
`rmu_bert_encode`: no per-text Python lists on the indexing path.

A tokenizer needs a vocabulary, which does not exist offline in the build image: pass `tokenizer=` (any
callable HF-style tokenizer) or a `model_dir` containing tokenizer files.  `embed_ids` / `score_ids` take
pre-tokenised input and are what the kernel benchmarks use.
"""
from __future__ import annotations

import sys
import threading
from typing import Any, Optional, Sequence

import numpy as np

from . import bert as B
from ._lc import CROSS_ENCODER_BASES, Embeddings
from .bert import BertEncoder


class _SwitchInterval:
    """Process-wide `sys.setswitchinterval` held short while at least one indexing pipeline runs: the first caller in saves the
    host application's value, the last caller out restores it (two overlapping calls must not leave the interpreter at 0.2 ms)."""
    _lock = threading.Lock()
    _users = 0
    _saved = None

    def __init__(self, seconds: float):
        self.seconds = seconds

    def __enter__(self):
        cls = _SwitchInterval
        with cls._lock:
            if cls._users == 0:
                cls._saved = sys.getswitchinterval()
                sys.setswitchinterval(min(cls._saved, self.seconds))
            cls._users += 1

    def __exit__(self, *exc):
        cls = _SwitchInterval
        with cls._lock:
            cls._users -= 1
            if cls._users == 0:
                sys.setswitchinterval(cls._saved)
        return False


def _pad(seqs: Sequence[Sequence[int]], pad: int = 0, max_len: Optional[int] = None):
    """list of id lists -> (ids [n, L] int32, lens [n] int32); sequences longer than max_len keep their head."""
    n = len(seqs)
    lens = np.fromiter((len(s) for s in seqs), dtype=np.int32, count=n)
    if max_len is not None:
        np.minimum(lens, max_len, out=lens)
    L = max(1, int(lens.max(initial=0)))
    ids = np.full((n, L), pad, dtype=np.int32)
    for i, s in enumerate(seqs):
        ids[i, :lens[i]] = s[:lens[i]]
    return ids, lens


class _EncoderBase:
    _default_max_seq_length = 256

    def __init__(self, encoder: BertEncoder | None = None, model_dir: str | None = None, tokenizer: Any = None,
                 max_seq_length: int | None = None, token_budget: int = 1 << 20, device: int = 0):
        self.spec = None
        if encoder is None:
            if model_dir is None:
                raise ValueError("pass a BertEncoder or a checkpoint directory")
            self.spec = self._read_spec(model_dir)
            encoder = BertEncoder.from_spec(self.spec, device=device, head=self._wants_head)
        self.encoder = encoder
        self._text_lower = False
        if self.spec is not None:
            self._text_lower = self.spec.st_lower_case
            if tokenizer is None:
                if self.spec.vocab_file:        # BERT WordPiece checkpoints: the native host tokenizer (rmu_tok_*)
                    from .tokenizer import WordPieceTokenizer
                    tokenizer = WordPieceTokenizer(self.spec.vocab_file, do_lower_case=self.spec.tok_lower_case)
                else:
                    from transformers import AutoTokenizer
                    tokenizer = AutoTokenizer.from_pretrained(model_dir)
            if max_seq_length is None:
                max_seq_length = self.spec.max_seq_length
        if max_seq_length is None:
            max_seq_length = self._default_max_seq_length
        self.tokenizer = tokenizer
        self.max_seq_length = min(int(max_seq_length), encoder.max_pos)
        self.token_budget = int(token_budget)

    _wants_head: bool | None = None

    def _read_spec(self, model_dir):      # overridden
        raise NotImplementedError

    # ---- tokenisation: texts -> padded numpy arrays ---------------------------------------------------------------
    def _tokenize_arrays(self, texts_a: list[str], texts_b: list[str] | None = None, want_types: bool = False):
        """-> ids [n, L] int32, type_ids [n, L] int32 | None, lens [n] int32 (host numpy)."""
        if self.tokenizer is None:
            raise RuntimeError("no tokenizer: give model_dir/tokenizer, or call embed_ids / score_ids with token ids")
        from .tokenizer import WordPieceTokenizer
        if isinstance(self.tokenizer, WordPieceTokenizer):
            ids, tt, lens = self.tokenizer.encode(texts_a, texts_b, max_len=self.max_seq_length)
            L = max(1, int(lens.max(initial=1)))
            return ids[:, :L], (tt[:, :L] if want_types else None), lens
        if texts_b is None:
            enc = self.tokenizer(texts_a, truncation=True, max_length=self.max_seq_length, padding=False, add_special_tokens=True)
        else:
            enc = self.tokenizer(texts_a, texts_b, truncation="longest_first", max_length=self.max_seq_length, padding=False,
                                 return_token_type_ids=True)
        ids, lens = _pad(enc["input_ids"], max_len=self.max_seq_length)
        tt = _pad(enc["token_type_ids"], max_len=self.max_seq_length)[0] if want_types else None
        return ids, tt, lens

    #: a block that fits the token budget runs as one forward in input order (no length sort)
    one_forward = True

    def _batches(self, lens: np.ndarray):
        """Length-sorted batches under a token budget: yields index arrays (into the original order)."""
        order = np.argsort(-lens, kind="stable")
        i = 0
        while i < len(order):
            L = int(lens[order[i]])
            n = max(1, min(len(order) - i, self.token_budget // max(L, 1), 65535))
            yield order[i:i + n]
            i += n

    def _run_arrays(self, ids: np.ndarray, types: np.ndarray | None, lens: np.ndarray, mode: int, out=None):
        """Padded id arrays -> torch CUDA [n, 384] (pooling modes) | [n] (cross-encoder), rows in input order."""
        import torch
        n = ids.shape[0]
        if out is None:
            out = torch.empty((n, 384) if (mode & 0xff) != B.MODE_CE else (n,), dtype=torch.float32, device=self.encoder.device)
        if n == 0:
            return out
        lens = np.minimum(np.asarray(lens, dtype=np.int32), min(ids.shape[1], self.max_seq_length))
        Lmax = max(1, int(lens.max()))
        if self.one_forward and n * Lmax <= self.token_budget and n <= 65535:
            # the whole block is ONE forward: no length sort (the packed-token kernels spend nothing on padding; sorting only
            # serves to cut a block into batches), no gather of the id rows, no scatter of the result rows
            self.encoder.encode_ids(ids[:, :Lmax], lens, None if types is None else types[:, :Lmax], mode=mode, out=out)
            return out
        for idx in self._batches(lens):
            L = max(1, int(lens[idx[0]]))                          # length-sorted: the first is the longest
            res = self.encoder.encode_ids(ids[idx, :L], lens[idx], None if types is None else types[idx, :L], mode=mode)
            out[torch.as_tensor(idx, device=out.device, dtype=torch.long)] = res
        return out

    def _run(self, seqs, types, mode: int):
        ids, lens = _pad(seqs, max_len=self.max_seq_length)
        tt = None if types is None else _pad(types, max_len=self.max_seq_length)[0]
        return self._run_arrays(ids, tt, lens, mode)


class MI355XEmbeddings(_EncoderBase, Embeddings):
    """Drop-in for langchain_huggingface.HuggingFaceEmbeddings on the reference's call sites; an instance of
    langchain_core's `Embeddings` when LangChain is installed (SemanticChunker(self.embeddings, ...),
    server/RAGHelper.py:336-341, and the vector stores type-check it)."""

    #: texts per pipeline block: while the GPU encodes block i the host tokenises block i+1.  Both the
    #: tokenizer (rmu_tok_encode) and the encoder (rmu_bert_encode) run in librmu.so with the GIL released.
    pipeline_block = 8192
    _wants_head = False

    def __init__(self, *a, pooling: str | None = None, normalize: bool | None = None, **kw):
        super().__init__(*a, **kw)
        from . import checkpoint as C
        spec = self.spec
        self.pooling = pooling if pooling is not None else (spec.pooling if spec is not None else C.POOL_MEAN)
        self.normalize = bool(normalize) if normalize is not None else (spec.normalize if spec is not None else True)
        if self.pooling not in (C.POOL_MEAN, C.POOL_CLS):
            raise ValueError(f"pooling={self.pooling!r}: 'mean' or 'cls'")
        self._mode = (B.MODE_MEAN if self.pooling == C.POOL_MEAN else B.MODE_CLS) | (0 if self.normalize else B.NO_NORMALIZE)

    def _read_spec(self, model_dir):
        from . import checkpoint as C
        return C.read_sentence_transformer(model_dir)

    def _prep(self, texts: list[str]) -> list[str]:
        from .tokenizer import WordPieceTokenizer
        # langchain-huggingface embed_documents replaces "\n" by " " before sentence-transformers sees the text; the native
        # tokenizer treats both as whitespace (BERT's BasicTokenizer), so the copy of every text is skipped there
        out = texts if isinstance(self.tokenizer, WordPieceTokenizer) else [t.replace("\n", " ") for t in texts]
        if self._text_lower:
            out = [t.lower() for t in out]                           # sentence_bert_config.json do_lower_case
        return out

    def _tokenize(self, texts: list[str]):
        ids, _, lens = self._tokenize_arrays(self._prep(texts))
        return ids, lens

    def embed_ids(self, seqs: Sequence[Sequence[int]]):
        """Pre-tokenised sequences (already carrying [CLS]/[SEP]) -> torch CUDA [n, 384] fp32."""
        return self._run(seqs, None, mode=self._mode)

    def embed_id_arrays(self, ids: np.ndarray, lens: np.ndarray, out=None):
        """Padded [n, L] id array + lengths -> torch CUDA [n, 384] fp32 (rows in input order)."""
        return self._run_arrays(np.asarray(ids), None, np.asarray(lens), self._mode, out=out)

    def embed_documents_device(self, texts: list[str], out=None):
        """texts -> torch CUDA [n, 384] fp32, rows in text order.  The vector store appends this tensor to
        the HBM-resident corpus device-to-device (`rmu_index_add(is_device=1)`): embeddings never visit the host."""
        import torch
        texts = list(texts)
        blk = max(1, int(self.pipeline_block))
        if len(texts) <= blk:
            return self.embed_id_arrays(*self._tokenize(texts), out=out)
        # two-stage pipeline over blocks of texts: a worker thread tokenises the next block while this thread encodes
        from concurrent.futures import ThreadPoolExecutor
        if out is None:
            out = torch.empty((len(texts), 384), dtype=torch.float32, device=self.encoder.device)
        # blocks: a short first one (its tokenisation has nothing to hide behind), then `blk` texts each
        first = min(blk, max(1024, blk // 4))
        starts = [0] + list(range(first, len(texts), blk))
        ends = starts[1:] + [len(texts)]
        can_upload = isinstance(self.encoder, BertEncoder)

        def prepare(i):
            """tokenise block i and (native encoder) put its ids on the device: everything the forward needs, off its critical path"""
            ids, lens = self._tokenize(texts[starts[i]:ends[i]])
            lens = np.minimum(np.asarray(lens, dtype=np.int32), min(ids.shape[1], self.max_seq_length))
            Lmax = max(1, int(lens.max(initial=1)))
            if can_upload and self.one_forward and ids.shape[0] * Lmax <= self.token_budget:
                ids_d, lens_d = self.encoder.upload([ids[:, :Lmax], lens], slot=f"blk{i & 1}", min_cap=self.token_budget)
                return ids_d, lens_d, True
            return ids, lens, False

        # Both stages spend their time in librmu.so with the GIL released, but each needs it back for a few lines of Python per
        # block; with CPython's default 5 ms switch interval the encoding thread waited up to that long behind the tokenising
        # thread's string handling, every block (~30 ms per 65536 texts measured).  A short interval for the duration of the call.
        import contextlib
        # the upload slots belong to ONE pipeline at a time: a second thread embedding a large batch on the same encoder waits here
        # (the GPU is the shared resource either way)
        guard = self.encoder.pipeline_lock if can_upload else contextlib.nullcontext()
        # (round 5) Block i's forward is ENQUEUED on a stream of its own and left in flight; block i + 1's is queued behind it as soon as its
        # ids are on the device, so the GPU never waits for this thread between two blocks (a synchronous call per block left it idle for
        # the return trip through Python, every block, with the tokenizer's threads competing for the cores: 0.85 of the encoder-only
        # rate).  A block's upload slot is reused two blocks later: the tokenising of block i + 1 starts once block i - 1's forward has ended.
        fwd = getattr(self.encoder, "_fwd_stream", None) if can_upload else None
        if can_upload and fwd is None:
            fwd = self.encoder._fwd_stream = torch.cuda.Stream(self.encoder.device)
        if fwd is not None:
            # `out` came from the caching allocator on the CURRENT stream: a block it handed back may still be read by work queued there
            # (the consumer of an earlier result); the forwards that overwrite it run on `fwd` -- order them behind
            fwd.wait_stream(torch.cuda.current_stream(self.encoder.device))
        with _SwitchInterval(2e-4), guard, ThreadPoolExecutor(max_workers=1) as pool:
            fut = pool.submit(prepare, 0)
            done_prev = None                                # event: the forward enqueued one iteration ago has ended
            try:
                for i, lo in enumerate(starts):
                    ids, lens, on_device = fut.result()
                    dst = out[lo:lo + ids.shape[0]]
                    done = None
                    if on_device:
                        self.encoder.encode_ids(ids, lens, None, mode=self._mode, out=dst, stream=fwd)
                        done = torch.cuda.Event()
                        done.record(fwd)
                    if done_prev is not None:
                        done_prev.synchronize()             # its upload slot is block i + 1's
                    if i + 1 < len(starts):
                        fut = pool.submit(prepare, i + 1)
                    if not on_device:                       # (a block over the token budget: the synchronous path, ordered behind by the library)
                        self.embed_id_arrays(ids, lens, out=dst)
                    done_prev = done
            finally:
                if fwd is not None:
                    fwd.synchronize()                       # results complete on return; nothing of this call stays in flight on an error
        return out

    def tokenize_for_index(self, texts: list[str]):
        """The HOST half of `embed_documents_device` for one pipeline block (<= pipeline_block texts): (ids, lens) numpy arrays, or
        None when this object is not backed by the native tokenizer + encoder.  MI355XVectorStore runs it for call i + 1 while the
        GPU half of call i (`embed_token_arrays_device` + the index append) is still in flight."""
        if not self.can_tokenize_for_index():
            return None
        return self._tokenize(list(texts))

    def can_tokenize_for_index(self) -> bool:
        """Whether `tokenize_for_index` / `embed_token_arrays_device` are available: the native tokenizer AND the native encoder."""
        from .tokenizer import WordPieceTokenizer
        return isinstance(getattr(self, "encoder", None), BertEncoder) and isinstance(getattr(self, "tokenizer", None), WordPieceTokenizer)

    def enqueue_token_arrays(self, parts: list, round_no: int):
        """The GPU half, left IN FLIGHT: `parts` = the (ids, lens) pairs of `tokenize_for_index` of one or more calls, in row order ->
        (torch CUDA [n, 384] fp32, event that marks its end), or None when these rows have to take `embed_token_arrays_device` (no native
        encoder, more tokens than one forward).  The ids are written part by part into the pinned staging slot `add<round_no & 1>` (no
        concatenated host copy in between): a caller keeps at most two rounds going and waits for round r before it starts round r + 2."""
        import torch
        if not isinstance(getattr(self, "encoder", None), BertEncoder) or not self.one_forward or not parts:
            return None
        cut = [np.minimum(np.asarray(ln, dtype=np.int32), min(a.shape[1], self.max_seq_length)) for a, ln in parts]
        n = int(sum(a.shape[0] for a, _ in parts))
        Lmax = max(1, max(int(c.max(initial=1)) for c in cut))
        if n == 0 or n * Lmax > self.token_budget or n > 65535:
            return None
        enc = self.encoder
        fwd = getattr(enc, "_fwd_stream", None)
        if fwd is None:
            fwd = enc._fwd_stream = torch.cuda.Stream(enc.device)
        ids_d, lens_d = enc.upload_rows([a for a, _ in parts], cut, Lmax, slot=f"add{round_no & 1}", min_cap=self.token_budget)
        out = torch.empty((n, 384), dtype=torch.float32, device=enc.device)
        fwd.wait_stream(torch.cuda.current_stream(enc.device))      # `out` may be a recycled block still read on the allocating stream
        enc.encode_ids(ids_d, lens_d, None, mode=self._mode, out=out, stream=fwd)
        done = torch.cuda.Event()
        done.record(fwd)
        return out, done

    def embed_token_arrays_device(self, ids: np.ndarray, lens: np.ndarray):
        """The GPU half: token arrays of `tokenize_for_index` -> torch CUDA [n, 384] fp32 (what `embed_documents_device` returns)."""
        return self.embed_id_arrays(ids, lens)

    def embed_documents_array(self, texts: list[str]) -> np.ndarray:
        return self.embed_documents_device(texts).cpu().numpy()

    def embed_documents(self, texts: list[str]) -> list[list[float]]:
        return self.embed_documents_array(texts).tolist()

    def query_ids(self, text: str):
        """(ids [1, L] int32, lens [1]) of one query when it can go through the library's host entry points (native encoder +
        tokenizer, <= 256 tokens), else None.  What MI355XVectorStore hands to `BertEncoder.search_host`."""
        enc = getattr(self, "encoder", None)
        if not isinstance(enc, BertEncoder) or getattr(self, "tokenizer", None) is None:
            return None
        ids, lens = self._tokenize([text])
        return (ids, lens) if ids.shape[1] <= enc.SMALL_TOKENS else None

    def _embed_query_fast(self, text: str):
        """One query of up to 256 tokens through the graph-replayed host entry point (rmu_bert_encode_host): the reference's
        per-request pattern (one embed_query per /chat call, server/RAGHelper.py:497-499) is launch-bound.  None when this object
        is not backed by the native encoder + tokenizer (subclasses that only override embed_documents keep working)."""
        enc = getattr(self, "encoder", None)
        if not isinstance(enc, BertEncoder) or getattr(self, "tokenizer", None) is None:
            return None
        ids, lens = self._tokenize([text])
        if ids.shape[1] > enc.SMALL_TOKENS:
            return self.embed_id_arrays(ids, lens).cpu().numpy()[0]
        return enc.encode_host(ids, lens, None, self._mode)[0]

    def embed_query_array(self, text: str) -> np.ndarray:
        v = self._embed_query_fast(text)
        return v if v is not None else np.asarray(self.embed_documents([text])[0], dtype=np.float32)

    def embed_query(self, text: str) -> list[float]:
        v = self._embed_query_fast(text)
        return v.tolist() if v is not None else self.embed_documents([text])[0]

    def token_embeddings_ids(self, ids: np.ndarray, lens: np.ndarray):
        """Final hidden state of every token (sentence-transformers output_value="token_embeddings"): one batch, packed
        [sum(lens), 384] fp32 torch CUDA in input order."""
        return self.encoder.encode_ids(np.asarray(ids), np.asarray(lens), None, mode=B.MODE_TOKENS)


class MI355XCrossEncoder(_EncoderBase, *CROSS_ENCODER_BASES):
    """Drop-in for langchain_community.cross_encoders.HuggingFaceCrossEncoder: `.score(text_pairs)`; an instance of every
    importable `BaseCrossEncoder` (the type of the reranker's `model` field, server/ScoredCrossEncoderReranker.py:15)."""

    _default_max_seq_length = 512
    _wants_head = True

    def __init__(self, *a, activation: str | None = None, **kw):
        super().__init__(*a, **kw)
        if not self.encoder.has_head:
            raise ValueError("checkpoint has no pooler/classifier head")
        # bare encoder: Identity, the ms-marco cross-encoders' configured activation (BASELINE.json configs[4])
        self.activation = activation if activation is not None else (self.spec.activation if self.spec is not None else "identity")
        if self.activation not in ("identity", "sigmoid"):
            raise ValueError(f"activation={self.activation!r}: 'identity' or 'sigmoid'")

    def _read_spec(self, model_dir):
        from . import checkpoint as C
        return C.read_cross_encoder(model_dir)

    def _activate(self, logits):
        return logits if self.activation == "identity" else logits.sigmoid()

    def score_ids(self, seqs: Sequence[Sequence[int]], type_ids: Sequence[Sequence[int]]):
        return self._activate(self._run(seqs, type_ids, mode=B.MODE_CE))

    def score_id_arrays(self, ids: np.ndarray, type_ids: np.ndarray, lens: np.ndarray):
        return self._activate(self._run_arrays(np.asarray(ids), np.asarray(type_ids), np.asarray(lens), B.MODE_CE))

    def score(self, text_pairs: list[tuple[str, str]]) -> list[float]:
        if not text_pairs:
            return []
        ids, tt, lens = self._tokenize_arrays([p[0] for p in text_pairs], [p[1] for p in text_pairs], want_types=True)
        enc = self.encoder
        lens = np.minimum(np.asarray(lens, dtype=np.int32), min(ids.shape[1], self.max_seq_length))
        Lmax = max(1, int(lens.max()))
        if isinstance(enc, BertEncoder) and enc.host_shape(ids.shape[0], Lmax, B.MODE_CE) is not None:
            # the reference's rerank call (<= 14 pairs, ScoredCrossEncoderReranker.py:42): host ids in, host logits out, one graph replay
            logits = enc.encode_host(ids[:, :Lmax], lens, tt[:, :Lmax], B.MODE_CE).astype(np.float64)
            if self.activation == "sigmoid":
                logits = 1.0 / (1.0 + np.exp(-logits))
            return logits.tolist()
        return self.score_id_arrays(ids, tt, lens).cpu().numpy().astype(np.float64).tolist()

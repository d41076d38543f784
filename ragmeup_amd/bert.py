"""BertEncoder -- Python handle on the hand-written BERT-6x384 forward in librmu.so (rmu_bert_*).

Serves the two transformer forwards on the reference's hot path:
  * sentence-transformers bi-encoder: BertModel -> Pooling (masked mean | CLS token) -> [Normalize]   (MODE_MEAN / MODE_CLS)
  * CrossEncoder: BertForSequenceClassification(num_labels=1) logit                                     (MODE_CE)
  * the final hidden state of every token (ST output_value="token_embeddings"; what the parity tests compare)  (MODE_TOKENS)
PyTorch is used only to hold the weight / id tensors on the device; every FLOP runs in our kernels.
"""
from __future__ import annotations

import ctypes
import threading
import os
from typing import Mapping

import numpy as np

from . import _native as N


MODE_MEAN, MODE_CE, MODE_CLS, MODE_TOKENS = 0, 1, 2, 3      # include/rmu.h RMU_BERT_*
NO_NORMALIZE = 0x100


class BertCfg(ctypes.Structure):
    _fields_ = [("vocab_size", ctypes.c_int), ("hidden", ctypes.c_int), ("layers", ctypes.c_int),
                ("heads", ctypes.c_int), ("ffn", ctypes.c_int), ("max_pos", ctypes.c_int),
                ("type_vocab", ctypes.c_int), ("ln_eps", ctypes.c_float), ("has_head", ctypes.c_int)]


def weight_order(layers: int, has_head: bool) -> list[str]:
    """Order of the fp32 tensors rmu_bert_create expects (HF parameter names, 'bert.' prefix stripped)."""
    names = ["embeddings.word_embeddings.weight", "embeddings.position_embeddings.weight",
             "embeddings.token_type_embeddings.weight", "embeddings.LayerNorm.weight", "embeddings.LayerNorm.bias"]
    for l in range(layers):
        p = f"encoder.layer.{l}."
        names += [p + "attention.self.query.weight", p + "attention.self.query.bias",
                  p + "attention.self.key.weight", p + "attention.self.key.bias",
                  p + "attention.self.value.weight", p + "attention.self.value.bias",
                  p + "attention.output.dense.weight", p + "attention.output.dense.bias",
                  p + "attention.output.LayerNorm.weight", p + "attention.output.LayerNorm.bias",
                  p + "intermediate.dense.weight", p + "intermediate.dense.bias",
                  p + "output.dense.weight", p + "output.dense.bias",
                  p + "output.LayerNorm.weight", p + "output.LayerNorm.bias"]
    if has_head:
        names += ["pooler.dense.weight", "pooler.dense.bias", "classifier.weight", "classifier.bias"]
    return names


def _strip(state: Mapping) -> dict:
    out = {}
    for k, v in state.items():
        for pre in ("bert.", "0.auto_model.", "auto_model.", "model."):
            if k.startswith(pre):
                k = k[len(pre):]
        out[k] = v
    return out


class BertEncoder:
    """weights: mapping HF-name -> array/tensor (fp32).  Use `from_pretrained_dir` for a checkpoint directory."""

    HIDDEN = 384

    def __init__(self, weights: Mapping, layers: int = 6, heads: int = 12, ffn: int = 1536, ln_eps: float = 1e-12,
                 has_head: bool | None = None, device: int = 0):
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("BertEncoder needs an MI355X: the encoder has no CPU fallback")
        self._torch = torch
        self.device = torch.device("cuda", device)
        self._lib = N.lib()
        N.check(self._lib.rmu_init(device), "rmu_init")
        w = _strip(weights)
        if has_head is None:
            has_head = "classifier.weight" in w
        self.has_head = bool(has_head)
        names = weight_order(layers, self.has_head)
        missing = [n for n in names if n not in w]
        if missing:
            raise KeyError(f"checkpoint lacks {missing[:4]}{'...' if len(missing) > 4 else ''}")
        tens = [torch.as_tensor(np.asarray(w[n], dtype=np.float32) if not torch.is_tensor(w[n]) else w[n])
                .to(self.device, torch.float32).contiguous() for n in names]
        hid = tens[0].shape[1]
        cfg = BertCfg(int(tens[0].shape[0]), int(hid), int(layers), int(heads), int(ffn), int(tens[1].shape[0]),
                      int(tens[2].shape[0]), float(ln_eps), int(self.has_head))
        ptrs = (ctypes.c_void_p * len(tens))(*[t.data_ptr() for t in tens])
        torch.cuda.synchronize(self.device)
        h = ctypes.c_void_p()
        N.check(self._lib.rmu_bert_create(ctypes.byref(h), ctypes.byref(cfg), ptrs, len(tens)), "rmu_bert_create")
        self._h = h
        self._lock = threading.Lock()              # one forward at a time per encoder (the library serialises them anyway)
        self._stage = {}                           # slot -> (pinned host int32, device int32) staging pair
        self._side = None                          # side stream of `upload`
        self.pipeline_lock = threading.Lock()      # one user of the `upload` slots at a time (MI355XEmbeddings' block pipeline)
        self.max_pos = int(tens[1].shape[0])
        self.vocab_size = int(tens[0].shape[0])
        del tens                                   # the library keeps its own (bf16 / fp32) copies

    @classmethod
    def from_transformers(cls, model, device: int = 0) -> "BertEncoder":
        c = model.config
        return cls(model.state_dict(), layers=c.num_hidden_layers, heads=c.num_attention_heads,
                   ffn=c.intermediate_size, ln_eps=c.layer_norm_eps, device=device)

    @classmethod
    def from_pretrained_dir(cls, path: str, device: int = 0, head: bool | None = None) -> "BertEncoder":
        """HF checkpoint directory (config.json + model.safetensors | pytorch_model.bin); the architecture is validated
        by ragmeup_amd.checkpoint (anything this build does not execute raises)."""
        from . import checkpoint as C
        spec = C._common(path)
        return cls.from_spec(spec, device=device, head=head)

    @classmethod
    def from_spec(cls, spec, device: int = 0, head: bool | None = None) -> "BertEncoder":
        from . import checkpoint as C
        a = spec.arch
        return cls(C.load_state(spec), layers=a.layers, heads=a.heads, ffn=a.ffn, ln_eps=a.ln_eps, has_head=head, device=device)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.rmu_bert_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- forward ------------------------------------------------------------------------------------------
    SMALL_TOKENS = 256          # one query through the host entry points (embed_query)
    HOST_TOKENS = 4096          # rmu_bert_encode_host / rmu_bert_search_mmr: bucketed batch * max_len they accept
    HOST_ROWS = 256             # ... and result rows (pooled vectors / token states) per call

    @staticmethod
    def _bucket(n: int, steps=(1, 2, 4, 8, 12, 16, 24, 32, 48, 64, 96, 128, 192, 256)) -> int:
        for v in steps:
            if n <= v:
                return v
        return n

    def host_shape(self, batch: int, max_len: int, mode: int) -> tuple[int, int] | None:
        """The bucketed (batch, max_len) a host call of this shape runs at -- the library keeps one captured graph per SHAPE, so
        shapes are rounded up to a few sizes (a padded sequence has length 0: the packed-token kernels spend nothing on it, the
        results are those of the unpadded call bit for bit) -- or None when the call does not fit the host entry points."""
        kind = mode & 0xff
        lb = min(-(-max(int(max_len), 1) // 32) * 32, self.max_pos) if max_len > 16 else 16
        if kind == MODE_TOKENS:                            # packed token states: the row count is the token count, no padding rows
            return (batch, lb) if batch * lb <= self.HOST_ROWS else None
        bb = self._bucket(int(batch))
        if bb * lb > self.HOST_TOKENS or (kind != MODE_CE and bb > self.HOST_ROWS):
            bb = int(batch)                                # the exact batch may still fit
        if bb * lb > self.HOST_TOKENS or (kind != MODE_CE and bb > self.HOST_ROWS) or lb < max_len:
            return None
        return bb, lb

    def _host_arrays(self, ids, lens, type_ids, mode):
        ids = np.asarray(ids)
        lens = np.asarray(lens)
        B, L = ids.shape
        shp = self.host_shape(B, L, mode)
        if shp is None:
            raise ValueError(f"a host call carries at most {self.HOST_TOKENS} tokens and {self.HOST_ROWS} result rows: got batch {B} x max_len {L}")
        bb, lb = shp
        pi = np.zeros((bb, lb), dtype=np.int32)
        pi[:B, :L] = ids
        pl = np.zeros((bb,), dtype=np.int32)
        pl[:B] = np.minimum(np.maximum(lens, 0), L)
        pt = None
        if type_ids is not None:
            pt = np.zeros((bb, lb), dtype=np.int32)
            pt[:B, :L] = type_ids
        return pi, pl, pt, B, bb, lb

    def encode_host(self, ids, lens, type_ids=None, mode: int = 0) -> np.ndarray:
        """The interactive path (embed_query, the <= 14 pairs of a rerank call): HOST int arrays in, HOST fp32 array out, up to 4096
        tokens.  One captured hipGraph per bucketed input shape is replayed inside librmu.so (rmu_bert_encode_host): one graph
        launch + one synchronisation instead of ~45 launches and three torch tensor copies."""
        pi, pl, pt, B, bb, lb = self._host_arrays(ids, lens, type_ids, mode)
        kind = mode & 0xff
        if kind == MODE_CE:
            out = np.empty((bb,), dtype=np.float32)
        elif kind == MODE_TOKENS:
            out = np.empty((int(pl.sum()), self.HIDDEN), dtype=np.float32)
        else:
            out = np.empty((bb, self.HIDDEN), dtype=np.float32)
        N.check(self._lib.rmu_bert_encode_host(self._h, pi.ctypes.data, pt.ctypes.data if pt is not None else None, pl.ctypes.data,
                                               int(bb), int(lb), int(mode), out.ctypes.data, 1 if kind == MODE_CE else self.HIDDEN),
                "rmu_bert_encode_host")
        return out if kind == MODE_TOKENS else out[:B]

    def search_host(self, index, ids, lens, mode: int, fetch_k: int, k: int, lambda_mult: float | None = 0.5, row_base: int = 0,
                    want_vectors: bool = False):
        """rmu_bert_search_mmr: the query's token ids in, result rows out -- forward (graph replay), dense top-fetch_k and the greedy
        MMR selection (lambda_mult None: no selection, the top-k in score order) in ONE library call with ONE synchronisation; the
        query vector never visits the host.  -> rows [B, k] int64, scores [B, k] fp32 (, vectors [B, 384] fp32)."""
        pi, pl, pt, B, bb, lb = self._host_arrays(ids, lens, None, mode)
        rows = np.empty((bb, int(k)), dtype=np.int64)
        scores = np.empty((bb, int(k)), dtype=np.float32)
        vecs = np.empty((bb, self.HIDDEN), dtype=np.float32) if want_vectors else None
        N.check(self._lib.rmu_bert_search_mmr(self._h, index._h, pi.ctypes.data, None, pl.ctypes.data, int(bb), int(lb), int(mode), int(fetch_k),
                                              int(k), -1.0 if lambda_mult is None else float(lambda_mult), int(row_base), rows.ctypes.data,
                                              scores.ctypes.data, vecs.ctypes.data if vecs is not None else None), "rmu_bert_search_mmr")
        return (rows[:B], scores[:B], vecs[:B]) if want_vectors else (rows[:B], scores[:B])

    def encode_ids(self, ids, lens, type_ids=None, mode: int = 0, out=None, stream=None):
        """ids [B, L] int (numpy or torch, padded), lens [B].  mode = MODE_MEAN / MODE_CLS (| NO_NORMALIZE) -> [B, 384] fp32
        (torch CUDA); MODE_CE -> [B] fp32 logits; MODE_TOKENS -> [sum(min(lens, L)), 384] fp32, sequences packed in batch order.
        `out` may be a pre-allocated CUDA tensor (e.g. a slice of a corpus matrix).
        `stream` (a torch.cuda.Stream; inputs must be int32 CUDA tensors that are complete, `out` given): the forward is ENQUEUED on that
        stream and the call returns with it in flight -- the caller synchronises (or records an event) before reading `out` or reusing the
        inputs.  The library orders this model's next call on any other stream behind it."""
        torch = self._torch
        with self._lock:
            if stream is not None:
                ok = (all(torch.is_tensor(t) and t.is_cuda and t.dtype == torch.int32 and t.is_contiguous() for t in (ids, lens))
                      and type_ids is None and out is not None)
                if not ok:
                    raise ValueError("encode_ids(stream=...): ids / lens must be contiguous int32 CUDA tensors, no type ids, `out` given")
                B, L = ids.shape
                stride = 1 if (mode & 0xff) == MODE_CE else out.stride(0)
                N.check(self._lib.rmu_bert_encode(self._h, ids.data_ptr(), None, lens.data_ptr(), int(B), int(L), int(mode), out.data_ptr(),
                                                  int(stride), int(stream.cuda_stream)), "rmu_bert_encode")
                return out
            return self._encode_ids_locked(ids, lens, type_ids, mode, out)

    def _stage_in(self, a, slot: str, min_cap: int = 1 << 16):
        """HOST array -> device int32 through this encoder's pinned staging buffer.  A transient pageable buffer handed to the
        runtime gets registered with the GPU for the copy; freeing it (munmap) then invalidates the registration through the
        kernel driver, which stalls the device queues for ~90 ms at unpredictable later points (measured: add_documents in
        1000-text calls, 28k instead of 180k chunks/s).  Pinned staging never registers anything."""
        torch = self._torch
        if torch.is_tensor(a):
            if a.is_cuda:
                return a.to(self.device, torch.int32).contiguous()
            a = a.numpy()
        a = np.asarray(a)
        n = a.size
        st = self._stage.get(slot)
        if st is None or st[0].numel() < n:
            cap = max(n + n // 4, min_cap)
            st = (torch.empty(cap, dtype=torch.int32).pin_memory(), torch.empty(cap, dtype=torch.int32, device=self.device))
            self._stage[slot] = st
        if n:
            np.copyto(st[0][:n].numpy().reshape(a.shape), a, casting="unsafe")     # strided views and int64 copy straight in
            st[1][:n].copy_(st[0][:n], non_blocking=True)       # complete before this call returns: encode_ids synchronises
        return st[1][:n].view(a.shape)

    def upload(self, arrays, slot: str, min_cap: int = 1 << 16):
        """HOST int arrays -> device int32 tensors through the staging pair named `slot`, on a side stream, complete on return.
        For a caller that prepares the NEXT block while `encode_ids` runs the current one (MI355XEmbeddings' pipeline: the staging
        copy + H2D of a block's ids then no longer sit between two forwards); the tensors stay valid until `slot` is used again."""
        torch = self._torch
        if self._side is None:
            self._side = torch.cuda.Stream(self.device)
        with torch.cuda.stream(self._side):
            outs = [self._stage_in(a, f"{slot}.{i}", min_cap if i == 0 else 1 << 16) for i, a in enumerate(arrays)]   # sized once: pinning is slow
        self._side.synchronize()
        return outs

    def upload_rows(self, id_parts, len_parts, width: int, slot: str, min_cap: int = 1 << 16):
        """Several HOST id arrays [n_i, L_i] (row order) + their length arrays -> ONE device int32 [sum n_i, width] (columns past L_i zero) and
        the lengths [sum n_i], written part by part into the pinned staging pair `slot` -- what `upload` does for one array, without the
        concatenated host copy in front of it.  Complete on return; valid until `slot` is used again."""
        torch = self._torch
        if self._side is None:
            self._side = torch.cuda.Stream(self.device)
        n = int(sum(a.shape[0] for a in id_parts))
        with torch.cuda.stream(self._side):
            pair_i = self._stage_pair(f"{slot}.0", n * width, min_cap)
            pair_l = self._stage_pair(f"{slot}.1", n, 1 << 16)
            hv = pair_i[0][:n * width].numpy().reshape(n, width)
            hl = pair_l[0][:n].numpy()
            lo = 0
            for a, ln in zip(id_parts, len_parts):
                m, w = a.shape[0], min(a.shape[1], width)
                np.copyto(hv[lo:lo + m, :w], a[:, :w], casting="unsafe")
                if w < width:
                    hv[lo:lo + m, w:] = 0
                np.copyto(hl[lo:lo + m], ln, casting="unsafe")
                lo += m
            pair_i[1][:n * width].copy_(pair_i[0][:n * width], non_blocking=True)
            pair_l[1][:n].copy_(pair_l[0][:n], non_blocking=True)
        self._side.synchronize()
        return pair_i[1][:n * width].view(n, width), pair_l[1][:n]

    def _stage_pair(self, slot: str, n: int, min_cap: int):
        """(pinned host int32, device int32) staging tensors of at least n elements under the name `slot` (see _stage_in)"""
        torch = self._torch
        st = self._stage.get(slot)
        if st is None or st[0].numel() < n:
            cap = max(n + n // 4, min_cap)
            st = (torch.empty(cap, dtype=torch.int32).pin_memory(), torch.empty(cap, dtype=torch.int32, device=self.device))
            self._stage[slot] = st
        return st

    def _encode_ids_locked(self, ids, lens, type_ids, mode, out):
        torch = self._torch
        lens_host = None if torch.is_tensor(lens) else np.asarray(lens)
        ids_t = self._stage_in(ids, "ids")
        lens_t = self._stage_in(lens, "lens")
        tt_t = None if type_ids is None else self._stage_in(type_ids, "types")
        B, L = ids_t.shape
        kind = mode & 0xff
        if out is None:
            if kind == MODE_CE:
                shape = (B,)
            elif kind == MODE_TOKENS:
                n_tok = int(np.minimum(np.maximum(np.asarray(lens.cpu() if lens_host is None else lens_host, dtype=np.int64), 0), L).sum())
                shape = (n_tok, self.HIDDEN)
            else:
                shape = (B, self.HIDDEN)
            out = torch.empty(shape, dtype=torch.float32, device=self.device)
        stride = 1 if kind == MODE_CE else out.stride(0)
        torch.cuda.current_stream(self.device).synchronize()
        N.check(self._lib.rmu_bert_encode(self._h, ids_t.data_ptr(), tt_t.data_ptr() if tt_t is not None else None,
                                          lens_t.data_ptr(), int(B), int(L), int(mode), out.data_ptr(), int(stride), 0),
                "rmu_bert_encode")
        return out

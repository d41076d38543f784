"""WordPieceTokenizer -- Python handle on the host C++ tokenizer in librmu.so (rmu_tok_*).
Restates transformers' BertTokenizer; needs a ``vocab.txt`` (none exists in the offline build image)."""
from __future__ import annotations

import ctypes

import numpy as np

from . import _native as N


class WordPieceTokenizer:
    def __init__(self, vocab_path: str, do_lower_case: bool = True):
        self._lib = N.lib()
        h = ctypes.c_void_p()
        N.check(self._lib.rmu_tok_create(ctypes.byref(h), str(vocab_path).encode(), 1 if do_lower_case else 0), "rmu_tok_create")
        self._h = h
        self.vocab_size = int(self._lib.rmu_tok_vocab_size(h))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.rmu_tok_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def encode(self, texts_a: list[str], texts_b: list[str] | None = None, max_len: int = 256):
        """-> ids [n, max_len] int32, type_ids [n, max_len] int32, lens [n] int32 (numpy, host)."""
        n = len(texts_a)
        if texts_b is not None and len(texts_b) != n:
            raise ValueError("texts_b must match texts_a")
        ids = np.empty((n, max_len), dtype=np.int32)
        tt = np.empty((n, max_len), dtype=np.int32)
        lens = np.empty((n,), dtype=np.int32)
        if n == 0:
            return ids, tt, lens
        # one NUL-separated UTF-8 blob per side (a join + an encode, both C-speed) instead of n bytes objects and a ctypes pointer
        # array; a text that itself contains NUL falls back to the pointer form
        blob_a = ("\0".join(texts_a) + "\0").encode("utf-8")
        blob_b = None if texts_b is None else ("\0".join(texts_b) + "\0").encode("utf-8")
        if blob_a.count(b"\0") == n and (blob_b is None or blob_b.count(b"\0") == n):
            N.check(self._lib.rmu_tok_encode_blob(self._h, blob_a, len(blob_a), blob_b, 0 if blob_b is None else len(blob_b), n, int(max_len),
                                                  ids.ctypes.data, tt.ctypes.data, lens.ctypes.data), "rmu_tok_encode_blob")
            return ids, tt, lens
        arr_a = (ctypes.c_char_p * n)(*[t.encode("utf-8") for t in texts_a])
        arr_b = None if texts_b is None else (ctypes.c_char_p * n)(*[t.encode("utf-8") for t in texts_b])
        N.check(self._lib.rmu_tok_encode(self._h, arr_a, arr_b, n, int(max_len), ids.ctypes.data, tt.ctypes.data,
                                         lens.ctypes.data), "rmu_tok_encode")
        return ids, tt, lens

    # HF-style call used by MI355XEmbeddings / MI355XCrossEncoder
    def __call__(self, a, b=None, truncation=True, max_length: int = 256, padding=False, add_special_tokens=True,
                 return_token_type_ids=True, **kw):
        single = isinstance(a, str)
        ta = [a] if single else list(a)
        tb = None if b is None else ([b] if isinstance(b, str) else list(b))
        ids, tt, lens = self.encode(ta, tb, max_len=max_length)
        out = {"input_ids": [ids[i, :lens[i]].tolist() for i in range(len(ta))],
               "token_type_ids": [tt[i, :lens[i]].tolist() for i in range(len(ta))]}
        return out

"""MI355XVectorStore -- the VectorStore the reference builds at server/RAGHelper.py:385-404 and queries at
:497-499, backed by the HBM-resident flat index (librmu.so).

Surface actually touched by the reference, all provided here:
  * ``MI355XVectorStore.from_documents([], embedding, drop_old=..., connection_args={"uri":...},
    collection_name=...)``  (Milvus form, RAGHelper.py:388-394) and
    ``MI355XVectorStore(embeddings=..., collection_name=..., connection=..., use_jsonb=True)``
    (PGVector form, :399-404);
  * ``add_documents(documents, ids=ids) -> ids``  (:431, :525);
  * ``as_retriever(search_type="mmr", search_kwargs={"k": K})`` -> object with ``invoke(str)`` that
    composes with ``|`` (:497-499, RAGHelper_local.py:158,256);
  * returned ``Document.metadata`` carries ``source``, ``id`` and ``pk`` (server.py:278-281);
  * ``delete(...)`` incl. the ``source == "<path>"`` expression server.py:373-377 sends to Milvus.

Scores: by default (``metric="ip"``) the index ranks by inner product (embeddings are unit-norm, as
sentence-transformers' Normalize module makes them) and ``score_mode`` converts to what the replaced store would
report: "l2" -> 2 - 2*ip (Milvus metric_type "L2", smaller = better), "cosine_distance" -> 1 - ip (pgvector `<=>`),
"ip" -> raw.  For embedding models that do NOT normalise, ``metric="l2"`` ranks by the native squared-L2 distance
(`RMU_METRIC_L2SQ`, Milvus' default metric on raw vectors) and ``metric="cosine"`` by cosine (pgvector `<=>`).

The class derives from LangChain's `VectorStore` and its retriever from `VectorStoreRetriever` whenever LangChain is
installed (ragmeup_amd._lc), so `EnsembleRetriever(retrievers=[bm25, retriever])` (server/RAGHelper.py:501-503) and
LCEL dict coercion (server/RAGHelper_local.py:254-258) accept them.  MMR restates langchain_core's maximal_marginal_relevance (fetch_k=20, lambda_mult=0.5,
strict '>' so the lowest index wins ties) on the fetch_k vectors gathered from HBM.
"""
from __future__ import annotations

import atexit
import json
import os
import re
import sys
import threading
import time
import weakref
from typing import Any, Iterable, Optional

import numpy as np

from . import _log
from . import _native as N
from ._lc import Document, VectorStore, VectorStoreRetriever
from .index import FlatIndex


def _cosine_similarity(x: np.ndarray, y: np.ndarray) -> np.ndarray:
    """dot(X, Y^T) / outer(|X|, |Y|), NaN/inf -> 0 -- the float64 expression the replaced MMR uses.  Kept in
    exactly this form: when the query equals a stored vector every second-pick score is 0 in exact
    arithmetic and rounding noise picks the winner, so an algebraically equivalent shortcut diverges."""
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    xn = np.linalg.norm(x, axis=1)
    yn = np.linalg.norm(y, axis=1)
    with np.errstate(divide="ignore", invalid="ignore"):
        sim = np.dot(x, y.T) / np.outer(xn, yn)
    sim[np.isnan(sim) | np.isinf(sim)] = 0.0
    return sim


def maximal_marginal_relevance(query_vec: np.ndarray, cand: np.ndarray, k: int = 4,
                               lambda_mult: float = 0.5) -> list[int]:
    """Greedy MMR over `cand` [n, d] (cosine similarities, float64), as the reference's dense retriever
    applies it (server/RAGHelper.py:497-499, search_type="mmr" -> langchain_core maximal_marginal_relevance):
    first pick = argmax sim to the query; then repeatedly the candidate maximising
    lambda*sim_q - (1-lambda)*max(sim to picked), strict '>' so the lowest index wins ties."""
    cand = np.asarray(cand)
    n = int(cand.shape[0])
    if min(k, n) <= 0:
        return []
    sim_q = _cosine_similarity(np.asarray(query_vec).reshape(1, -1), cand)[0]
    picked = [int(np.argmax(sim_q))]
    selected = np.array([cand[picked[0]]])
    while len(picked) < min(k, n):
        best, best_i = -np.inf, -1
        sim_sel = _cosine_similarity(cand, selected)
        for i in range(n):
            if i in picked:
                continue
            val = lambda_mult * sim_q[i] - (1.0 - lambda_mult) * max(sim_sel[i])
            if val > best:
                best, best_i = val, i
        picked.append(best_i)
        selected = np.append(selected, [cand[best_i]], axis=0)
    return picked


class MI355XRetriever(VectorStoreRetriever):
    """What ``as_retriever`` returns: ``invoke(query) -> list[Document]`` (a LangChain `VectorStoreRetriever` with the
    fields ``vectorstore``, ``search_type``, ``search_kwargs``), plus ``batch_invoke``."""

    def _get_relevant_documents(self, query: str, *, run_manager: Any = None, **kw) -> list[Document]:
        if self.search_type == "mmr":
            return self.vectorstore.max_marginal_relevance_search(query, **self.search_kwargs)
        if self.search_type == "similarity_score_threshold":
            thr = self.search_kwargs.get("score_threshold")
            kw2 = {k: v for k, v in self.search_kwargs.items() if k != "score_threshold"}
            pairs = self.vectorstore.similarity_search_with_relevance_scores(query, **kw2)
            # a native-L2 store ranks by DISTANCE (smaller = better): the threshold is an upper bound there
            if getattr(self.vectorstore, "metric", "ip") == "l2":
                return [d for d, s in pairs if thr is None or s <= thr]
            return [d for d, s in pairs if thr is None or s >= thr]
        return self.vectorstore.similarity_search(query, **self.search_kwargs)

    def batch_invoke(self, queries: list[str]) -> list[list[Document]]:
        """New capability beside the LangChain API: one fused scan for a whole query batch."""
        k = int(self.search_kwargs.get("k", 4))
        if self.search_type == "mmr":
            return self.vectorstore.max_marginal_relevance_search_batch(queries, **self.search_kwargs)
        return [[d for d, _ in row] for row in self.vectorstore.similarity_search_with_score_batch(queries, k=k)]


_METRICS = {"ip": N.METRIC_IP, "cosine": N.METRIC_COSINE, "l2": N.METRIC_L2SQ}


def _json_default(o):
    """Metadata values loaders produce that JSON does not know (numpy scalars, datetimes, bytes, Paths): stored as plain
    numbers where they are numbers, else as their string form -- persist() must not fail on them."""
    if isinstance(o, np.integer):
        return int(o)
    if isinstance(o, np.floating):
        return float(o)
    if isinstance(o, np.ndarray):
        return o.tolist()
    if isinstance(o, (bytes, bytearray)):
        return o.decode("utf-8", "replace")
    return str(o)


class MI355XVectorStore(VectorStore):
    # drop_old=False re-attaches to a live collection; weak values: the registry does not keep a 15 GB index alive
    _collections: "weakref.WeakValueDictionary[str, MI355XVectorStore]" = weakref.WeakValueDictionary()
    _collections_lock = threading.Lock()

    def __init__(self, embeddings: Any = None, collection_name: str = "LangChainCollection", connection: Any = None,
                 use_jsonb: bool = True, *, embedding_function: Any = None, connection_args: dict | None = None,
                 drop_old: bool = False, score_mode: str | None = None, metric: str = "ip", dim: int | None = None,
                 device: int | None = None, auto_persist: bool | str = "atexit", pipeline_inserts: bool | str = "auto",
                 pipeline_window: float = 0.25):
        self._embeddings = embeddings if embeddings is not None else embedding_function
        if self._embeddings is None:
            raise ValueError("an Embeddings object is required")
        self.collection_name = collection_name
        self.connection = connection if connection is not None else (connection_args or {}).get("uri")
        if metric not in _METRICS:
            raise ValueError("metric must be 'ip', 'cosine' or 'l2'")
        self.metric = metric
        if score_mode is None:   # native metrics report their own number; IP on unit vectors imitates Milvus "L2"
            score_mode = {"ip": "l2", "cosine": "cosine_distance", "l2": "raw"}[metric]
        if score_mode not in ("l2", "cosine_distance", "ip", "raw"):
            raise ValueError("score_mode must be 'l2', 'cosine_distance', 'ip' or 'raw'")
        self.score_mode = score_mode
        self._device = device
        self._dim = dim
        # Milvus-Lite writes through to its file.  True: rewrite the files after every add/delete (O(N) each);
        # "atexit" (default): write once at interpreter exit if anything changed; False: only on persist()
        self.auto_persist = auto_persist
        self._dirty = False
        self._superseded = False             # a newer store took this (uri, collection): this one never writes the files again
        self._index: FlatIndex | None = None
        self._lock = threading.RLock()       # writer lock (add/delete); searches take the C-side shared lock
        self._texts: list[str] = []
        self._metas: list[dict] = []
        self._pks: list[str] = []
        self._alive: list[bool] = []
        self._pk_to_row: dict[str, int] = {}
        # Cross-call insert pipeline (see _add_pipelined).  "auto" (default): a call's GPU half is deferred only INSIDE an insert loop --
        # another add_texts ended less than `pipeline_window` seconds ago or halves are still pending -- so a single upload (the
        # reference's POST /add_document -> _add_to_vector_database, server/RAGHelper.py:518-538) is synchronous and a failure is raised
        # in the call that caused it; True: every call of 128..pipeline_block texts is deferred (round 4's behaviour); False: never.
        if pipeline_inserts not in (True, False, "auto"):
            raise ValueError("pipeline_inserts must be True, False or 'auto'")
        self.pipeline_inserts = pipeline_inserts
        self.pipeline_window = float(pipeline_window)
        self._last_add_end = float("-inf")   # time.monotonic() at the end of the latest add_texts
        self._pending = []                   # GPU halves of add_texts calls still in flight, oldest first (see _add_pipelined)
        self._pipe_failed = False            # set by the worker when a half fails: the halves queued behind it do nothing
        self._work = []                      # GPU halves not yet taken by the worker (tok, n0, cnt, stale, future), oldest first
        self._wlock = threading.Lock()
        self._worker = None
        if drop_old:
            self._remove_persisted()
        if auto_persist == "atexit" and self._persist_paths():
            ref = weakref.ref(self)
            atexit.register(lambda: (lambda s: s is not None and s._dirty and not s._superseded and s._persist_quietly())(ref()))

    def __del__(self):
        # a dirty "atexit" store that is collected BEFORE interpreter exit would otherwise never be written -- unless a newer
        # store replaced it (from_documents(drop_old=True) / a re-created collection): that one owns the files now
        try:
            if (self.auto_persist == "atexit" and self._dirty and not self._superseded and self._index is not None
                    and self._persist_paths()):
                self._persist_quietly()
        except Exception:   # noqa: BLE001 - partially constructed object / interpreter teardown
            pass

    @property
    def embeddings(self):
        """The `Embeddings` object (a read-only property on LangChain's VectorStore)."""
        return self._embeddings

    # ---- construction as the reference does it (RAGHelper.py:388-394) --------------------------------
    @classmethod
    def from_documents(cls, documents: list[Document], embedding: Any, drop_old: bool = False,
                       connection_args: dict | None = None, collection_name: str = "LangChainCollection",
                       ids: list[str] | None = None, **kw) -> "MI355XVectorStore":
        key = f"{(connection_args or {}).get('uri')}::{collection_name}"
        with cls._collections_lock:
            old = cls._collections.get(key)
            store = None if drop_old else old
            if store is None:
                if old is not None:
                    old._superseded = True       # its finalizer / atexit hook must not rewrite the files the new store owns
                store = cls(embeddings=embedding, collection_name=collection_name, connection_args=connection_args,
                            drop_old=drop_old, **kw)
                cls._collections[key] = store
                # vector_store_initial_load=False: re-open what an earlier run persisted (RAGHelper.py:391, :417)
                paths = store._persist_paths()
                if not drop_old and paths and all(os.path.exists(p) for p in paths):
                    store.load()
                elif not drop_old and paths and os.path.exists(paths[0]) and os.path.exists(paths[1][:-len(".json")] + ".pkl"):
                    # a store written before the JSON format: re-opening it EMPTY would silently drop the collection
                    cls._collections.pop(key, None)
                    raise RuntimeError(f"{paths[1][:-len('.json')]}.pkl is a pre-JSON metadata file this version does not read: "
                                       f"re-index (vector_store_initial_load=True) or convert it to {paths[1]}")
        if documents:
            store.add_documents(documents, ids=ids)
            store.flush()                    # a constructor-style call: the rows are in the index (or the failure raised) when it returns
        return store

    @classmethod
    def from_texts(cls, texts: list[str], embedding: Any, metadatas: Optional[list[dict]] = None,
                   ids: Optional[list[str]] = None, **kw) -> "MI355XVectorStore":
        store = cls.from_documents([], embedding, **kw)
        if texts:
            store.add_texts(texts, metadatas, ids=ids)
        return store

    # ---- persistence: <uri>.<collection>.rmu (corpus matrix) + .meta.json (texts, metadata, pks) ------------
    def _persist_paths(self) -> tuple[str, str] | None:
        if not self.connection or not isinstance(self.connection, str) or "://" in self.connection:
            return None
        base = f"{self.connection}.{self.collection_name}"
        return base + ".rmu", base + ".meta.json"

    def _remove_persisted(self):
        for p in self._persist_paths() or ():
            for q in (p, p + ".tmp"):
                if os.path.exists(q):
                    os.remove(q)

    def persist(self) -> bool:
        """Both files are written to temporaries and renamed into place (a crash leaves the previous pair)."""
        paths = self._persist_paths()
        if paths is None or self._index is None or self._superseded:
            return False
        with self._lock:
            self._drain()
            self._index.save(paths[0] + ".tmp")
            with open(paths[1] + ".tmp", "w", encoding="utf-8") as f:
                json.dump({"format": 1, "n": len(self._texts), "dim": self._dim, "metric": self.metric,
                           "score_mode": self.score_mode, "texts": self._texts, "metas": self._metas, "pks": self._pks,
                           "alive": [1 if a else 0 for a in self._alive]}, f, default=_json_default)
            os.replace(paths[0] + ".tmp", paths[0])
            os.replace(paths[1] + ".tmp", paths[1])
            self._dirty = False
        return True

    def _persist_quietly(self):
        """The atexit / finalizer hook: a failure cannot propagate at interpreter shutdown, so it is REPORTED (stderr), the
        half-written temporaries are removed and the previous pair of files stays as it was."""
        try:
            self.persist()
        except Exception as e:   # noqa: BLE001
            import sys
            _log.get_logger().error(f"ragmeup_amd: persisting collection {self.collection_name!r} failed: {type(e).__name__}: {e}")
            for p in self._persist_paths() or ():
                try:
                    if os.path.exists(p + ".tmp"):
                        os.remove(p + ".tmp")
                except OSError:
                    pass

    def load(self) -> bool:
        paths = self._persist_paths()
        if paths is None:
            return False
        with self._lock:
            with open(paths[1], "r", encoding="utf-8") as f:
                m = json.load(f)
            index = self._open_index(paths[0])
            n = len(m["texts"])
            if not (len(m["metas"]) == len(m["pks"]) == len(m["alive"]) == n == int(m.get("n", n)) == len(index)):
                raise ValueError(f"{paths[1]} does not describe {paths[0]} ({n} records vs {len(index)} rows)")
            want = _METRICS[m.get("metric", "ip")]
            if getattr(index, "metric", want) != want:
                raise ValueError(f"{paths[0]}: metric differs from {paths[1]}")
            self._index = index
            self._texts, self._metas, self._pks = m["texts"], m["metas"], m["pks"]
            self._alive = [bool(a) for a in m["alive"]]
            self._dim = m["dim"]
            self.metric = m.get("metric", "ip")
            self.score_mode = m.get("score_mode", self.score_mode)
            self._pk_to_row = {pk: r for r, pk in enumerate(self._pks) if self._alive[r]}
            self._dirty = False
        return True

    # ---- helpers ----------------------------------------------------------------------------------------
    def _new_index(self, dim: int):
        """The HBM-resident flat index (librmu.so); raises when the library is missing -- there is no CPU index."""
        return FlatIndex(dim, _METRICS[self.metric], device=self._device)

    def _open_index(self, path: str):
        return FlatIndex.load(path, device=self._device)

    def _ensure_index(self, dim: int):
        if self._index is None:
            self._dim = dim
            self._index = self._new_index(dim)
        elif dim != self._dim:
            raise ValueError(f"embedding dimension changed: {self._dim} -> {dim}")

    def _embed_docs_for_index(self, texts: list[str]):
        """Embeddings for insertion: a torch CUDA tensor when the Embeddings object can produce one (ours: the rows go
        device-to-device into the corpus, no host round trip), else a numpy array."""
        if hasattr(self._embeddings, "embed_documents_device"):
            return self._embeddings.embed_documents_device(texts)
        return self._embed_docs(texts)

    def _embed_docs(self, texts: list[str]) -> np.ndarray:
        if hasattr(self._embeddings, "embed_documents_array"):     # our Embeddings: no Python float lists
            return np.asarray(self._embeddings.embed_documents_array(texts), dtype=np.float32)
        return np.asarray(self._embeddings.embed_documents(texts), dtype=np.float32)

    def _embed_query(self, text: str) -> np.ndarray:
        if hasattr(self._embeddings, "embed_query_array"):         # our Embeddings: no Python float list in between
            return np.asarray(self._embeddings.embed_query_array(text), dtype=np.float32)
        return np.asarray(self._embeddings.embed_query(text), dtype=np.float32)

    def _doc(self, row: int) -> Document:
        md = dict(self._metas[row])
        md["pk"] = self._pks[row]
        return Document(page_content=self._texts[row], metadata=md)

    def _convert(self, s: float) -> float:
        if self.score_mode == "l2":
            return float(2.0 - 2.0 * s)
        if self.score_mode == "cosine_distance":
            return float(1.0 - s)
        return float(s)

    def __len__(self) -> int:
        if self._pending:
            self.flush()
        return sum(self._alive)

    # ---- the reference's insert loop, pipelined ACROSS calls ---------------------------------------------------------
    # server/RAGHelper.py:423-434 inserts in 1000-document calls; each call is tokenise (host, ~1 ms) -> forward + append (GPU, ~3.4 ms)
    # -> Python bookkeeping, strictly in series: 0.81 of the rate of one big call (whose blocks overlap inside the call).  A call of
    # that size therefore returns once its HOST half is done -- ids fixed, records appended, the row numbers it will occupy known --
    # and hands the GPU half to a worker thread; the next call's tokenising overlaps it.  Up to `pipeline_depth` halves are queued, and the
    # worker runs EVERYTHING that is queued when it becomes free as ONE forward + ONE append: a 1000-chunk forward is ~14 % slower per chunk
    # than an 8192-chunk one, which is what held the pattern at 0.86 of the one-call rate however the hand-over was arranged (depth 1 and 2
    # measured alike); the host half of a call (~1.5 ms) is shorter than its GPU half (~3.4 ms), so the queue fills and the forwards grow.
    # Nothing can observe the difference: every entry point that reads or changes the index (search, delete, persist, flush()) first
    # waits for the pending halves, and a failure rolls back that call's records AND those of the calls queued with or behind it, and is
    # raised there.
    # insert calls whose GPU half may be pending (1 = no coalescing).  RMU_ADD_DEPTH is a tuning switch like librmu's: honoured only with
    # RMU_TUNING=1, and a value that is not a number is ignored (it must not break the import)
    pipeline_depth = 8
    if os.environ.get("RMU_TUNING") == "1" and os.environ.get("RMU_ADD_DEPTH"):
        try:
            pipeline_depth = max(1, int(os.environ["RMU_ADD_DEPTH"]))
        except ValueError:
            import warnings
            warnings.warn(f"RMU_ADD_DEPTH={os.environ['RMU_ADD_DEPTH']!r} is not an integer: ignored")

    def _drain(self, keep: int = 0):
        """(under self._lock) wait for pending GPU halves, oldest first, until at most `keep` are left; on failure undo and re-raise."""
        from concurrent.futures import wait as _wait
        while len(self._pending) > keep:
            # An interrupt of THIS wait (KeyboardInterrupt / SystemExit in the waiting thread) is not a failure of the GPU half: it
            # propagates with the records untouched and the half still pending.  Whether the half failed is read off the FINISHED future.
            _wait([self._pending[0][0]])
            exc = self._pending[0][0].exception()
            if exc is not None:
                ents, self._pending = self._pending, []
                _wait([ent[0] for ent in ents[1:]])      # queued behind the failure: skipped by the worker (their futures carry that)
                self._pipe_failed = False
                for (_f, n02, _c, undo2, stale2) in reversed(ents[1:]):   # newest first: a pk may appear in several calls
                    del self._texts[n02:], self._metas[n02:], self._pks[n02:], self._alive[n02:]
                    self._undo_pks(undo2, stale2)
                self._rollback_failed_half(ents[0], exc)
                raise AssertionError("unreachable: _rollback_failed_half raises")
            self._pending.pop(0)

    def _rollback_failed_half(self, ent, exc):
        _fut, n0, cnt, undo, stale = ent
        try:
            raise exc
        except _RowsOutOfStep as e:
            # the rows ARE in the index (tombstoned by the worker): dead placeholder records keep row numbers and records in step
            for r in range(min(n0, e.first), len(self._alive)):
                self._alive[r] = False
            while len(self._texts) < e.first + e.total:
                self._texts.append(""); self._metas.append({}); self._pks.append(""); self._alive.append(False)
            self._undo_pks(undo, stale)
            raise RuntimeError(f"index rows ({e.first}) and host records ({n0}) out of step: the batch was rolled back") from None
        except BaseException:            # whatever the worker caught (it re-raises nothing itself: the exception lives in the future)
            del self._texts[n0:], self._metas[n0:], self._pks[n0:], self._alive[n0:]
            self._undo_pks(undo, stale)
            raise

    def _undo_pks(self, undo, stale):
        for pk, old in undo:
            if old is None:
                self._pk_to_row.pop(pk, None)
            else:
                self._pk_to_row[pk] = old
        for r in stale:
            self._alive[r] = True

    def flush(self):
        """Wait until everything added so far is in the HBM-resident index (and raise what a pending add failed with)."""
        with self._lock:
            self._drain()

    def _can_pipeline(self) -> bool:
        """The GPU halves call the index from a worker thread: only the native index (and the stock factory methods) are known to allow it."""
        return not ((self._index is not None and not hasattr(self._index, "_h")) or type(self)._new_index is not MI355XVectorStore._new_index)

    def _gpu_pump(self):
        """(worker thread) run the queued GPU halves -- as many as fit one pipeline block -- as one forward + one append each round; resolve
        their futures.  (Every call submits a pump; one that finds the queue empty returns: an earlier pump took its item along.)
        (round 5) With the native encoder a round's forward is ENQUEUED (the calls' ids written straight into a pinned staging slot, forward
        left in flight on the encoder's stream); when it ends the NEXT round is uploaded and enqueued first, and only then are this round's
        rows appended and its callers released -- the append and the Python around it ran between two forwards before (a median 1.9 ms of
        idle device per ~21 ms forward in the reference's 1000-document loop)."""
        emb = self._embeddings
        cap = int(getattr(emb, "pipeline_block", 0)) or 8192

        def take():
            with self._wlock:
                items, total = [], 0
                while self._work and (not items or total + self._work[0][2] <= cap):
                    items.append(self._work.pop(0))
                    total += items[-1][2]
            return items, total

        def fail(items, e):
            for it in items:
                it[4].set_exception(e)

        rounds = 0
        cur = nxt = None
        try:
            while True:
                if cur is None:
                    items, total = take()
                    if not items:
                        return
                    cur = self._start_round(items, total, rounds)
                    rounds += 1
                    if cur is None:
                        continue
                nxt = None
                held = None
                if cur[2] is not None:
                    # the forward is in flight.  When it ENDS, whatever has queued up meanwhile goes straight behind it -- before this round's
                    # rows are appended and its callers released (taking the next round any earlier would cut the queue short: a round is as
                    # large as the calls that arrived during the previous forward, which is what keeps the forwards at pipeline-block size)
                    cur[2][1].synchronize()
                    items2, total2 = take()
                    if items2:
                        # a failure to START the next round must not skip the healthy round in front of it (its forward has finished): the
                        # failure is recorded only after `cur` is appended, so only rounds BEHIND the failing one are rolled back
                        nxt = self._start_round(items2, total2, rounds, defer_failure=True)
                        rounds += 1
                        if isinstance(nxt, _DeferredFailure):
                            held, nxt = nxt, None
                self._finish_round(*cur)
                cur = None
                if held is not None:
                    self._round_failed(held.items, held.error)
                cur = nxt
        except BaseException as e:   # noqa: BLE001 - e.g. an asynchronous HIP error surfacing at the event wait: nobody may be left waiting
            stuck = [r for r in (cur, nxt) if r is not None and not isinstance(r, _DeferredFailure)]
            with self._wlock:
                rest, self._work[:] = list(self._work), []
            for r in stuck:
                self._round_failed([it for it in r[0] if not it[4].done()], e)
            if rest:
                self._round_failed(rest, e)
            if not stuck and not rest:
                self._pipe_failed = True

    def _start_round(self, items, total, round_no, defer_failure: bool = False):
        """-> (items, (ids, lens), in-flight handle | None), or None when the round was refused / failed before it started
        (defer_failure: a failure comes back as _DeferredFailure instead of being recorded -- the pump records it once the round in front is in)"""
        import contextlib
        import numpy as np
        emb = self._embeddings
        if self._pipe_failed:
            for it in items:
                it[4].set_exception(RuntimeError("skipped: an earlier insert call of the pipeline failed"))
            return None
        try:
            handle = None
            if hasattr(emb, "enqueue_token_arrays"):     # the calls' id arrays go into the staging buffer one by one: no concatenated copy
                dev = getattr(emb.encoder, "device", None)
                if dev is not None:
                    import torch
                with (torch.cuda.device(dev) if dev is not None else contextlib.nullcontext()):
                    # room for this round's rows NOW, while nothing is in flight: a re-allocation inside the append (behind the next round's
                    # forward) waits for that forward -- 12 re-allocations x ~22 ms on the way to 1M rows
                    if hasattr(self._index, "reserve"):
                        self._index.reserve(items[0][1] + total)
                    handle = emb.enqueue_token_arrays([it[0] for it in items], round_no)
            if handle is not None:
                return items, None, handle
            if len(items) == 1:
                ids, lens = items[0][0]
            else:
                width = max(it[0][0].shape[1] for it in items)
                ids = np.zeros((total, width), dtype=items[0][0][0].dtype)
                lo = 0
                for it in items:
                    a = it[0][0]
                    ids[lo:lo + a.shape[0], :a.shape[1]] = a
                    lo += a.shape[0]
                lens = np.concatenate([it[0][1] for it in items])
            return items, (ids, lens), None
        except BaseException as e:
            if defer_failure:
                return _DeferredFailure(items, e)
            self._round_failed(items, e)
            return None

    def _round_failed(self, items, e):
        self._pipe_failed = True
        # the futures FIRST: an injected logger that raises (the reference's, factory.from_env(logger=)) must not leave a caller waiting
        for it in items:
            if not it[4].done():
                it[4].set_exception(e)
        # said where it happens: with deferred halves the RuntimeError itself is raised by whichever call touches the store next
        try:
            _log.get_logger().error(f"ragmeup_amd: a deferred insert of collection {self.collection_name!r} failed ({type(e).__name__}: {e}); "
                                    f"{sum(it[2] for it in items)} records of {len(items)} add_texts call(s) and everything queued behind them are rolled back")
        except Exception:   # noqa: BLE001 - a logging failure is not the insert's failure
            pass

    def _finish_round(self, items, tok, handle):
        import contextlib
        emb = self._embeddings
        total = sum(it[2] for it in items)
        try:
            dev = getattr(emb.encoder, "device", None)
            if dev is not None:
                import torch
            with (torch.cuda.device(dev) if dev is not None else contextlib.nullcontext()):
                if handle is not None:
                    vecs, done = handle
                    done.synchronize()                   # (also when the round is about to be skipped: nothing of it stays in flight)
                if self._pipe_failed:                    # the round in front of this one failed while this one's forward was already queued
                    for it in items:
                        it[4].set_exception(RuntimeError("skipped: an earlier insert call of the pipeline failed"))
                    return
                if handle is None:
                    vecs = emb.embed_token_arrays_device(*tok)
                n0 = items[0][1]
                first = self._index.add(vecs)
                if first != n0:
                    self._index.remove_rows(list(range(min(n0, first), first + total)))
                    raise _RowsOutOfStep(first, total)
                try:
                    stale = [r for it in items for r in it[3]]
                    if stale:
                        self._index.remove_rows(stale)
                except Exception as e2:
                    # the rows ARE in the index: a plain rollback would delete the records and leave them live (a search could return a
                    # row without a record).  Tombstone them and report "out of step": placeholder records keep rows and records aligned.
                    try:
                        self._index.remove_rows(list(range(first, first + total)))
                    except Exception:   # noqa: BLE001 - the index is beyond repair from here; the caller still learns of e2
                        pass
                    raise _RowsOutOfStep(first, total) from e2
        except BaseException as e:
            self._round_failed(items, e)
            return
        for it in items:
            it[4].set_result(None)

    def _add_pipelined(self, sel_texts, sel_ids, sel_metas_fn, force: bool = False) -> bool:
        emb = self._embeddings
        if (self.pipeline_inserts is False or self.auto_persist is True or not hasattr(emb, "tokenize_for_index")
                or not (128 <= len(sel_texts) <= getattr(emb, "pipeline_block", 0)) or not self._can_pipeline()):
            return False
        if not force and self.pipeline_inserts == "auto" and not self._pending and time.monotonic() - self._last_add_end > self.pipeline_window:
            return False                                 # not inside an insert loop: this call is synchronous and raises its own failures
        # (force: a block of one big call -- the call itself waits for its blocks and raises their failures)
        tok = emb.tokenize_for_index(sel_texts)          # the previous call's GPU half may still be running: this is the overlap
        if tok is None:
            return False
        sel_metas = sel_metas_fn()
        with self._lock:
            self._drain(keep=self.pipeline_depth - 1)    # the newest half may still be running while this call's records are written
            self._ensure_index(int(emb.encoder.HIDDEN))
            n0 = len(self._texts)
            self._texts.extend(sel_texts)
            self._metas.extend(sel_metas)
            self._pks.extend(sel_ids)
            self._alive.extend([True] * len(sel_ids))
            old_rows = self._pk_to_row
            stale = [old_rows[pk] for pk in sel_ids if pk in old_rows and self._alive[old_rows[pk]]] if old_rows else []
            undo = [(pk, old_rows.get(pk)) for pk in sel_ids] if old_rows else [(pk, None) for pk in sel_ids]
            for r in stale:
                self._alive[r] = False
            old_rows.update(zip(sel_ids, range(n0, n0 + len(sel_ids))))
            self._dirty = True
            from concurrent.futures import Future, ThreadPoolExecutor
            if self._worker is None:
                self._worker = ThreadPoolExecutor(max_workers=1, thread_name_prefix="rmu-add")
            fut = Future()
            with self._wlock:
                self._work.append((tok, n0, len(sel_ids), stale, fut))
            self._pending.append((fut, n0, len(sel_ids), undo, stale))
            self._worker.submit(self._gpu_pump)
        return True

    # ---- insert (RAGHelper.py:431, :525) ------------------------------------------------------------------
    def add_texts(self, texts: Iterable[str], metadatas: Optional[list[dict]] = None, ids: Optional[list[str]] = None,
                  **kw) -> list[str]:
        return self._add(list(texts), lambda: metadatas, None if ids is None else list(ids))

    # While a big call's forward runs on the worker thread, the bookkeeping on this one must not hold the GIL for long: ONE C-level call over
    # 1M ids (set(ids): 80 ms, dict(zip(..)): 200 ms) keeps the worker from issuing the next block's forward for that long -- the GPU idles.
    # Such calls go over the ids in pieces; the interpreter hands the GIL over between them.
    _GIL_PIECE = 8192

    @classmethod
    def _all_distinct(cls, ids: list) -> bool:
        seen: set = set()
        for lo in range(0, len(ids), cls._GIL_PIECE):
            seen.update(ids[lo:lo + cls._GIL_PIECE])
        return len(seen) == len(ids)

    def _blockwise_ok(self, n: int) -> int:
        """Pipeline block size if a call of n texts can run as a sequence of pipelined block inserts (native tokenizer + encoder, the index
        callable from the worker, nothing that forbids deferring), else 0."""
        emb = self._embeddings
        blk = int(getattr(emb, "pipeline_block", 0) or 0)
        if (blk < 128 or n <= blk or self.pipeline_inserts is False or self.auto_persist is True or not hasattr(emb, "tokenize_for_index")
                or not self._can_pipeline()):
            return 0
        can = getattr(emb, "can_tokenize_for_index", None)
        return blk if (can is None or can()) else 0

    def _add_blockwise(self, n: int, texts_of, metas_of, ids, blk: int) -> list[str]:
        """One big call = the reference's insert loop (RAGHelper.py:423-434) in pipeline-block-sized steps: block i + 1 is tokenised and its
        records are written while the forward of block i runs, and the worker puts the next forward behind the current one -- measured 0.90-0.94
        of the encoder-only rate against 0.85 for "embed everything, then do the bookkeeping" (DESIGN.md 4.6).  The call returns when every
        block is in the index and raises a failed block's error itself.  Like the replaced stores' batched insert (Milvus inserts
        `batch_size` rows at a time) it is not atomic: blocks in front of a failing one stay inserted, the failing block and the ones behind
        it are rolled back.  Ids repeated across blocks follow the upsert rule (the last occurrence lives).  texts_of / metas_of(lo, hi):
        the block's texts and metadata -- taken block by block, so not even those lists are built in front of the first forward."""
        out: list[str] = []
        try:
            with self._lock:                             # the collection's final size is known: one re-allocation, before anything is in flight
                self._drain()
                self._ensure_index(int(self._embeddings.encoder.HIDDEN))
                if hasattr(self._index, "reserve"):
                    self._index.reserve(len(self._texts) + n)
            for lo in range(0, n, blk):
                hi = min(lo + blk, n)
                out += self._add(texts_of(lo, hi), (lambda lo=lo, hi=hi: metas_of(lo, hi)), None if ids is None else ids[lo:hi], _force_pipeline=True)
        finally:
            self.flush()
        return out

    def _add(self, texts: list, metas_fn, ids, _force_pipeline: bool = False) -> list[str]:
        if not texts:
            return []
        if ids is not None and len(ids) != len(texts):
            raise ValueError("texts, metadatas and ids must have equal lengths")
        blk = 0 if _force_pipeline else self._blockwise_ok(len(texts))
        if blk:
            metadatas = metas_fn()
            if metadatas is not None and len(metadatas) != len(texts):
                raise ValueError("texts, metadatas and ids must have equal lengths")
            return self._add_blockwise(len(texts), lambda lo, hi: texts[lo:hi], lambda lo, hi: None if metadatas is None else metadatas[lo:hi], ids, blk)
        # The indexing path's one big call (RAGHelper.py:423-434 with everything in one batch; BASELINE.json configs[2]): the embedding of ALL
        # texts starts before any Python bookkeeping -- metadata lists, str(ids), the duplicate-id check, the record copies, the pk map: 0.7 s
        # per 1M documents that used to sit serially in front of and behind a 2.7 s forward (tokenizer and encoder run in librmu.so with the
        # GIL released).  Ids repeated inside the batch (rare: md5 ids of distinct chunks) drop their earlier rows from the result afterwards.
        early = pool = None
        emb = self._embeddings
        if not _force_pipeline and len(texts) > max(4096, int(getattr(emb, "pipeline_block", 0) or 0)):
            from concurrent.futures import ThreadPoolExecutor
            pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix="rmu-embed")
            early = pool.submit(self._embed_docs_for_index, texts)
        try:
            metadatas = metas_fn() or [{} for _ in texts]
            if ids is None:
                import uuid
                ids = [str(uuid.uuid4()) for _ in texts]
            ids = [str(i) for i in ids]
            if len(ids) != len(texts) or len(metadatas) != len(texts):
                raise ValueError("texts, metadatas and ids must have equal lengths")
            # upsert semantics of the replaced stores: one row per pk -- inside a batch the LAST occurrence wins
            if self._all_distinct(ids):                  # the usual case (md5 ids of distinct chunks): nothing to drop
                keep = range(len(ids))
                sel_texts, sel_ids = texts, ids
            else:
                keep = sorted({pk: i for i, pk in enumerate(ids)}.values())
                sel_texts, sel_ids = [texts[i] for i in keep], [ids[i] for i in keep]
            return self._add_texts_body(texts, metadatas, ids, keep, sel_texts, sel_ids, early, _force_pipeline)
        finally:
            if pool is not None:
                pool.shutdown(wait=True)                 # (a failed validation above still waits for the forward it started)
            self._last_add_end = time.monotonic()

    def _add_texts_body(self, texts, metadatas, ids, keep, sel_texts, sel_ids, early=None, force_pipeline=False) -> list[str]:
        new_map = None
        if early is not None:                            # the forward of ALL texts is already running (see _add)
            sel_metas = [dict(metadatas[i]) for i in keep]
            n0_guess = len(self._texts)                  # (read without the lock: only used if it still holds under it)
            new_map = {}
            for lo in range(0, len(sel_ids), self._GIL_PIECE):     # (in pieces: see _all_distinct)
                new_map.update(zip(sel_ids[lo:lo + self._GIL_PIECE], range(n0_guess + lo, n0_guess + len(sel_ids))))
            vecs = early.result()
            if len(sel_ids) != len(texts):
                vecs = vecs[list(keep)]                  # ids repeated inside the batch: the last occurrence's row stays
        elif self._add_pipelined(sel_texts, sel_ids, lambda: [dict(metadatas[i]) for i in keep], force_pipeline):
            return list(ids)
        # The host records are prepared WHILE the GPU embeds (both the tokenizer and the encoder run in librmu.so with the GIL
        # released): on the indexing path (one big call) the Python bookkeeping would otherwise sit serially behind every embedding.
        elif len(sel_texts) >= 4096:
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(max_workers=1) as pool:
                fut = pool.submit(self._embed_docs_for_index, sel_texts)
                sel_metas = [dict(metadatas[i]) for i in keep]
                vecs = fut.result()
        else:
            vecs = self._embed_docs_for_index(sel_texts)
            sel_metas = [dict(metadatas[i]) for i in keep]
        with self._lock:
            self._drain()
            self._ensure_index(int(vecs.shape[1]))
            n0 = len(self._texts)
            # host records FIRST: a concurrent search may return a new row the moment index.add publishes it
            self._texts.extend(sel_texts)
            self._metas.extend(sel_metas)
            self._pks.extend(sel_ids)
            self._alive.extend([True] * len(sel_ids))
            try:
                first = self._index.add(vecs)
            except Exception:
                del self._texts[n0:], self._metas[n0:], self._pks[n0:], self._alive[n0:]
                raise
            if first != n0:
                # the rows ARE in the index now (and so is whatever made the counts differ): every index row from n0 on is
                # tombstoned and gets a dead placeholder record, so row numbers and records stay in step
                end = first + len(keep)
                self._index.remove_rows(list(range(min(n0, first), end)))
                for r in range(min(n0, first), len(self._alive)):
                    self._alive[r] = False
                while len(self._texts) < end:
                    self._texts.append(""); self._metas.append({}); self._pks.append(""); self._alive.append(False)
                raise RuntimeError(f"index rows ({first}) and host records ({n0}) out of step: the batch was rolled back")
            # the new copies are in: only now retire the rows they replace (a failed add loses nothing)
            old_rows = self._pk_to_row
            stale = [old_rows[pk] for pk in sel_ids if pk in old_rows and self._alive[old_rows[pk]]] if old_rows else []
            if stale:
                self._index.remove_rows(stale)
                for r in stale:
                    self._alive[r] = False
            if new_map is not None and n0 == n0_guess:   # built while the GPU embedded
                if old_rows:
                    old_rows.update(new_map)
                else:
                    self._pk_to_row = new_map
            else:
                old_rows.update(zip(sel_ids, range(n0, n0 + len(sel_ids))))
            self._dirty = True
            if self.auto_persist is True:
                self.persist()
        return list(ids)

    def add_documents(self, documents: list[Document], ids: Optional[list[str]] = None, **kw) -> list[str]:
        documents = documents if isinstance(documents, list) else list(documents)
        ids = None if ids is None else list(ids)
        blk = self._blockwise_ok(len(documents))
        if blk:
            if ids is not None and len(ids) != len(documents):
                raise ValueError("texts, metadatas and ids must have equal lengths")
            return self._add_blockwise(len(documents), lambda lo, hi: [d.page_content for d in documents[lo:hi]],
                                       lambda lo, hi: [d.metadata for d in documents[lo:hi]], ids, blk)
        return self._add([d.page_content for d in documents], lambda: [d.metadata for d in documents], ids)

    # ---- delete (server.py:373-377) -----------------------------------------------------------------------
    _EXPR = re.compile(r"""^\s*(\w+)\s*==\s*(['"])(.*)\2\s*$""")

    def delete(self, ids: Optional[list[str]] = None, expr: Optional[str] = None, filter: Optional[dict] = None, **kw):
        """Delete by pk list, by a Milvus-style `field == "value"` expression, or by a metadata dict.
        Returns an object with ``delete_count`` (what server.py:385 reads) that is also truthy/int-like."""
        with self._lock:
            self._drain()
            rows: list[int] = []
            if ids:
                rows += [self._pk_to_row[i] for i in ids if i in self._pk_to_row]
            cond = dict(filter or {})
            if expr:
                m = self._EXPR.match(expr)
                if not m:
                    raise ValueError(f"unsupported delete expression: {expr!r}")
                cond[m.group(1)] = m.group(3)
            if cond:
                for r, md in enumerate(self._metas):
                    if self._alive[r] and all(
                            (self._pks[r] if k == "pk" else md.get(k)) == v for k, v in cond.items()):
                        rows.append(r)
            rows = sorted({r for r in rows if self._alive[r]})
            if rows and self._index is not None:
                self._index.remove_rows(rows)
            for r in rows:
                self._alive[r] = False
            if rows:
                self._dirty = True
                if self.auto_persist is True:
                    self.persist()
        return _DeleteResult(len(rows))

    # ---- search ---------------------------------------------------------------------------------------------
    def _search_vecs(self, qvecs: np.ndarray, k: int):
        if self._pending:
            self.flush()
        if self._index is None or len(self._index) == 0:
            return np.full((qvecs.shape[0], 0), -np.inf, np.float32), np.full((qvecs.shape[0], 0), -1, np.int64)
        kk = max(1, min(int(k), N.MAX_K))
        return self._index.search(qvecs, kk)

    def similarity_search_with_score_by_vector(self, embedding, k: int = 4, **kw) -> list[tuple[Document, float]]:
        s, r = self._search_vecs(np.asarray(embedding, dtype=np.float32)[None], k)
        return [(self._doc(int(row)), self._convert(float(sc))) for sc, row in zip(s[0], r[0]) if row >= 0]

    def _fused_query(self, query: str, fetch_k: int, k: int, lambda_mult):
        """One query through `rmu_bert_search_mmr` (token ids in, rows out: forward, dense top-fetch_k and the selection in ONE library
        call with one synchronisation) when the Embeddings object and the index are the native ones; None otherwise."""
        if self._pending:
            self.flush()
        emb, idx = self._embeddings, self._index
        if (idx is None or not hasattr(idx, "_h") or len(idx) == 0 or not hasattr(emb, "query_ids") or not (1 <= k <= fetch_k <= 64)
                or idx.dim != 384):
            return None
        q = emb.query_ids(query)
        if q is None:
            return None
        rows, scores = emb.encoder.search_host(idx, q[0], q[1], emb._mode, fetch_k, k, lambda_mult)
        return rows[0], scores[0]

    def similarity_search_with_score(self, query: str, k: int = 4, **kw) -> list[tuple[Document, float]]:
        hit = self._fused_query(query, int(k), int(k), None)
        if hit is not None:
            return [(self._doc(int(row)), self._convert(float(sc))) for row, sc in zip(*hit) if row >= 0]
        return self.similarity_search_with_score_by_vector(self._embed_query(query), k, **kw)

    def similarity_search(self, query: str, k: int = 4, **kw) -> list[Document]:
        return [d for d, _ in self.similarity_search_with_score(query, k, **kw)]

    def similarity_search_with_relevance_scores(self, query: str, k: int = 4, **kw) -> list[tuple[Document, float]]:
        s, r = self._search_vecs(self._embed_query(query)[None], k)
        return [(self._doc(int(row)), float(sc)) for sc, row in zip(s[0], r[0]) if row >= 0]

    def similarity_search_with_score_batch(self, queries: list[str], k: int = 4) -> list[list[tuple[Document, float]]]:
        qv = self._embed_docs(list(queries))
        s, r = self._search_vecs(qv, k)
        return [[(self._doc(int(row)), self._convert(float(sc))) for sc, row in zip(ss, rr) if row >= 0]
                for ss, rr in zip(s, r)]

    def max_marginal_relevance_search_by_vector(self, embedding, k: int = 4, fetch_k: int = 20,
                                                lambda_mult: float = 0.5, **kw) -> list[Document]:
        q = np.asarray(embedding, dtype=np.float32)
        if self._pending:
            self.flush()
        if (self._index is not None and hasattr(self._index, "search_mmr") and len(self._index) > 0
                and 1 <= k <= int(fetch_k) <= 64):
            # dense top-fetch_k and the greedy selection in one library call (rmu_index_search_mmr, fetch_k <= 64): one host
            # round trip.  A larger pool takes the search + selection below with the FULL fetch_k candidates.
            rows, _ = self._index.search_mmr(q[None], int(fetch_k), k, lambda_mult)
            return [self._doc(int(x)) for x in rows[0] if x >= 0]
        s, r = self._search_vecs(q[None], fetch_k)
        rows = [int(x) for x in r[0] if x >= 0]
        if not rows:
            return []
        if r.shape[1] <= 64 and hasattr(self._index, "mmr"):
            # the greedy selection on the device (rmu_index_mmr: fp64, same rule and tie order; tests/test_search_gpu.py holds it to
            # the oracle): no re-fetch of the candidate vectors, no per-pick numpy calls (0.32 ms of a 0.58 ms query on the host)
            pos = self._index.mmr(q[None], r, k, lambda_mult)[0]
            return [self._doc(int(r[0, p])) for p in pos if p >= 0]
        cand = self._index.get_rows(rows)      # the `pk in [...]` vector re-fetch of the replaced store
        picked = maximal_marginal_relevance(q, cand, k=k, lambda_mult=lambda_mult)
        return [self._doc(rows[i]) for i in picked]

    def max_marginal_relevance_search(self, query: str, k: int = 4, fetch_k: int = 20, lambda_mult: float = 0.5,
                                      **kw) -> list[Document]:
        hit = self._fused_query(query, int(fetch_k), int(k), float(lambda_mult))     # the reference's per-request call (RAGHelper.py:497-499)
        if hit is not None:
            return [self._doc(int(x)) for x in hit[0] if x >= 0]
        return self.max_marginal_relevance_search_by_vector(self._embed_query(query), k, fetch_k, lambda_mult)

    def max_marginal_relevance_search_batch(self, queries: list[str], k: int = 4, fetch_k: int = 20,
                                            lambda_mult: float = 0.5, **kw) -> list[list[Document]]:
        """One dense search + ONE device-side MMR selection for the whole batch (`rmu_index_mmr`: fp64, same greedy rule and
        tie order as the single-query path; no per-query vector re-fetch).  fetch_k > 64 falls back to the host loop."""
        qv = self._embed_docs(list(queries))
        s, r = self._search_vecs(qv, fetch_k)
        if r.shape[1] == 0:
            return [[] for _ in range(qv.shape[0])]
        if r.shape[1] <= 64 and hasattr(self._index, "mmr"):
            pos = self._index.mmr(qv, r, k, lambda_mult)
            return [[self._doc(int(r[qi, p])) for p in pos[qi] if p >= 0] for qi in range(qv.shape[0])]
        out = []
        for qi in range(qv.shape[0]):
            rows = [int(x) for x in r[qi] if x >= 0]
            if not rows:
                out.append([])
                continue
            cand = self._index.get_rows(rows)
            out.append([self._doc(rows[i]) for i in maximal_marginal_relevance(qv[qi], cand, k, lambda_mult)])
        return out

    def as_retriever(self, search_type: str = "similarity", search_kwargs: dict | None = None, **kw) -> MI355XRetriever:
        """`db.as_retriever(search_type="mmr", search_kwargs={"k": K})` (server/RAGHelper.py:497-499, :533-535)."""
        tags = list(kw.pop("tags", None) or []) + [type(self).__name__]
        return MI355XRetriever(vectorstore=self, search_type=search_type, search_kwargs=dict(search_kwargs or {}),
                               tags=tags, **kw)


class _RowsOutOfStep(Exception):
    def __init__(self, first: int, total: int):
        super().__init__(first, total)
        self.first = first          # the row the index gave the batch
        self.total = total          # rows the batch added there (tombstoned by the worker)


class _DeferredFailure:
    """A round of the insert pipeline that failed to start while the round in front of it was still to be appended."""

    def __init__(self, items, error):
        self.items, self.error = items, error


class _DeleteResult(int):
    """int subclass carrying ``delete_count`` (pymilvus MutationResult field read at server.py:385)."""

    def __new__(cls, n: int):
        obj = super().__new__(cls, n)
        obj.delete_count = n
        return obj

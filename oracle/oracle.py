"""CPU ORACLE for the RAGMeUp retrieval hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module.  The product path (``ragmeup_amd``) never imports it and fails loudly
when the HIP library is missing.

PARITY UNPINNED: the reference (/root/reference, snapshot 2025-01-03) contains no tests,
golden vectors or fixtures for this path, and every floating-point
operation on the path runs inside third-party packages that are not vendored under
/root/reference and are not installable here (no network):

  langchain-huggingface==0.0.3 -> sentence-transformers==2.6.1 -> transformers==4.43.1
  langchain-milvus==0.1.3 / milvus-lite==2.4.7 / pymilvus==2.4.3
  langchain-postgres==0.0.12 + pgvector v0.8.0
  langchain-community==0.2.10 (HuggingFaceCrossEncoder), langchain==0.2.11

(server/requirements.txt:1-41).  Each function below restates the *published* algorithm of
the pinned dependency and anchors on the reference's own call site (cited per function).
The one reference-owned piece of semantics, ScoredCrossEncoderReranker.compress_documents
(server/ScoredCrossEncoderReranker.py:42-45), is restated literally in ``rerank``.

All arithmetic is numpy fp64 (scores) so that the oracle is the *more* precise side of every
comparison; tolerances live in the tests.
"""
from __future__ import annotations

import numpy as np

METRIC_IP = 0
METRIC_COSINE = 1
METRIC_L2SQ = 2


# ----------------------------------------------------------------------------------------------
# seeded synthetic inputs
# ----------------------------------------------------------------------------------------------
def make_corpus(n: int, d: int = 384, seed: int = 1234) -> np.ndarray:
    """Unit-norm fp32 corpus rows, rng seed 1234."""
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((n, d), dtype=np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    return np.ascontiguousarray(x, dtype=np.float32)


def make_queries(x: np.ndarray, nq: int, seed: int = 4321, noise: float = 0.1) -> np.ndarray:
    """Queries = perturbed corpus rows (so every query has a planted near neighbour)."""
    rng = np.random.default_rng(seed)
    perm = rng.permutation(x.shape[0])[:nq]
    if perm.shape[0] < nq:  # tiny corpora: sample with replacement
        perm = rng.integers(0, x.shape[0], nq)
    q = x[perm] + noise * rng.standard_normal((nq, x.shape[1]), dtype=np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    return np.ascontiguousarray(q, dtype=np.float32), perm


# ----------------------------------------------------------------------------------------------
# a5: flat dense search  (reference call site: server/RAGHelper.py:497-499 -> 3P similarity search;
#     Milvus-Lite FLAT/L2 or pgvector `<=>`)
# ----------------------------------------------------------------------------------------------
def scores_f64(q: np.ndarray, x: np.ndarray, metric: int = METRIC_IP) -> np.ndarray:
    """Similarity in fp64, larger = better for every metric.

    IP:      q.x
    COSINE:  q.x / (|q| |x|)
    L2SQ:    -(|q - x|^2)     (Milvus returns the positive squared distance, smaller = better)
    """
    q64 = q.astype(np.float64)
    x64 = x.astype(np.float64)
    s = q64 @ x64.T
    if metric == METRIC_COSINE:
        qn = np.maximum(np.linalg.norm(q64, axis=1, keepdims=True), 1e-300)
        xn = np.maximum(np.linalg.norm(x64, axis=1, keepdims=True), 1e-300)
        s = s / qn / xn.T
    elif metric == METRIC_L2SQ:
        s = -((q64 * q64).sum(1, keepdims=True) - 2.0 * s + (x64 * x64).sum(1)[None, :])
    return s


def flat_search(q: np.ndarray, x: np.ndarray, k: int, metric: int = METRIC_IP,
                alive: np.ndarray | None = None, block: int = 262144):
    """Exact top-k.  Order: (-score, row) -- SURVEY 8c-5 tie rule.

    Returns (scores fp64 [nq,k], rows int64 [nq,k]); slots beyond the number of live rows hold
    (-inf, -1).
    """
    nq = q.shape[0]
    n = x.shape[0]
    best_s = np.full((nq, k), -np.inf, dtype=np.float64)
    best_r = np.full((nq, k), -1, dtype=np.int64)
    for b0 in range(0, n, block):
        b1 = min(n, b0 + block)
        s = scores_f64(q, x[b0:b1], metric)
        if alive is not None:
            s = np.where(alive[b0:b1][None, :], s, -np.inf)
        rows = np.arange(b0, b1, dtype=np.int64)
        cs = np.concatenate([best_s, s], axis=1)
        cr = np.concatenate([best_r, np.broadcast_to(rows, (nq, b1 - b0))], axis=1)
        # dead / padding entries sort last: score -inf, row forced to a huge value
        rkey = np.where(np.isneginf(cs), np.iinfo(np.int64).max, cr)
        order = np.lexsort((rkey, -cs), axis=1)[:, :k]
        best_s = np.take_along_axis(cs, order, axis=1)
        best_r = np.take_along_axis(cr, order, axis=1)
        best_r = np.where(np.isneginf(best_s), -1, best_r)
    return best_s, best_r


def flat_search_f32_blas(q: np.ndarray, x: np.ndarray, k: int, block: int = 131072):
    """CPU *baseline* form of the same search (what a SIMD FLAT scan such as Milvus-Lite's does):
    fp32 sgemm (OpenBLAS, all host threads) + argpartition + sort of the k survivors.
    Inner product only.  Used by bench.py's cpu_baseline leg; ordering ties are not normalised here.
    """
    nq = q.shape[0]
    best_s = np.full((nq, k), -np.inf, dtype=np.float32)
    best_r = np.full((nq, k), -1, dtype=np.int64)
    for b0 in range(0, x.shape[0], block):
        b1 = min(x.shape[0], b0 + block)
        s = q @ x[b0:b1].T
        kk = min(k, b1 - b0)
        part = np.argpartition(-s, kk - 1, axis=1)[:, :kk]
        cs = np.concatenate([best_s, np.take_along_axis(s, part, axis=1)], axis=1)
        cr = np.concatenate([best_r, part + b0], axis=1)
        order = np.argsort(-cs, axis=1, kind="stable")[:, :k]
        best_s = np.take_along_axis(cs, order, axis=1)
        best_r = np.take_along_axis(cr, order, axis=1)
    return best_s, best_r


def merge_topk(part_scores: np.ndarray, part_rows: np.ndarray, k: int):
    """Merge per-shard top-k lists [parts, nq, k'] -> [nq, k] by (-score, row)  (SURVEY 8e)."""
    parts, nq, kk = part_scores.shape
    cs = np.transpose(part_scores, (1, 0, 2)).reshape(nq, parts * kk).astype(np.float64)
    cr = np.transpose(part_rows, (1, 0, 2)).reshape(nq, parts * kk).astype(np.int64)
    rkey = np.where((cr < 0) | np.isneginf(cs), np.iinfo(np.int64).max, cr)
    order = np.lexsort((rkey, -cs), axis=1)[:, :k]
    s = np.take_along_axis(cs, order, axis=1)
    r = np.take_along_axis(cr, order, axis=1)
    return s, np.where(np.isneginf(s), -1, r)


# ----------------------------------------------------------------------------------------------
# a6: MMR (reference call site: server/RAGHelper.py:497-499, search_type="mmr";
#     3P langchain_core.vectorstores.utils.maximal_marginal_relevance, fetch_k=20, lambda=0.5)
# ----------------------------------------------------------------------------------------------
def _cosine_matrix(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """langchain_core.vectorstores.utils.cosine_similarity (numpy branch): dot(X, Y^T)/outer(|X|,|Y|) in
    float64, NaN/inf -> 0."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    an = np.linalg.norm(a, axis=1)
    bn = np.linalg.norm(b, axis=1)
    with np.errstate(divide="ignore", invalid="ignore"):
        s = np.dot(a, b.T) / np.outer(an, bn)
    s[np.isnan(s) | np.isinf(s)] = 0.0
    return s


def mmr(query_vec: np.ndarray, cand: np.ndarray, k: int = 4, lambda_mult: float = 0.5) -> list[int]:
    """Greedy maximal marginal relevance; strict '>' so the lowest index wins ties.  The similarity to the
    selected set is re-evaluated against the growing `selected` matrix each round, as published."""
    cand = np.asarray(cand)
    n = cand.shape[0]
    if min(k, n) <= 0:
        return []
    sim_q = _cosine_matrix(np.asarray(query_vec).reshape(1, -1), cand)[0]
    first = int(np.argmax(sim_q))
    picked = [first]
    selected = np.array([cand[first]])
    while len(picked) < min(k, n):
        sim_sel = _cosine_matrix(cand, selected)
        best, best_i = -np.inf, -1
        for i, qs in enumerate(sim_q):
            if i in picked:
                continue
            val = lambda_mult * qs - (1.0 - lambda_mult) * max(sim_sel[i])
            if val > best:
                best, best_i = val, i
        picked.append(best_i)
        selected = np.append(selected, [cand[best_i]], axis=0)
    return picked


# ----------------------------------------------------------------------------------------------
# a8: ScoredCrossEncoderReranker.compress_documents (server/ScoredCrossEncoderReranker.py:42-45)
# ----------------------------------------------------------------------------------------------
def rerank(scores, top_n: int = 3) -> list[tuple[int, float]]:
    """zip(docs, scores) -> sorted(key=score, reverse=True) -> [:top_n].

    Python's sort is stable and ``reverse=True`` preserves the original order of equal keys, so
    ties keep input order.  Returns [(doc_index, score)].
    """
    pairs = list(enumerate(scores))
    pairs = sorted(pairs, key=lambda p: p[1], reverse=True)
    return [(i, s) for i, s in pairs[:top_n]]


# ----------------------------------------------------------------------------------------------
# f-1: weighted reciprocal rank fusion (server/RAGHelper.py:500-503 -> 3P EnsembleRetriever,
#      c=60, weights 0.5/0.5, de-dup on page_content)
# ----------------------------------------------------------------------------------------------
def weighted_rrf(rank_lists: list[list], weights: list[float], c: int = 60) -> list:
    score: dict = {}
    order: list = []
    for lst, w in zip(rank_lists, weights):
        for rank, key in enumerate(lst, start=1):
            if key not in score:
                score[key] = 0.0
                order.append(key)
            score[key] += w / (rank + c)
    # sorted() is stable: ties keep first-seen order (chain over the retrievers' lists)
    return sorted(order, key=lambda kx: score[kx], reverse=True)


# ----------------------------------------------------------------------------------------------
# a2/a3/a7: BERT-6x384 encoder forward, restated in numpy fp64 from the published architecture
# (transformers BertModel: embeddings+LN, 6 x [MHA, add+LN, FFN(GELU erf), add+LN]); pooling as
# sentence-transformers (masked mean, L2 normalise) or CrossEncoder (pooler tanh + Linear(384,1)).
# Reference call sites: RAGHelper_local.py:107-117 (embeddings), RAGHelper.py:483-486 (cross-encoder).
# Weights are a dict of numpy arrays keyed with HF parameter names (see tests/helpers.make_bert_weights).
# ----------------------------------------------------------------------------------------------
def _ln(x, g, b, eps):
    mu = x.mean(-1, keepdims=True)
    var = ((x - mu) ** 2).mean(-1, keepdims=True)
    return (x - mu) / np.sqrt(var + eps) * g + b


def _gelu_erf(x):
    from scipy.special import erf
    return 0.5 * x * (1.0 + erf(x / np.sqrt(2.0)))


def bert_hidden(w: dict, ids: np.ndarray, type_ids: np.ndarray, lens: np.ndarray,
                n_layers: int = 6, n_heads: int = 12, eps: float = 1e-12) -> np.ndarray:
    """Final hidden states [B, L, H] fp64.  ids/type_ids [B, L] int, lens [B] (valid prefix length)."""
    f = lambda name: np.asarray(w[name], dtype=np.float64)
    B, L = ids.shape
    pos = np.arange(L)
    h = f("embeddings.word_embeddings.weight")[ids] + f("embeddings.position_embeddings.weight")[pos][None] \
        + f("embeddings.token_type_embeddings.weight")[type_ids]
    h = _ln(h, f("embeddings.LayerNorm.weight"), f("embeddings.LayerNorm.bias"), eps)
    H = h.shape[-1]
    dh = H // n_heads
    mask = (pos[None, :] < lens[:, None])                      # [B, L] True = valid key
    neg = np.where(mask, 0.0, -np.inf)[:, None, None, :]       # additive key mask
    for l in range(n_layers):
        p = f"encoder.layer.{l}."
        q = h @ f(p + "attention.self.query.weight").T + f(p + "attention.self.query.bias")
        k = h @ f(p + "attention.self.key.weight").T + f(p + "attention.self.key.bias")
        v = h @ f(p + "attention.self.value.weight").T + f(p + "attention.self.value.bias")
        sp = lambda t: t.reshape(B, L, n_heads, dh).transpose(0, 2, 1, 3)
        s = sp(q) @ sp(k).transpose(0, 1, 3, 2) / np.sqrt(dh) + neg
        s = s - s.max(-1, keepdims=True)
        pr = np.exp(s)
        pr /= pr.sum(-1, keepdims=True)
        ctx = (pr @ sp(v)).transpose(0, 2, 1, 3).reshape(B, L, H)
        a = ctx @ f(p + "attention.output.dense.weight").T + f(p + "attention.output.dense.bias")
        h = _ln(a + h, f(p + "attention.output.LayerNorm.weight"), f(p + "attention.output.LayerNorm.bias"), eps)
        m = _gelu_erf(h @ f(p + "intermediate.dense.weight").T + f(p + "intermediate.dense.bias"))
        o = m @ f(p + "output.dense.weight").T + f(p + "output.dense.bias")
        h = _ln(o + h, f(p + "output.LayerNorm.weight"), f(p + "output.LayerNorm.bias"), eps)
    return h


def embed_pool(hidden: np.ndarray, lens: np.ndarray) -> np.ndarray:
    """sentence-transformers Pooling(mean) + Normalize: sum(mask*h)/max(sum(mask),1e-9); x/max(|x|,1e-12)."""
    B, L, H = hidden.shape
    mask = (np.arange(L)[None, :] < lens[:, None]).astype(np.float64)[..., None]
    pooled = (hidden * mask).sum(1) / np.maximum(mask.sum(1), 1e-9)
    return pooled / np.maximum(np.linalg.norm(pooled, axis=1, keepdims=True), 1e-12)


def cross_encoder_logit(w: dict, hidden: np.ndarray) -> np.ndarray:
    """BertForSequenceClassification(num_labels=1): tanh(W_p h_CLS + b_p) -> Linear(384,1); logits[:,0]."""
    f = lambda name: np.asarray(w[name], dtype=np.float64)
    cls = hidden[:, 0, :]
    pooled = np.tanh(cls @ f("pooler.dense.weight").T + f("pooler.dense.bias"))
    return (pooled @ f("classifier.weight").T + f("classifier.bias"))[:, 0]

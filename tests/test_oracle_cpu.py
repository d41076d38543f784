"""CPU: the oracle against its own alternative statements, the committed goldens and the real
third-party BERT implementation (transformers) it restates."""
import json
import os

import numpy as np
import pytest

from oracle import cflat, oracle as O
from tests.helpers import assert_topk_parity, bert_weights_numpy, make_bert, synth_tokens

HERE = os.path.dirname(os.path.abspath(__file__))


def test_numpy_oracle_matches_c_oracle():
    x = O.make_corpus(20000)
    q, planted = O.make_queries(x, 24)
    s, r = O.flat_search(q, x, 10)
    cs, cr = cflat.flat_search(q, x, 10)
    assert_topk_parity(cs, cr, s, r)
    assert (r[:, 0] == planted).all()          # planted neighbour is the top hit


@pytest.mark.parametrize("metric", [O.METRIC_IP, O.METRIC_COSINE, O.METRIC_L2SQ])
def test_c_oracle_metrics(metric):
    rng = np.random.default_rng(3)
    x = rng.standard_normal((3000, 96)).astype(np.float32) * rng.uniform(0.5, 2.0, (3000, 1)).astype(np.float32)
    q = rng.standard_normal((5, 96)).astype(np.float32)
    s, r = O.flat_search(q, x, 7, metric)
    cs, cr = cflat.flat_search(q, x, 7, metric)
    assert_topk_parity(cs, cr, s, r, score_tol=1e-3 if metric == O.METRIC_L2SQ else 1e-4)


@pytest.mark.parametrize("metric,sk_metric", [(O.METRIC_L2SQ, "sqeuclidean"), (O.METRIC_COSINE, "cosine"), (O.METRIC_IP, None)])
def test_oracle_flat_search_against_independent_third_party_exact_search(metric, sk_metric):
    """The stores the reference runs (Milvus-Lite FLAT, pgvector) are not installable here, so the flat-search oracle cannot be pinned on
    THEM (parity unpinned) -- but exact nearest-neighbour search has independent third-party statements that ARE installed:
    scikit-learn's brute-force NearestNeighbors (squared Euclidean = Milvus' "L2", cosine distance = pgvector's `<=>`) and scipy's cdist
    (distance values).  Ids under the tie rule, distances to 1e-5; inner product against a plain float64 matmul + argsort."""
    from scipy.spatial.distance import cdist
    from sklearn.neighbors import NearestNeighbors
    rng = np.random.default_rng(11)
    x = (rng.standard_normal((6000, 384)) * rng.uniform(0.5, 2.0, (6000, 1))).astype(np.float32)
    q = (x[rng.permutation(6000)[:40]] + 0.2 * rng.standard_normal((40, 384))).astype(np.float32)
    k = 10
    s, r = O.flat_search(q, x, k + 2, metric)                                  # oracle: larger = better
    if sk_metric is None:
        full = q.astype(np.float64) @ x.astype(np.float64).T
        ref_r = np.argsort(-full, axis=1, kind="stable")[:, :k]
        ref_s = np.take_along_axis(full, ref_r, 1)
    else:
        nn = NearestNeighbors(n_neighbors=k, algorithm="brute", metric=sk_metric).fit(x.astype(np.float64))
        dist, ref_r = nn.kneighbors(q.astype(np.float64))
        d2 = cdist(q.astype(np.float64), x.astype(np.float64), sk_metric)
        assert np.allclose(np.take_along_axis(d2, ref_r, 1), dist, atol=1e-9)  # (the two third-party statements agree with each other)
        ref_s = -dist if metric == O.METRIC_L2SQ else 1.0 - dist             # oracle convention: -|q - x|^2, cosine similarity
    assert_topk_parity(ref_s.astype(np.float64), ref_r, s, r, score_tol=1e-5 * (1 + np.abs(ref_s).max()), tie_tol=1e-6 * (1 + np.abs(ref_s).max()))


def test_unit_norm_orderings_coincide():
    """on unit vectors L2 (Milvus), cosine distance (pgvector) and IP rank identically."""
    x = O.make_corpus(4000)
    q, _ = O.make_queries(x, 8)
    _, r_ip = O.flat_search(q, x, 10, O.METRIC_IP)
    _, r_cos = O.flat_search(q, x, 10, O.METRIC_COSINE)
    s_l2, r_l2 = O.flat_search(q, x, 10, O.METRIC_L2SQ)
    assert np.array_equal(r_ip, r_cos) and np.array_equal(r_ip, r_l2)
    s_ip, _ = O.flat_search(q, x, 10, O.METRIC_IP)
    assert np.allclose(-s_l2, 2.0 - 2.0 * s_ip, atol=1e-6)


def test_edges_empty_ragged_ties_alive():
    x = O.make_corpus(7)
    q, _ = O.make_queries(x, 3)
    s, r = O.flat_search(q, x, 10)                     # k > n
    assert (r[:, 7:] == -1).all() and np.isneginf(s[:, 7:]).all() and (r[:, :7] >= 0).all()
    xd = np.concatenate([x, x[:3]])                    # exact duplicates: lower row id first
    s, r = O.flat_search(x[:1], xd, 3)
    assert list(r[0][:2]) == [0, 7]
    alive = np.ones(7, bool); alive[int(O.flat_search(q, x, 1)[1][0, 0])] = False
    s2, r2 = O.flat_search(q[:1], x, 1, alive=alive)
    assert r2[0, 0] != O.flat_search(q[:1], x, 1)[1][0, 0]
    cs, cr = cflat.flat_search(q[:1], x, 1, alive=alive)
    assert cr[0, 0] == r2[0, 0]


def test_merge_topk_equals_global_search():
    x = O.make_corpus(6000)
    q, _ = O.make_queries(x, 9)
    parts_s, parts_r = [], []
    for lo in range(0, 6000, 1500):
        s, r = O.flat_search(q, x[lo:lo + 1500], 10)
        parts_s.append(s); parts_r.append(r + lo)
    ms, mr = O.merge_topk(np.stack(parts_s), np.stack(parts_r), 10)
    gs, gr = O.flat_search(q, x, 10)
    assert np.array_equal(mr, gr) and np.allclose(ms, gs)


def test_mmr_known_answer():
    # two near-duplicates and one orthogonal-ish vector: MMR must skip the duplicate
    q = np.array([1.0, 0.0, 0.0])
    c = np.array([[1.0, 0.0, 0.0], [0.999, 0.01, 0.0], [0.6, 0.8, 0.0], [0.0, 0.0, 1.0]])
    # second pick: lambda*sim_q - (1-lambda)*max sim to picked.  lambda=0.3: c1 -> -0.40, c2 -> -0.24, c3 -> 0
    assert O.mmr(q, c, k=2, lambda_mult=0.3) == [0, 3]
    assert O.mmr(q, c, k=2, lambda_mult=1.0) == [0, 1]      # pure relevance
    assert O.mmr(q, c, k=2, lambda_mult=0.5) == [0, 1]      # all three score exactly 0: lowest index wins
    assert O.mmr(q, c, k=10, lambda_mult=0.5)[:1] == [0] and len(O.mmr(q, c, 10)) == 4
    assert O.mmr(q, c[:0], k=3) == []
    # ties: identical candidates -> lowest index first (strict '>')
    assert O.mmr(q, np.stack([c[2], c[2], c[0]]), k=3, lambda_mult=0.5)[0] == 2


def test_rerank_against_reference_golden():
    """tests/golden/rerank_golden.json was produced by the reference's own ScoredCrossEncoderReranker.py."""
    g = json.load(open(os.path.join(HERE, "golden", "rerank_golden.json")))
    for case in g["cases"]:
        top_n = case["top_n"] if case["top_n"] is not None else case["default_top_n"]
        got = O.rerank(case["scores"], top_n)
        want = [(int(d["page_content"].split()[1]), d["metadata"]["relevance_score"]) for d in case["result"]]
        assert got == want, case["name"]


def test_weighted_rrf():
    fused = O.weighted_rrf([["a", "b", "c"], ["b", "d"]], [0.5, 0.5], c=60)
    assert fused[0] == "b" and set(fused) == {"a", "b", "c", "d"}
    assert fused.index("a") < fused.index("c")


@pytest.mark.parametrize("layers", [2, 6])
def test_bert_restatement_matches_transformers(layers):
    """Pins the encoder oracle on the third-party implementation the reference calls (transformers BertModel)."""
    import torch
    m = make_bert(seed=0, layers=layers)
    w = bert_weights_numpy(m)
    ids, tt, lens = synth_tokens(6, seed=11, lmax=48, mean=32, std=10)
    mask = (np.arange(ids.shape[1])[None] < lens[:, None]).astype(np.int64)
    with torch.no_grad():
        ref = m(input_ids=torch.tensor(ids, dtype=torch.long), attention_mask=torch.tensor(mask),
                token_type_ids=torch.tensor(tt, dtype=torch.long)).last_hidden_state.numpy()
    got = O.bert_hidden(w, ids, tt, lens, n_layers=layers)
    for i, l in enumerate(lens):                     # padded positions are unspecified
        assert np.abs(got[i, :l] - ref[i, :l]).max() < 2e-4
    emb = O.embed_pool(got, lens)
    assert np.allclose(np.linalg.norm(emb, axis=1), 1.0, atol=1e-9)


def test_cross_encoder_head_matches_transformers():
    import torch
    m = make_bert(seed=1, layers=2, head=True)
    w = bert_weights_numpy(m)
    ids, tt, lens = synth_tokens(5, seed=12, lmax=40, mean=30, std=6, pair=True)
    mask = (np.arange(ids.shape[1])[None] < lens[:, None]).astype(np.int64)
    with torch.no_grad():
        ref = m(input_ids=torch.tensor(ids, dtype=torch.long), attention_mask=torch.tensor(mask),
                token_type_ids=torch.tensor(tt, dtype=torch.long)).logits[:, 0].numpy()
    got = O.cross_encoder_logit(w, O.bert_hidden(w, ids, tt, lens, n_layers=2))
    assert np.abs(got - ref).max() < 2e-4


def test_search_golden_fixture():
    """Committed regression fixture of the flat search (seeded inputs, fp64 oracle outputs)."""
    g = np.load(os.path.join(HERE, "golden", "search_golden.npz"))
    x = O.make_corpus(int(g["n"]), int(g["d"]), int(g["seed_x"]))
    q, _ = O.make_queries(x, int(g["nq"]), int(g["seed_q"]))
    s, r = O.flat_search(q, x, int(g["k"]))
    assert np.array_equal(r, g["rows"]) and np.allclose(s, g["scores"], atol=1e-12)
    cs, cr = cflat.flat_search(q, x, int(g["k"]))
    assert_topk_parity(cs, cr, g["scores"], g["rows"])

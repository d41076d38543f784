"""Shared test helpers (tests may import the oracle; the product package never does)."""
from __future__ import annotations

import numpy as np


def assert_topk_parity(got_s, got_r, ref_s, ref_r, score_tol=1e-4, tie_tol=1e-6):
    """Bit-exact ids; a position may differ only where the oracle's fp64 scores are within `tie_tol`
    (SURVEY.md 8c-5 tie rule).  Scores within `score_tol` (north_star: cosine scores within 1e-4).
    `ref_*` may be DEEPER than `got_*` (oracle run with k + a few): a near-tie that straddles the k boundary is then
    recognised instead of being reported as a miss."""
    got_s, got_r, ref_s, ref_r = map(np.asarray, (got_s, got_r, ref_s, ref_r))
    k = got_r.shape[1]
    assert ref_r.shape[0] == got_r.shape[0] and ref_r.shape[1] >= k, (got_r.shape, ref_r.shape)
    ref_sk, ref_rk = ref_s[:, :k], ref_r[:, :k]
    finite = np.isfinite(ref_sk)
    assert np.array_equal(np.isfinite(got_s), finite)
    bad = np.nonzero(got_r != ref_rk)
    for qi, pi in zip(*bad):
        # a swap between near-ties: the id we returned must appear in the oracle row with ~the same score
        where = np.nonzero(ref_r[qi] == got_r[qi, pi])[0]
        assert where.size == 1, f"query {qi} pos {pi}: row {got_r[qi, pi]} not in the oracle top-{ref_r.shape[1]}"
        assert abs(ref_s[qi, where[0]] - ref_sk[qi, pi]) <= tie_tol, (
            f"query {qi} pos {pi}: id mismatch is not a near-tie "
            f"({ref_s[qi, where[0]]} vs {ref_sk[qi, pi]})")
    assert np.abs(got_s[finite].astype(np.float64) - ref_sk[finite]).max(initial=0.0) <= score_tol


def bert_config(layers=6):
    from transformers import BertConfig
    return BertConfig(vocab_size=30522, hidden_size=384, num_hidden_layers=layers, num_attention_heads=12,
                      intermediate_size=1536, max_position_embeddings=512, type_vocab_size=2,
                      layer_norm_eps=1e-12, hidden_act="gelu", attn_implementation="eager")


def make_bert(seed=0, layers=6, head=False):
    """Architecture-exact, weight-synthetic BERT-6x384 (no checkpoints exist offline)."""
    import torch
    from transformers import BertForSequenceClassification, BertModel
    torch.manual_seed(seed)
    cfg = bert_config(layers)
    if head:
        cfg.num_labels = 1
        m = BertForSequenceClassification(cfg)
    else:
        m = BertModel(cfg, add_pooling_layer=False)
    m.eval()
    # random-init LayerNorm/bias are 1/0: perturb them so a swapped gamma/beta or a dropped bias is caught
    g = torch.Generator().manual_seed(seed + 1000)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if "LayerNorm" in n or n.endswith(".bias"):
                p.add_(0.05 * torch.randn(p.shape, generator=g))
    return m


def bert_weights_numpy(model) -> dict:
    """HF parameter names without the 'bert.' prefix -> numpy fp32."""
    out = {}
    for n, p in model.state_dict().items():
        n = n[5:] if n.startswith("bert.") else n
        out[n] = p.detach().cpu().numpy()
    return out


def synth_tokens(n, seed=7, lmin=16, lmax=256, mean=128, std=32, pair=False):
    """Synthetic token ids as SURVEY.md 8d: [CLS]=101 ... [SEP]=102, L ~ clip(N(128,32),16,256)."""
    rng = np.random.default_rng(seed)
    lens = np.clip(np.rint(rng.normal(mean, std, n)), lmin, lmax).astype(np.int32)
    L = int(lens.max())
    ids = np.zeros((n, L), dtype=np.int32)
    tt = np.zeros((n, L), dtype=np.int32)
    for i, l in enumerate(lens):
        ids[i, :l] = rng.integers(1000, 30522, l)
        ids[i, 0] = 101
        ids[i, l - 1] = 102
        if pair:
            ql = min(16, l // 2)
            ids[i, ql] = 102
            tt[i, ql + 1:l] = 1
    return ids, tt, lens

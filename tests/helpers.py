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


def make_bert(seed=0, layers=6, head=False, scale=1.0):
    """Architecture-exact, weight-synthetic BERT-6x384 (no checkpoints exist offline).  scale > 1 multiplies
    every Linear weight of the encoder and head: at the 0.02 init every pair collapses onto one logit (spread 3e-3, the size
    of the bf16 noise); at 3x the cross-encoder logits spread by ~0.17 while the bf16 rounding model predicts |d| ~ 1e-2."""
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
            elif scale != 1.0 and p.dim() == 2 and "embeddings" not in n:
                p.mul_(scale)
    return m


def token_rel_error(got_packed, ref_hidden, lens):
    """Per-token relative L2 error |got - ref| / |ref| of final hidden states: got [sum(lens), H] packed in batch order
    (RMU_BERT_TOKENS), ref [B, L, H] padded (oracle.bert_hidden)."""
    rows = np.concatenate([np.asarray(ref_hidden[b, :l], np.float64) for b, l in enumerate(lens)])
    got = np.asarray(got_packed, np.float64)
    assert got.shape == rows.shape, (got.shape, rows.shape)
    return np.linalg.norm(got - rows, axis=1) / np.linalg.norm(rows, axis=1)


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


# ---- synthetic checkpoint directories (what `embedding_model` / `rerank_model` name in the reference's .env) -----------
_WORDS = ("retrieval augment ##ed ##ation gener ##ate vector store ##s query docu ##ment rank re ##rank chunk the a of and to in "
          "is for with on cafe naive model embed ##ding ##ing index search dense sparse score top answer question context "
          "gpu mi ##35 ##5 ##x hbm kernel wave lane tile batch token ##ize ##r layer norm head pool mean cls . , ? ! ( ) - : ; "
          "中 文 0 1 2 3 4 5 6 7 8 9").split()


def synth_vocab() -> list[str]:
    letters = [chr(c) for c in range(ord("a"), ord("z") + 1)]
    toks = ["[PAD]"] + [f"[unused{i}]" for i in range(99)] + ["[UNK]", "[CLS]", "[SEP]", "[MASK]"]
    toks += letters + ["##" + c for c in letters]
    seen = set(toks)
    for w in _WORDS:
        if w not in seen:
            toks.append(w); seen.add(w)
    return toks


def synth_texts(n, seed=0, wmin=3, wmax=60):
    import random
    rng = random.Random(seed)
    surface = ["Retrieval", "augmented", "augmentation", "generation", "generate", "vector", "stores", "query", "documents", "rerank",
               "chunks", "the", "a", "of", "and", "to", "in", "is", "for", "with", "on", "café", "naïve", "MI355X", "GPU", "HBM",
               "(rank)", "re-rank?", "中文", "zebra", "quux", "store,", "query.", "embedding", "indexing", "search", "dense", "sparse",
               "score", "top-10", "answer", "question:", "context;", "kernel", "wave", "lane", "tile", "batch", "tokenizer", "layer",
               "norm", "head", "pooling", "mean", "CLS!", "model\nmodel"]
    return [" ".join(rng.choice(surface) for _ in range(rng.randint(wmin, wmax))) for _ in range(n)]


def _write_tokenizer_files(d, lower=True, model_max_length=512):
    import json
    import os
    toks = synth_vocab()
    with open(os.path.join(d, "vocab.txt"), "w", encoding="utf-8") as f:
        f.write("\n".join(toks) + "\n")
    with open(os.path.join(d, "tokenizer_config.json"), "w") as f:
        json.dump({"do_lower_case": lower, "model_max_length": model_max_length, "tokenizer_class": "BertTokenizer"}, f)
    return toks


def write_st_checkpoint(d, pooling="mean", normalize=True, max_seq_length=256, layers=6, seed=0, st_files=True, lower=True, scale=1.0):
    """A sentence-transformers style directory around a seeded random-init BERT-{layers}x384: config.json +
    model.safetensors (BertModel.save_pretrained), vocab.txt, tokenizer_config.json, modules.json,
    sentence_bert_config.json, 1_Pooling/config.json (+ 2_Normalize).  Returns the transformers model (eval, fp32)."""
    import json
    import os
    import torch
    from transformers import BertModel
    os.makedirs(d, exist_ok=True)
    toks = synth_vocab()
    torch.manual_seed(seed)
    cfg = bert_config(layers)
    cfg.vocab_size = len(toks)
    m = BertModel(cfg, add_pooling_layer=False).eval()
    g = torch.Generator().manual_seed(seed + 1000)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if "LayerNorm" in n or n.endswith(".bias"):
                p.add_(0.05 * torch.randn(p.shape, generator=g))
            elif scale != 1.0 and p.dim() == 2 and "embeddings" not in n:
                p.mul_(scale)                                   # see make_bert: weights that do not collapse
    m.save_pretrained(d, safe_serialization=True)
    _write_tokenizer_files(d, lower=lower)
    if st_files:
        mods = [{"idx": 0, "name": "0", "path": "", "type": "sentence_transformers.models.Transformer"},
                {"idx": 1, "name": "1", "path": "1_Pooling", "type": "sentence_transformers.models.Pooling"}]
        if normalize:
            mods.append({"idx": 2, "name": "2", "path": "2_Normalize", "type": "sentence_transformers.models.Normalize"})
        json.dump(mods, open(os.path.join(d, "modules.json"), "w"))
        json.dump({"max_seq_length": max_seq_length, "do_lower_case": False}, open(os.path.join(d, "sentence_bert_config.json"), "w"))
        os.makedirs(os.path.join(d, "1_Pooling"), exist_ok=True)
        json.dump({"word_embedding_dimension": 384, "pooling_mode_cls_token": pooling == "cls",
                   "pooling_mode_mean_tokens": pooling == "mean", "pooling_mode_max_tokens": pooling == "max",
                   "pooling_mode_mean_sqrt_len_tokens": False}, open(os.path.join(d, "1_Pooling", "config.json"), "w"))
    return m


def write_ce_checkpoint(d, layers=6, seed=1, activation="identity"):
    """cross-encoder/ms-marco-MiniLM style directory: BertForSequenceClassification(num_labels=1).save_pretrained + tokenizer
    files; `activation` = "identity" writes sbert_ce_default_activation_function as the ms-marco configs do, None omits it
    (sentence-transformers then applies Sigmoid)."""
    import json
    import os
    import torch
    from transformers import BertForSequenceClassification
    os.makedirs(d, exist_ok=True)
    toks = synth_vocab()
    torch.manual_seed(seed)
    cfg = bert_config(layers)
    cfg.vocab_size = len(toks)
    cfg.num_labels = 1
    m = BertForSequenceClassification(cfg).eval()
    g = torch.Generator().manual_seed(seed + 1000)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if "LayerNorm" in n or n.endswith(".bias"):
                p.add_(0.05 * torch.randn(p.shape, generator=g))
    m.save_pretrained(d, safe_serialization=True)
    _write_tokenizer_files(d)
    if activation is not None:
        p = os.path.join(d, "config.json")
        c = json.load(open(p))
        c["sbert_ce_default_activation_function"] = {"identity": "torch.nn.modules.linear.Identity",
                                                     "sigmoid": "torch.nn.modules.activation.Sigmoid"}[activation]
        json.dump(c, open(p, "w"))
    return m


def hf_tokenizer(vocab_file, lower=True):
    """transformers' BertTokenizer over a vocab.txt (v5 takes the mapping, not the path)."""
    from transformers import BertTokenizer
    toks = open(vocab_file, encoding="utf-8").read().split("\n")
    if toks and toks[-1] == "":
        toks.pop()
    return BertTokenizer(vocab={t: i for i, t in enumerate(toks)}, do_lower_case=lower)


def st_reference_embed(model, vocab_file, texts, pooling="mean", normalize=True, max_seq_length=256, lower=True, batch=64):
    """What the reference's HuggingFaceEmbeddings computes for `texts`, restated with the third-party pieces that ARE
    installed: transformers' BertTokenizer + BertModel (fp32, host), then sentence-transformers' Pooling / Normalize."""
    import torch
    tok = hf_tokenizer(vocab_file, lower)
    texts = [t.replace("\n", " ") for t in texts]
    out = np.empty((len(texts), 384), np.float32)
    with torch.no_grad():
        for b0 in range(0, len(texts), batch):
            enc = tok(texts[b0:b0 + batch], padding=True, truncation=True, max_length=max_seq_length, return_tensors="pt")
            h = model(input_ids=enc["input_ids"], attention_mask=enc["attention_mask"],
                      token_type_ids=enc["token_type_ids"]).last_hidden_state
            if pooling == "cls":
                pooled = h[:, 0]
            else:
                mk = enc["attention_mask"].unsqueeze(-1).float()
                pooled = (h * mk).sum(1) / mk.sum(1).clamp(min=1e-9)
            if normalize:
                pooled = torch.nn.functional.normalize(pooled, p=2, dim=1)
            out[b0:b0 + batch] = pooled.numpy()
    return out


def centred_cosine(a, b):
    """cosine after removing the batch mean vector of `b` from both sides: random-init BERT maps every input close to one
    common direction (pairwise cosine ~0.98), which a plain cosine mostly measures; this one measures what is left."""
    mu = np.asarray(b, np.float64).mean(0, keepdims=True)
    ca, cb = np.asarray(a, np.float64) - mu, np.asarray(b, np.float64) - mu
    return (ca * cb).sum(1) / np.maximum(np.linalg.norm(ca, axis=1) * np.linalg.norm(cb, axis=1), 1e-300)

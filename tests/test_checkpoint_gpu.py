"""The product entry the reference's factories map onto, executed end to end on the MI355X from LOCAL
CHECKPOINT DIRECTORIES: `factory.from_env({...})` -> MI355XEmbeddings(model_dir=) / MI355XCrossEncoder(model_dir=) ->
checkpoint.read_* -> BertEncoder.from_spec -> librmu.so, with the native tokenizer built from the directory's vocab.txt.

Reference sites: HuggingFaceEmbeddings(model_name=os.getenv('embedding_model')) server/RAGHelper_local.py:107-117;
HuggingFaceCrossEncoder(model_name=self.rerank_model) server/RAGHelper.py:483-486; the 1000-document
`add_documents(documents, ids=ids)` loop :423-434; `as_retriever(search_type="mmr")` :497-499;
`compress_documents` server/ScoredCrossEncoderReranker.py:25-45; the template's default embedding model (CLS pooling)
server/.env.template:3.

Checkpoints are architecture-exact and weight-synthetic (transformers' own `save_pretrained` of a seeded random-init
model + a generated vocabulary: no real checkpoint exists offline).  The comparison side is the third-party code the
reference itself runs, as far as it is installed: transformers' BertTokenizer + BertModel / BertForSequenceClassification
in fp32 on the host, followed by sentence-transformers' pooling restated in tests/helpers.st_reference_embed.

Tolerances (bf16 MFMA activations vs fp32): plain cosine >= 0.999 AND mean-centred cosine >= 0.99 (random-init BERT maps
all inputs near one direction -- pairwise cosine ~0.98 -- so the centred value is the one that discriminates: another
chunk's vector scores < 0.5); cross-encoder logits within 2e-2 (1 + |logit|)."""
import hashlib

import numpy as np
import pytest

from tests.helpers import centred_cosine, hf_tokenizer, st_reference_embed, synth_texts, write_ce_checkpoint, write_st_checkpoint

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dirs(tmp_path_factory):
    root = tmp_path_factory.mktemp("ckpt")
    d_bi, d_ce = str(root / "all-MiniLM-like"), str(root / "ms-marco-like")
    m_bi = write_st_checkpoint(d_bi, pooling="mean", normalize=True, max_seq_length=64, layers=6, seed=0)
    m_ce = write_ce_checkpoint(d_ce, layers=6, seed=1, activation="identity")
    return d_bi, m_bi, d_ce, m_ce, root


def test_env_factory_end_to_end_from_checkpoint_directories(dirs):
    import torch
    from ragmeup_amd import factory
    from ragmeup_amd.documents import Document
    d_bi, m_bi, d_ce, m_ce, root = dirs
    env = {"vector_store": "mi355x", "embedding_model": d_bi, "rerank": "True", "rerank_model": d_ce, "rerank_k": "3",
           "vector_store_k": "4", "vector_store_initial_load": "True", "vector_store_uri": str(root / "data.db"),
           "vector_store_collection": "ckpt_e2e", "force_cpu": "False"}
    hp = factory.from_env(env)
    emb = hp.embeddings
    assert (emb.pooling, emb.normalize, emb.max_seq_length) == ("mean", True, 64)          # read from the directory
    assert type(emb.tokenizer).__name__ == "WordPieceTokenizer"
    texts = synth_texts(2300, seed=5)                                                         # > 2 of the reference's 1000-doc batches
    docs = [Document(t, {"source": f"s{i % 7}.pdf", "id": hashlib.md5(t.encode()).hexdigest()}) for i, t in enumerate(texts)]
    for i in range(0, len(docs), 1000):                                                       # RAGHelper.py:423-434
        batch = docs[i:i + 1000]
        hp.db.add_documents(batch, ids=[d.metadata["id"] for d in batch])
    n_unique = len({d.metadata["id"] for d in docs})
    assert len(hp.db) == n_unique
    # embeddings of the stored rows == what HuggingFaceEmbeddings would have computed (transformers fp32 + ST pooling)
    sample = texts[:256]
    got = np.asarray(emb.embed_documents(sample), np.float32)
    ref = st_reference_embed(m_bi, d_bi + "/vocab.txt", sample, "mean", True, 64)
    cos = (got * ref).sum(1)
    assert cos.min() >= 0.999, float(cos.min())
    cc = centred_cosine(got, ref)
    assert cc.min() >= 0.99, float(cc.min())
    assert centred_cosine(np.roll(ref, 1, axis=0), ref).max() < 0.9                           # the bar can fail: another chunk's vector does
    assert np.abs(np.linalg.norm(got, axis=1) - 1).max() < 1e-5
    # retrieval through the reference's call: every chunk finds itself first
    hits = hp.retriever.invoke(texts[17])
    assert hits and hits[0].page_content == texts[17] and {"source", "id", "pk"} <= set(hits[0].metadata)
    # rerank: the compressor built from `rerank_model`, logits vs BertForSequenceClassification fp32
    q = texts[3][:60]
    cands = [Document(t, {"source": "s", "id": str(i)}) for i, t in enumerate(texts[100:130])]
    out = hp.compressor.compress_documents(cands, q)
    assert len(out) == 3
    tok = hf_tokenizer(d_ce + "/vocab.txt")
    enc = tok([q] * 30, [d.page_content for d in cands], padding=True, truncation="longest_first", max_length=512, return_tensors="pt")
    with torch.no_grad():
        ref_logits = m_ce(**enc).logits[:, 0].numpy()
    got_logits = np.asarray(hp.compressor.model.score([(q, d.page_content) for d in cands]))
    assert np.all(np.abs(got_logits - ref_logits) <= 2e-2 * (1 + np.abs(ref_logits))), (got_logits, ref_logits)
    order = np.argsort(-ref_logits, kind="stable")[:3]
    for o, i in zip(out, order):
        assert o.page_content == cands[i].page_content or abs(o.metadata["relevance_score"] - ref_logits[i]) < 1e-2
    hp.db.close() if hasattr(hp.db, "close") else None


@pytest.mark.parametrize("pooling,normalize,layers,scale", [("cls", True, 12, 3.0), ("mean", False, 6, 1.0), ("cls", False, 6, 3.0)],
                         ids=["gist_small_like_cls_12_layers", "mean_without_normalize", "cls_without_normalize"])
def test_pooling_and_normalize_follow_the_checkpoint(tmp_path, pooling, normalize, layers, scale):
    """server/.env.template:3's default (GIST-small: 12 layers, CLS pooling, Normalize) and the other combinations the
    sentence-transformers config files can declare.  The CLS cases use the 3x-scaled weight set: at the 0.02 init the [CLS]
    state of every input is the same vector to 2e-3 (pairwise cosine 0.998), below the bf16 noise of twelve layers (1.2e-2:
    the numpy rounding model gives a centred cosine of 0.94 there, 0.995 on the scaled weights), so nothing could be told apart.
    Bars: cosine >= 0.999, |got - ref| / |ref| <= 2.5e-2, mean-centred cosine >= 0.99."""
    from ragmeup_amd.embeddings import MI355XEmbeddings
    d = str(tmp_path / "m")
    model = write_st_checkpoint(d, pooling=pooling, normalize=normalize, max_seq_length=128, layers=layers, seed=3, scale=scale)
    emb = MI355XEmbeddings(model_dir=d)
    assert (emb.pooling, emb.normalize, emb.max_seq_length) == (pooling, normalize, 128)
    texts = synth_texts(96, seed=9, wmax=120)                                                 # some exceed 128 tokens: truncation
    got = emb.embed_documents_array(texts)
    ref = st_reference_embed(model, d + "/vocab.txt", texts, pooling, normalize, 128)
    gn, rn = np.linalg.norm(got, axis=1), np.linalg.norm(ref, axis=1)
    assert ((got * ref).sum(1) / gn / rn).min() >= 0.999
    assert (np.linalg.norm(got - ref, axis=1) / rn).max() <= 2.5e-2
    assert centred_cosine(got / gn[:, None], ref / rn[:, None]).min() >= 0.99
    if normalize:
        assert np.abs(gn - 1).max() < 1e-5
    else:
        assert np.abs(gn / rn - 1).max() < 2e-2 and rn.std() > 0                              # the raw pooled norm is kept
    # one query alone (the small-batch kernels, LayerNorms folded into their consumers) vs the same text inside the batch: two roundings of
    # the same fp32 forward.  Measured on the 12-layer, 3x-scaled CLS case over all 96 texts: each within 3.2e-3 / 3.5e-3 (max abs) and
    # 1.7e-2 / 1.9e-2 (rel. L2) of the fp32 reference, 3.2e-3 apart from each other (2.8e-3 before the LayerNorm fold).
    one = np.asarray(emb.embed_query(texts[5]), np.float32)
    assert np.abs(one - got[5]).max() < 4e-3 * max(1.0, float(rn[5]))
    assert np.linalg.norm(one - ref[5]) / rn[5] <= 2.5e-2


def test_cross_encoder_default_activation_is_sigmoid(tmp_path):
    """CrossEncoder applies Sigmoid when num_labels == 1 and config.json names no activation (the ms-marco models name
    Identity); HuggingFaceCrossEncoder.score returns what predict returns."""
    import torch
    from ragmeup_amd.embeddings import MI355XCrossEncoder
    d = str(tmp_path / "ce")
    model = write_ce_checkpoint(d, layers=6, seed=4, activation=None)
    ce = MI355XCrossEncoder(model_dir=d)
    assert ce.activation == "sigmoid" and ce.max_seq_length == 512
    texts = synth_texts(40, seed=2)
    pairs = [(texts[i][:50], texts[i + 20]) for i in range(20)]
    got = np.asarray(ce.score(pairs))
    tok = hf_tokenizer(d + "/vocab.txt")
    enc = tok([p[0] for p in pairs], [p[1] for p in pairs], padding=True, truncation="longest_first", max_length=512, return_tensors="pt")
    with torch.no_grad():
        ref = torch.sigmoid(model(**enc).logits[:, 0]).numpy()
    assert np.abs(got - ref).max() < 1e-2 and got.min() > 0 and got.max() < 1


def test_unsupported_checkpoint_raises_before_touching_the_gpu(tmp_path):
    from ragmeup_amd import checkpoint as C, factory
    d = str(tmp_path / "maxpool")
    write_st_checkpoint(d, pooling="max", layers=1)
    with pytest.raises(C.UnsupportedCheckpoint):
        factory.embeddings_from_env({"embedding_model": d})

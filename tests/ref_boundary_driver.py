"""Runs the REFERENCE's own retrieval wiring (server/RAGHelper.py, imported from /root/reference, never copied) on
top of our drop-in objects, with the third-party modules it imports replaced by tests/lcstub.py stand-ins that keep
LangChain's pydantic `isinstance` field validation.  Executed in a fresh interpreter by tests/test_boundary_cpu.py
(the package resolves its LangChain base classes at import time).  Prints one JSON object.

What executes unmodified from the reference: RAGHelper.__init__ (env parsing), _initialize_vector_store (:385-434,
incl. the 1000-document batch loop with md5 ids), _setup_retrievers (:492-505: BM25 + `db.as_retriever(search_type=
"mmr")` + EnsembleRetriever), _initialize_reranker (:476-490: the reference's ScoredCrossEncoderReranker around our
cross-encoder + ContextualCompressionRetriever), _add_to_vector_database (:518-538), format_documents, and the LCEL
compositions of RAGHelper_local.py:158 / :254-258.  The only edits are the bindings INTEGRATION.md section 2 lists
(names `Milvus` and `HuggingFaceCrossEncoder` of the RAGHelper module pointed at our classes).
"""
import hashlib
import json
import logging
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("RMU_REFERENCE_DIR", "/root/reference/server")
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import lcstub  # noqa: E402

lcstub.install()
sys.path.insert(0, REF)

import numpy as np  # noqa: E402

import ragmeup_amd  # noqa: E402,F401
from ragmeup_amd import _lc  # noqa: E402
from ragmeup_amd.vectorstore import MI355XRetriever, MI355XVectorStore  # noqa: E402

out = {"have_langchain": _lc.HAVE_LANGCHAIN}
assert _lc.HAVE_LANGCHAIN and _lc.Document is lcstub.Document

# ---- CPU stand-ins for the two GPU models (the boundary, not the arithmetic, is under test here) --------------------------
sys.path.insert(0, HERE)
from test_host_cpu import FakeStore  # noqa: E402  (MI355XVectorStore over an oracle-backed index: a TEST subclass)

from ragmeup_amd.embeddings import MI355XCrossEncoder, MI355XEmbeddings  # noqa: E402


class HashEmbeddings(MI355XEmbeddings):
    """MI355XEmbeddings with the encoder replaced by a hash (same class hierarchy -> same isinstance answers)."""

    def __init__(self):
        pass

    def embed_documents_array(self, texts):
        rows = []
        for t in texts:
            seed = int(hashlib.md5(t.replace("\n", " ").encode()).hexdigest()[:8], 16)
            v = np.random.default_rng(seed).standard_normal(384)
            rows.append(v / np.linalg.norm(v))
        return np.asarray(rows, dtype=np.float32)

    embed_documents_device = None

    def embed_documents(self, texts):
        return self.embed_documents_array(texts).tolist()


class OverlapCrossEncoder(MI355XCrossEncoder):
    def __init__(self, model_name=None):
        self.model_name = model_name

    def score(self, text_pairs):
        return [float(len(set(a.lower().split()) & set(b.lower().split()))) + 1e-4 * (int(hashlib.md5(b.encode()).hexdigest()[:4], 16) % 97)
                for a, b in text_pairs]


del HashEmbeddings.embed_documents_device
out["embeddings_is_Embeddings"] = isinstance(HashEmbeddings(), lcstub.Embeddings)
out["cross_encoder_is_both_bases"] = (isinstance(OverlapCrossEncoder(), lcstub.BaseCrossEncoder)
                                      and isinstance(OverlapCrossEncoder(), lcstub.CommunityBaseCrossEncoder))

# ---- the reference, as the maintainer's binding leaves it ----------------------------------------------------------------
tmp = tempfile.mkdtemp()
os.environ.update({
    "vector_store": "milvus", "vector_store_uri": os.path.join(tmp, "data.db"), "vector_store_collection": "ragmeup_documents",
    "vector_store_initial_load": "True", "vector_store_k": "4", "rerank": "True", "rerank_k": "3",
    "rerank_model": "cross-encoder/ms-marco-MiniLM-L-6-v2", "document_chunks_pickle": os.path.join(tmp, "chunks.pickle"),
    "data_directory": tmp, "file_types": "txt", "splitter": "RecursiveCharacterTextSplitter", "chunk_size": "512",
    "chunk_overlap": "20", "breakpoint_threshold_type": "percentile",
})
import RAGHelper as ref  # noqa: E402  (the reference's module)

out["reference_file"] = ref.__file__
ref.Milvus = FakeStore                               # vector store binding (here: the test subclass
                                                     # of MI355XVectorStore whose index is the CPU oracle -- no GPU on this box)
ref.HuggingFaceCrossEncoder = OverlapCrossEncoder    # cross-encoder binding

h = ref.RAGHelper(logging.getLogger("ref"))
h.embeddings = HashEmbeddings()
topics = ["alpha", "beta", "gamma", "delta", "epsilon"]
h.chunked_documents = []
for i in range(2300):                                 # > 2 of the reference's 1000-document insert batches
    text = f"chunk {i} about {topics[i % 5]} and {topics[(i // 5) % 5]} number {i * 7919 % 1000}"
    h.chunked_documents.append(ref.Document(page_content=text, metadata={
        "source": f"{topics[i % 5]}.pdf", "id": hashlib.md5(text.encode()).hexdigest()}))

chunker = h._create_semantic_chunker()                # SemanticChunker(self.embeddings, ...) type-checks Embeddings
out["semantic_chunker_ok"] = chunker.embeddings is h.embeddings

h._initialize_vector_store()                          # Milvus.from_documents([], emb, drop_old=..., ...) + batch loop
out["db_type"] = [c.__name__ for c in type(h.db).__mro__ if c.__module__.startswith("ragmeup_amd.")][0]
out["db_rows"] = len(h.db)
out["db_is_VectorStore"] = isinstance(h.db, lcstub.VectorStore)

h._setup_retrievers()                                 # BM25 + as_retriever("mmr") + EnsembleRetriever + reranker wiring
dense = h.ensemble_retriever.retrievers[1]
out["dense_type"] = type(dense).__name__
out["dense_is_VectorStoreRetriever"] = isinstance(dense, lcstub.VectorStoreRetriever) and isinstance(dense, MI355XRetriever)
out["compressor_type"] = type(h.compressor).__module__ + "." + type(h.compressor).__name__
out["rerank_retriever_type"] = type(h.rerank_retriever).__name__

query = "what about gamma and delta number 57"
ens_docs = h.ensemble_retriever.invoke(query)
out["ensemble_n"] = len(ens_docs)
out["ensemble_has_pk"] = all("pk" in d.metadata for d in dense.invoke(query))
rr_docs = h.rerank_retriever.invoke(query)
out["rerank_n"] = len(rr_docs)
out["rerank_scores_desc"] = [d.metadata["relevance_score"] for d in rr_docs]
out["rerank_keeps_source_id"] = all("source" in d.metadata and "id" in d.metadata for d in rr_docs)

# expected result, recomputed independently: same members, scored by the same model, stable sort, top_n
exp = sorted(((d, s) for d, s in zip(ens_docs, OverlapCrossEncoder().score([(query, d.page_content) for d in ens_docs]))),
             key=lambda p: p[1], reverse=True)[:3]
out["rerank_matches_expected"] = [d.page_content for d in rr_docs] == [d.page_content for d, _ in exp]

# LCEL compositions of RAGHelper_local.py:158 and :254-258
chain = (h.rerank_retriever | ref.RAGHelper.format_documents)
out["pipe_format_ok"] = chain.invoke(query).startswith("Document 0 content: ")
par = {"docs": h.ensemble_retriever, "context": h.ensemble_retriever | ref.RAGHelper.format_documents,
       "question": lcstub.RunnablePassthrough()} | lcstub.RunnableLambda(lambda d: d)
res = par.invoke(query)
out["dict_coercion_ok"] = sorted(res) == ["context", "docs", "question"] and len(res["docs"]) == len(ens_docs)

# our own compressor inside the reference's ContextualCompressionRetriever (either class works)
from ragmeup_amd.reranker import ScoredCrossEncoderReranker as OurReranker  # noqa: E402
ours = ref.ContextualCompressionRetriever(base_compressor=OurReranker(model=OverlapCrossEncoder(), top_n=3),
                                          base_retriever=h.ensemble_retriever)
out["our_reranker_same_result"] = [d.page_content for d in ours.invoke(query)] == [d.page_content for d in rr_docs]

# strictness of the stand-ins: a duck-typed object must be rejected (otherwise the test above proves nothing)
class Duck:
    def invoke(self, q, config=None):
        return []

    def score(self, pairs):
        return [0.0] * len(pairs)


rejected = []
for label, make in (("ensemble", lambda: ref.EnsembleRetriever(retrievers=[h.sparse_retriever, Duck()], weights=[0.5, 0.5])),
                    ("ref_reranker", lambda: ref.ScoredCrossEncoderReranker(model=Duck(), top_n=3)),
                    ("our_reranker", lambda: OurReranker(model=Duck(), top_n=3)),
                    ("our_reranker_extra", lambda: OurReranker(model=OverlapCrossEncoder(), bogus=1))):
    try:
        make()
    except Exception:  # noqa: BLE001
        rejected.append(label)
out["duck_rejected"] = rejected

# incremental add (RAGHelper.py:518-538) and delete by source (server.py:373-377)
new = [ref.Document(page_content="a brand new chunk about zeta", metadata={"source": "zeta.pdf", "id": "z1"})]
h._add_to_vector_database(new)
out["after_add_rows"] = len(h.db)
out["new_dense_type"] = type(h.ensemble_retriever.retrievers[1]).__name__
out["finds_new"] = any(d.metadata.get("source") == "zeta.pdf" for d in h.db.similarity_search("a brand new chunk about zeta", k=1))
out["delete_count"] = h.db.delete(expr='source == "alpha.pdf"').delete_count

# env factory (ragmeup_amd.factory): same variables, vector_store=mi355x
from ragmeup_amd import factory  # noqa: E402
env = dict(os.environ, vector_store="mi355x", vector_store_initial_load="True", rerank="False",
           vector_store_collection="factory_probe")   # a fresh collection: the product class itself (index built lazily)
class _Rec:                      # the logger server.py hands to its helper (server/server.py:134-146): anything with .info / .error
    def __init__(self):
        self.lines = []

    def info(self, m):
        self.lines.append(("info", str(m)))

    def error(self, m):
        self.lines.append(("error", str(m)))

    warning = info


rec = _Rec()
hp = factory.from_env(env, embeddings=HashEmbeddings(), logger=rec)
out["factory_logged"] = [m for _, m in rec.lines]
from ragmeup_amd import _log  # noqa: E402
out["logger_injected"] = _log.get_logger() is rec
_log.set_logger(None)
out["factory_db"] = type(hp.db).__name__
out["factory_retriever"] = [hp.retriever.search_type, hp.retriever.search_kwargs]
out["factory_compressor"] = hp.compressor
try:
    factory.embeddings_from_env(dict(env, force_cpu="True", embedding_model=tmp))
    out["force_cpu_raises"] = False
except RuntimeError:
    out["force_cpu_raises"] = True

print("RESULT " + json.dumps(out))

"""CPU: the drop-in boundary.  (1) With LangChain importable our classes ARE LangChain types and the
reference's own wiring code accepts them -- executed for real in tests/ref_boundary_driver.py against pydantic-strict
stand-ins (tests/lcstub.py); needs /root/reference, so it self-skips on the GPU box.  (2) Without LangChain the shims
give the same constructor/field behaviour."""
import json
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/server"


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "RAGHelper.py")), reason="reference tree not present on this box")
def test_reference_wiring_accepts_our_objects():
    r = subprocess.run([sys.executable, os.path.join(HERE, "ref_boundary_driver.py")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1]
    o = json.loads(line[len("RESULT "):])
    assert o["have_langchain"] and o["reference_file"].startswith(REF)
    # isinstance answers the reference's pydantic containers ask for
    assert o["embeddings_is_Embeddings"] and o["cross_encoder_is_both_bases"] and o["db_is_VectorStore"]
    assert o["dense_is_VectorStoreRetriever"] and o["semantic_chunker_ok"]
    # RAGHelper._initialize_vector_store: 2300 chunks through the reference's 1000-document insert loop
    assert o["db_type"] == "MI355XVectorStore" and o["db_rows"] == 2300
    # _setup_retrievers / _initialize_reranker: the REFERENCE's reranker class around our cross-encoder
    assert o["dense_type"] == "MI355XRetriever" and o["rerank_retriever_type"] == "ContextualCompressionRetriever"
    assert o["compressor_type"] == "ScoredCrossEncoderReranker.ScoredCrossEncoderReranker"
    # end to end: ensemble -> rerank, top_n = rerank_k = 3, scores descending, provenance metadata intact
    assert 4 <= o["ensemble_n"] <= 8 and o["ensemble_has_pk"]
    assert o["rerank_n"] == 3 and o["rerank_scores_desc"] == sorted(o["rerank_scores_desc"], reverse=True)
    assert o["rerank_keeps_source_id"] and o["rerank_matches_expected"] and o["our_reranker_same_result"]
    assert o["pipe_format_ok"] and o["dict_coercion_ok"]
    # the stand-ins are strict: duck-typed look-alikes are refused by every container
    assert o["duck_rejected"] == ["ensemble", "ref_reranker", "our_reranker", "our_reranker_extra"]
    # _add_to_vector_database + delete-by-source
    assert o["after_add_rows"] == 2301 and o["finds_new"] and o["new_dense_type"] == "MI355XRetriever"
    assert o["delete_count"] == 460
    # env factory
    assert o["factory_db"] == "MI355XVectorStore" and o["factory_retriever"] == ["mmr", {"k": 4}]
    assert o["factory_compressor"] is None and o["force_cpu_raises"]
    # the reference's logger (server/server.py:134-146) injected through the factory: the set-up lines of RAGHelper.py:387/496 go through it
    assert o["logger_injected"] and o["factory_logged"] == ["Setting up the MI355X vector store.", "Setting up the Vector Retriever."]


def test_shims_without_langchain():
    from ragmeup_amd import _lc
    if _lc.HAVE_LANGCHAIN:
        pytest.skip("LangChain is installed: the shims are not in use")
    from ragmeup_amd.embeddings import MI355XCrossEncoder, MI355XEmbeddings
    from ragmeup_amd.ensemble import MI355XEnsembleRetriever
    from ragmeup_amd.reranker import ScoredCrossEncoderReranker
    from ragmeup_amd.vectorstore import MI355XRetriever, MI355XVectorStore
    assert issubclass(MI355XEmbeddings, _lc.Embeddings) and issubclass(MI355XVectorStore, _lc.VectorStore)
    assert issubclass(MI355XRetriever, _lc.VectorStoreRetriever) and issubclass(MI355XRetriever, _lc.BaseRetriever)
    assert issubclass(MI355XCrossEncoder, _lc.BaseCrossEncoder) and issubclass(ScoredCrossEncoderReranker, _lc.BaseDocumentCompressor)
    assert issubclass(MI355XEnsembleRetriever, _lc.BaseRetriever)

    class M:
        def score(self, pairs):
            return [0.0] * len(pairs)

    rr = ScoredCrossEncoderReranker(model=M())                      # duck-typed model is fine without LangChain
    assert rr.top_n == 3
    with pytest.raises(TypeError):
        ScoredCrossEncoderReranker(model=M(), bogus=1)              # Config.extra = "forbid"
    with pytest.raises(TypeError):
        ScoredCrossEncoderReranker(model=object())                  # no score()
    with pytest.raises(TypeError):
        ScoredCrossEncoderReranker()                                # model is required
    with pytest.raises(ValueError):
        MI355XRetriever(vectorstore=None, search_type="bogus")
    r = MI355XRetriever(vectorstore=None, search_type="mmr", search_kwargs={"k": 5})
    assert r.search_kwargs == {"k": 5} and callable(r.invoke) and callable(r.get_relevant_documents)


def test_factory_env_errors(tmp_path):
    from ragmeup_amd import factory
    with pytest.raises(RuntimeError, match="force_cpu"):
        factory.embeddings_from_env({"force_cpu": "True", "embedding_model": str(tmp_path)})
    with pytest.raises(KeyError):
        factory.embeddings_from_env({})
    with pytest.raises(FileNotFoundError):
        factory.embeddings_from_env({"embedding_model": "sentence-transformers/all-MiniLM-L6-v2"})
    with pytest.raises(ValueError, match="mi355x"):
        factory.vector_store_from_env(object(), {"vector_store": "milvus"})
    assert factory.reranker_from_env({"rerank": "False"}) is None
    with pytest.raises(ValueError):
        factory.reranker_from_env({"rerank": "True", "rerank_model": "flashrank"})

"""Environment-driven construction of the hot-path objects -- the code behind the reference's factories.

The reference builds its plug-ins from environment variables (loaded from `.env` by server.py):
  embeddings   server/RAGHelper_local.py:107-117, server/RAGHelper_cloud.py:90-103
               (`embedding_model`, `force_cpu`)
  vector store server/RAGHelper.py:385-415   (`vector_store`, `vector_store_uri`, `vector_store_collection`,
                                              `vector_store_initial_load`)
  reranker     server/RAGHelper.py:476-490   (`rerank`, `rerank_model`, `rerank_k`)
  retriever    server/RAGHelper.py:497-499   (`vector_store_k`)
`from_env()` reads the same variables with the same meaning and returns our implementations; the reference-side
binding is one call to it from the `vector_store == "mi355x"` branches.

`force_cpu=True` is an error here, not a silent downgrade: the MI355X path has no CPU fallback.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Any, Mapping, Optional


def _flag(env: Mapping[str, str], name: str, default: bool = False) -> bool:
    v = env.get(name)
    return default if v is None else v == "True"      # the reference compares with the string "True"


def embeddings_from_env(env: Optional[Mapping[str, str]] = None, device: int = 0):
    """`_initialize_embeddings()` (RAGHelper_local.py:107-117): `embedding_model` names a local HF checkpoint
    directory (config.json + weights + vocab.txt); `force_cpu == "True"` raises."""
    env = os.environ if env is None else env
    if _flag(env, "force_cpu"):
        raise RuntimeError("force_cpu=True: the MI355X retrieval path has no CPU fallback "
                           "(unset force_cpu, or keep vector_store=milvus/postgres for the CPU reference path)")
    model = env.get("embedding_model")
    if not model:
        raise KeyError("embedding_model is not set")
    if not os.path.isdir(model):
        raise FileNotFoundError(f"embedding_model={model!r}: a local checkpoint directory is required "
                                "(the hub id of the reference's HuggingFaceEmbeddings cannot be downloaded here)")
    from .embeddings import MI355XEmbeddings
    return MI355XEmbeddings(model_dir=model, device=device)


def vector_store_from_env(embeddings: Any, env: Optional[Mapping[str, str]] = None, device: Optional[int] = None):
    """`_initialize_vector_store()` (RAGHelper.py:385-415) for `vector_store == "mi355x"`: the Milvus-form call with
    `drop_old = not vector_store_initial_load`, `connection_args={"uri": vector_store_uri}`, `collection_name`."""
    env = os.environ if env is None else env
    kind = env.get("vector_store")
    if kind != "mi355x":
        raise ValueError(f"vector_store={kind!r}: only 'mi355x' is served here "
                         "('milvus' / 'postgres' stay on the reference's own stores)")
    from .vectorstore import MI355XVectorStore
    return MI355XVectorStore.from_documents(
        [], embeddings,
        drop_old=not _flag(env, "vector_store_initial_load"),
        connection_args={"uri": env.get("vector_store_uri")},
        collection_name=env.get("vector_store_collection") or "LangChainCollection",
        device=device,
    )


def reranker_from_env(env: Optional[Mapping[str, str]] = None, device: int = 0):
    """`_initialize_reranker()` (RAGHelper.py:476-486): ScoredCrossEncoderReranker(model=<cross-encoder named by
    `rerank_model`>, top_n=`rerank_k`); None when `rerank != "True"`.  `flashrank` is the reference's other branch and
    is not served here."""
    env = os.environ if env is None else env
    if not _flag(env, "rerank"):
        return None
    if _flag(env, "force_cpu"):
        raise RuntimeError("force_cpu=True: the MI355X retrieval path has no CPU fallback")
    model = env.get("rerank_model")
    if not model or model == "flashrank":
        raise ValueError(f"rerank_model={model!r}: a local cross-encoder checkpoint directory is required")
    if not os.path.isdir(model):
        raise FileNotFoundError(f"rerank_model={model!r}: a local checkpoint directory is required")
    from .embeddings import MI355XCrossEncoder
    from .reranker import ScoredCrossEncoderReranker
    return ScoredCrossEncoderReranker(model=MI355XCrossEncoder(model_dir=model, device=device),
                                      top_n=int(env.get("rerank_k", "3")))


@dataclass
class HotPath:
    embeddings: Any
    db: Any
    retriever: Any
    compressor: Any


def from_env(env: Optional[Mapping[str, str]] = None, device: int = 0, embeddings: Any = None, logger: Any = None) -> HotPath:
    """Everything `RAGHelper` builds for the retrieval path, from the reference's own environment variables.
    `logger`: the logger the reference passes into its helper (server/server.py:134-146, `RAGHelperLocal(logger)`); the hot-path objects
    then log through it -- the same lines at the same places as server/RAGHelper.py:387, 482, 496 -- instead of the package's own."""
    from . import _log
    if logger is not None:
        _log.set_logger(logger)
    log = _log.get_logger()
    env = os.environ if env is None else env
    emb = embeddings if embeddings is not None else embeddings_from_env(env, device)
    log.info("Setting up the MI355X vector store.")
    db = vector_store_from_env(emb, env, device)
    log.info("Setting up the Vector Retriever.")
    retriever = db.as_retriever(search_type="mmr", search_kwargs={"k": int(env.get("vector_store_k", "4"))})
    if _flag(env, "rerank"):
        log.info("Setting up the ScoredCrossEncoderReranker.")
    return HotPath(emb, db, retriever, reranker_from_env(env, device))

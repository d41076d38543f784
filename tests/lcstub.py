"""Stand-ins for the third-party modules the reference's server/RAGHelper.py imports (none is installable offline).

NOT a LangChain re-implementation: just enough of its published 0.2.x surface for the reference's retrieval wiring
(`_initialize_vector_store`, `_setup_retrievers`, `_initialize_reranker`, LCEL `|` / dict coercion) to execute, with
the one property the boundary test is about kept strict -- every container is a pydantic model whose fields are
validated by `isinstance` against the LangChain base types (`arbitrary_types_allowed`), exactly how the real
`EnsembleRetriever`, `ContextualCompressionRetriever`, `VectorStoreRetriever` and the reference's
`ScoredCrossEncoderReranker.model: BaseCrossEncoder` reject foreign objects.  `install()` registers the modules in
`sys.modules`; run it before importing `ragmeup_amd` (the package resolves its base classes at import time).
"""
from __future__ import annotations

import sys
import types
import warnings
from abc import ABC, abstractmethod
from typing import Any, Callable, ClassVar, Dict, List, Optional, Sequence

from pydantic import BaseModel, ConfigDict, Field

warnings.filterwarnings("ignore", category=DeprecationWarning)
try:
    from pydantic.warnings import PydanticDeprecatedSince20
    warnings.filterwarnings("ignore", category=PydanticDeprecatedSince20)
except Exception:  # noqa: BLE001
    pass


# ---- langchain_core.runnables ---------------------------------------------------------------------------------
class Runnable(ABC):
    def invoke(self, input: Any, config: Any = None, **kw) -> Any:
        raise NotImplementedError

    def __or__(self, other):
        return RunnableSequence(self, coerce_to_runnable(other))

    def __ror__(self, other):
        return RunnableSequence(coerce_to_runnable(other), self)


class RunnableLambda(Runnable):
    def __init__(self, fn: Callable):
        self.fn = fn

    def invoke(self, input, config=None, **kw):
        return self.fn(input)


class RunnablePassthrough(Runnable):
    def invoke(self, input, config=None, **kw):
        return input


class RunnableParallel(Runnable):
    def __init__(self, steps: Dict[str, Any]):
        self.steps = {k: coerce_to_runnable(v) for k, v in steps.items()}

    def invoke(self, input, config=None, **kw):
        return {k: r.invoke(input) for k, r in self.steps.items()}


class RunnableSequence(Runnable):
    def __init__(self, first: Runnable, last: Runnable):
        self.first, self.last = first, last

    def invoke(self, input, config=None, **kw):
        return self.last.invoke(self.first.invoke(input))


def coerce_to_runnable(x) -> Runnable:
    if isinstance(x, Runnable):
        return x
    if isinstance(x, dict):
        return RunnableParallel(x)
    if callable(x):
        return RunnableLambda(x)
    raise TypeError(f"Expected a Runnable, callable or dict. Instead got an unsupported type: {type(x)}")


class _Model(BaseModel):
    model_config = ConfigDict(arbitrary_types_allowed=True)


# ---- langchain_core.documents ---------------------------------------------------------------------------------
class Document(_Model):
    page_content: str
    metadata: dict = Field(default_factory=dict)

    def __init__(self, page_content: str = "", **kw):
        super().__init__(page_content=page_content, **kw)


class BaseDocumentCompressor(_Model, ABC):
    @abstractmethod
    def compress_documents(self, documents: Sequence[Document], query: str, callbacks: Any = None) -> Sequence[Document]: ...


# ---- langchain_core.embeddings / retrievers / vectorstores ----------------------------------------------------------
class Embeddings(ABC):
    @abstractmethod
    def embed_documents(self, texts: List[str]) -> List[List[float]]: ...

    @abstractmethod
    def embed_query(self, text: str) -> List[float]: ...


class BaseRetriever(_Model, Runnable, ABC):
    tags: Optional[List[str]] = None
    metadata: Optional[Dict[str, Any]] = None

    def invoke(self, input: str, config: Any = None, **kw) -> List[Document]:
        return self._get_relevant_documents(input, run_manager=None)

    def get_relevant_documents(self, query: str, **kw) -> List[Document]:
        return self.invoke(query)

    @abstractmethod
    def _get_relevant_documents(self, query: str, *, run_manager: Any) -> List[Document]: ...


class VectorStore(ABC):
    @abstractmethod
    def add_texts(self, texts, metadatas=None, **kw) -> List[str]: ...

    @property
    def embeddings(self) -> Optional[Embeddings]:
        return None

    def add_documents(self, documents: List[Document], **kw) -> List[str]:
        return self.add_texts([d.page_content for d in documents], [d.metadata for d in documents], **kw)

    @abstractmethod
    def similarity_search(self, query: str, k: int = 4, **kw) -> List[Document]: ...

    @classmethod
    def from_documents(cls, documents: List[Document], embedding: Embeddings, **kw):
        return cls.from_texts([d.page_content for d in documents], embedding, metadatas=[d.metadata for d in documents], **kw)

    @classmethod
    @abstractmethod
    def from_texts(cls, texts, embedding, metadatas=None, **kw): ...

    def as_retriever(self, **kw) -> "VectorStoreRetriever":
        return VectorStoreRetriever(vectorstore=self, **kw)


class VectorStoreRetriever(BaseRetriever):
    vectorstore: VectorStore
    search_type: str = "similarity"
    search_kwargs: dict = Field(default_factory=dict)
    allowed_search_types: ClassVar[Sequence[str]] = ("similarity", "similarity_score_threshold", "mmr")

    def _get_relevant_documents(self, query: str, *, run_manager: Any = None) -> List[Document]:
        if self.search_type == "similarity":
            return self.vectorstore.similarity_search(query, **self.search_kwargs)
        if self.search_type == "mmr":
            return self.vectorstore.max_marginal_relevance_search(query, **self.search_kwargs)
        raise ValueError(f"search_type of {self.search_type} not allowed.")


# ---- langchain.retrievers ---------------------------------------------------------------------------------------
class EnsembleRetriever(BaseRetriever):
    """weighted Reciprocal Rank Fusion (c = 60), de-duplicated on page_content -- langchain 0.2.11 semantics."""
    retrievers: List[Runnable]
    weights: List[float]
    c: int = 60

    def _get_relevant_documents(self, query: str, *, run_manager: Any = None) -> List[Document]:
        lists = [r.invoke(query) for r in self.retrievers]
        score: Dict[str, float] = {}
        first: Dict[str, Document] = {}
        for docs, w in zip(lists, self.weights):
            for rank, d in enumerate(docs, start=1):
                if d.page_content not in first:
                    first[d.page_content], score[d.page_content] = d, 0.0
                score[d.page_content] += w / (rank + self.c)
        return sorted(first.values(), key=lambda d: score[d.page_content], reverse=True)


class ContextualCompressionRetriever(BaseRetriever):
    base_compressor: BaseDocumentCompressor
    base_retriever: Runnable

    def _get_relevant_documents(self, query: str, *, run_manager: Any = None) -> List[Document]:
        docs = self.base_retriever.invoke(query)
        return list(self.base_compressor.compress_documents(docs, query, callbacks=None)) if docs else []


class BaseCrossEncoder(ABC):
    @abstractmethod
    def score(self, text_pairs: List[tuple]) -> List[float]: ...


class CommunityBaseCrossEncoder(ABC):       # langchain_community keeps its own copy of the interface
    @abstractmethod
    def score(self, text_pairs: List[tuple]) -> List[float]: ...


class HuggingFaceCrossEncoder(CommunityBaseCrossEncoder):
    def __init__(self, model_name: str = "", **kw):
        raise RuntimeError("HuggingFaceCrossEncoder is not available offline: bind MI355XCrossEncoder (INTEGRATION.md)")

    def score(self, text_pairs):
        raise NotImplementedError


class FlashrankRerank(BaseDocumentCompressor):
    top_n: int = 3

    def compress_documents(self, documents, query, callbacks=None):
        return list(documents)[: self.top_n]


class BM25Retriever(BaseRetriever):
    """Tiny lexical retriever with `from_texts` (token-overlap score; stands in for rank_bm25)."""
    docs: List[Document] = Field(default_factory=list)
    k: int = 4

    @classmethod
    def from_texts(cls, texts, metadatas=None, **kw):
        metadatas = metadatas or [{} for _ in texts]
        return cls(docs=[Document(page_content=t, metadata=m) for t, m in zip(texts, metadatas)], **kw)

    def _get_relevant_documents(self, query: str, *, run_manager: Any = None) -> List[Document]:
        qs = set(query.lower().split())
        scored = sorted(enumerate(self.docs), key=lambda p: (-len(qs & set(p[1].page_content.lower().split())), p[0]))
        return [d for _, d in scored[: self.k]]


class SemanticChunker:
    def __init__(self, embeddings: Embeddings, **kw):
        if not isinstance(embeddings, Embeddings):
            raise TypeError("embeddings must be a langchain_core Embeddings")
        self.embeddings = embeddings


class _Unavailable:
    def __init__(self, *a, **kw):
        raise RuntimeError(f"{type(self).__name__} is not available offline")

    @classmethod
    def from_documents(cls, *a, **kw):
        raise RuntimeError(f"{cls.__name__} is not available offline")


def _mod(name: str, **attrs) -> types.ModuleType:
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    m.__path__ = []          # behave like a package so `import a.b.c` walks through
    sys.modules[name] = m
    parent, _, child = name.rpartition(".")
    if parent and parent in sys.modules:
        setattr(sys.modules[parent], child, m)
    return m


def install():
    """Register the stand-in modules.  Idempotent."""
    if "langchain_core" in sys.modules and getattr(sys.modules["langchain_core"], "__lcstub__", False):
        return
    unavailable = lambda n: type(n, (_Unavailable,), {})  # noqa: E731
    _mod("langchain_core", __lcstub__=True)
    _mod("langchain_core.callbacks", Callbacks=Any, CallbackManagerForRetrieverRun=Any)
    _mod("langchain_core.documents", Document=Document, BaseDocumentCompressor=BaseDocumentCompressor)
    _mod("langchain_core.documents.base", Document=Document)
    _mod("langchain_core.embeddings", Embeddings=Embeddings)
    _mod("langchain_core.retrievers", BaseRetriever=BaseRetriever)
    _mod("langchain_core.vectorstores", VectorStore=VectorStore, VectorStoreRetriever=VectorStoreRetriever)
    _mod("langchain_core.runnables", Runnable=Runnable, RunnableLambda=RunnableLambda, RunnableParallel=RunnableParallel,
         RunnablePassthrough=RunnablePassthrough, RunnableSequence=RunnableSequence)
    _mod("langchain")
    _mod("langchain.retrievers", ContextualCompressionRetriever=ContextualCompressionRetriever, EnsembleRetriever=EnsembleRetriever)
    _mod("langchain.retrievers.document_compressors", FlashrankRerank=FlashrankRerank)
    _mod("langchain.retrievers.document_compressors.cross_encoder", BaseCrossEncoder=BaseCrossEncoder)
    _mod("langchain.prompts", ChatPromptTemplate=unavailable("ChatPromptTemplate"), PromptTemplate=unavailable("PromptTemplate"))
    _mod("langchain.schema")
    _mod("langchain.schema.runnable", RunnablePassthrough=RunnablePassthrough)
    _mod("langchain_community")
    _mod("langchain_community.cross_encoders", HuggingFaceCrossEncoder=HuggingFaceCrossEncoder, BaseCrossEncoder=CommunityBaseCrossEncoder)
    _mod("langchain_community.cross_encoders.base", BaseCrossEncoder=CommunityBaseCrossEncoder)
    _mod("langchain_community.document_loaders", **{n: unavailable(n) for n in (
        "CSVLoader", "DirectoryLoader", "Docx2txtLoader", "JSONLoader", "PyPDFDirectoryLoader", "PyPDFLoader", "TextLoader",
        "UnstructuredExcelLoader", "UnstructuredPowerPointLoader")})
    _mod("langchain_community.retrievers", BM25Retriever=BM25Retriever)
    _mod("langchain_experimental")
    _mod("langchain_experimental.text_splitter", SemanticChunker=SemanticChunker)
    _mod("langchain_milvus")
    _mod("langchain_milvus.vectorstores", Milvus=unavailable("Milvus"))
    _mod("langchain_postgres")
    _mod("langchain_postgres.vectorstores", PGVector=unavailable("PGVector"))
    _mod("langchain_text_splitters", RecursiveCharacterTextSplitter=unavailable("RecursiveCharacterTextSplitter"))
    if "lxml" not in sys.modules:
        _mod("lxml", etree=types.SimpleNamespace())
    ext = _mod("psycopg2")
    _mod("psycopg2.extras")
    _mod("psycopg2.extensions", connection=object, cursor=object)
    ext.connect = lambda *a, **kw: (_ for _ in ()).throw(RuntimeError("no postgres offline"))

"""LangChain base classes for the drop-in objects -- the real ones when importable, shims otherwise.

The reference hands our objects to pydantic-validated LangChain containers:
  EnsembleRetriever(retrievers=[sparse, dense])                       server/RAGHelper.py:501-503
  ContextualCompressionRetriever(base_compressor=, base_retriever=)   server/RAGHelper.py:487-490
  ScoredCrossEncoderReranker.model: BaseCrossEncoder                  server/ScoredCrossEncoderReranker.py:15
  SemanticChunker(self.embeddings, ...)                               server/RAGHelper.py:336-341
  {"docs": retriever, ...} | LLMChain                                 server/RAGHelper_local.py:254-258
so every object must BE an instance of the corresponding LangChain type whenever LangChain is installed.  This module
resolves those types once; `ragmeup_amd/*.py` subclass whatever it exports.  Without LangChain (the build container)
the shims below provide the same constructor/field/`invoke` behaviour the call sites rely on, so the classes are
written once against one surface.
"""
from __future__ import annotations

import typing
from abc import ABC, abstractmethod
from typing import Any, Callable, List, Optional, Sequence

# ---------------------------------------------------------------------------------------------------------------
# real LangChain, if present
# ---------------------------------------------------------------------------------------------------------------
try:
    from langchain_core.documents import BaseDocumentCompressor, Document  # type: ignore
    from langchain_core.embeddings import Embeddings  # type: ignore
    from langchain_core.retrievers import BaseRetriever  # type: ignore
    from langchain_core.vectorstores import VectorStore, VectorStoreRetriever  # type: ignore
    HAVE_LANGCHAIN = True
except ImportError:
    HAVE_LANGCHAIN = False


def _cross_encoder_bases() -> tuple:
    """Every importable `BaseCrossEncoder`: langchain (what the reference's reranker field is typed with,
    ScoredCrossEncoderReranker.py:10) and langchain_community (what HuggingFaceCrossEncoder derives from,
    RAGHelper.py:12) are separate classes in some 0.2.x releases -- be an instance of both."""
    found = []
    for mod in ("langchain.retrievers.document_compressors.cross_encoder", "langchain_community.cross_encoders.base",
                "langchain_community.cross_encoders"):
        try:
            m = __import__(mod, fromlist=["BaseCrossEncoder"])
            c = getattr(m, "BaseCrossEncoder", None)
        except ImportError:
            c = None
        if isinstance(c, type) and not any(issubclass(f, c) for f in found):
            found = [f for f in found if not issubclass(c, f)] + [c]
    return tuple(found)


CROSS_ENCODER_BASES: tuple = _cross_encoder_bases() if HAVE_LANGCHAIN else ()


# ---------------------------------------------------------------------------------------------------------------
# shims (LangChain absent)
# ---------------------------------------------------------------------------------------------------------------
class RunnableShim:
    """``.invoke(x)`` and ``a | b`` -- enough for `retriever | RAGHelper.format_documents`
    (server/RAGHelper_local.py:158)."""

    def invoke(self, input, config: Any = None, **kw):
        raise NotImplementedError

    def __or__(self, other: Callable | "RunnableShim"):
        first = self

        class _Seq(RunnableShim):
            def invoke(self, input, config=None, **kw):
                y = first.invoke(input)
                return other.invoke(y) if hasattr(other, "invoke") else other(y)

        return _Seq()


class _FieldModel:
    """Keyword-initialised object with the slice of pydantic behaviour the reference's classes rely on: annotated
    fields with class-level defaults, `Config.extra = "forbid"`, required fields, isinstance checks for class-typed
    fields (`arbitrary_types_allowed`)."""

    def __init__(self, **data):
        cls = type(self)
        try:
            hints = typing.get_type_hints(cls)
        except Exception:  # noqa: BLE001 - an unresolvable annotation only loses the isinstance check
            hints = {k: Any for c in reversed(cls.__mro__) for k in getattr(c, "__annotations__", {})}
        fields = {k: t for k, t in hints.items() if not k.startswith("_") and typing.get_origin(t) is not typing.ClassVar}
        cfg = getattr(cls, "Config", None)
        extra = sorted(set(data) - set(fields))
        if extra and getattr(cfg, "extra", "ignore") == "forbid":
            raise TypeError(f"{cls.__name__}: extra fields not permitted: {extra}")
        for name, tp in fields.items():
            if name in data:
                val = data[name]
            elif hasattr(cls, name):
                val = getattr(cls, name)
                if isinstance(val, (list, dict, set)):
                    val = type(val)(val)
            else:
                raise TypeError(f"{cls.__name__}: field required: {name}")
            if isinstance(tp, type) and tp is not Any and val is not None and not isinstance(val, tp):
                if tp in (int, float, str, bool):
                    val = tp(val)
                else:
                    raise TypeError(f"{cls.__name__}.{name}: instance of {tp.__name__} expected, got {type(val).__name__}")
            object.__setattr__(self, name, val)
        for name in extra:
            object.__setattr__(self, name, data[name])


if not HAVE_LANGCHAIN:

    class Document:  # type: ignore[no-redef]
        """page_content + metadata, with the pydantic-v1 style ``copy(update=...)`` the reference's reranker uses
        (server/ScoredCrossEncoderReranker.py:45)."""

        __slots__ = ("page_content", "metadata")

        def __init__(self, page_content: str = "", metadata: dict | None = None, **kw):
            self.page_content = page_content
            self.metadata = dict(metadata) if metadata else {}

        def copy(self, update: dict | None = None):
            d = Document(self.page_content, dict(self.metadata))
            for k, v in (update or {}).items():
                setattr(d, k, v)
            return d

        def __repr__(self):
            return f"Document(page_content={self.page_content!r}, metadata={self.metadata!r})"

        def __eq__(self, other):
            return (isinstance(other, Document) and self.page_content == other.page_content
                    and self.metadata == other.metadata)

    class Embeddings(ABC):  # type: ignore[no-redef]
        @abstractmethod
        def embed_documents(self, texts: List[str]) -> List[List[float]]: ...

        @abstractmethod
        def embed_query(self, text: str) -> List[float]: ...

    class BaseRetriever(_FieldModel, RunnableShim):  # type: ignore[no-redef]
        """`invoke(query)` -> `_get_relevant_documents(query, run_manager=None)`, as langchain_core's BaseRetriever."""

        def invoke(self, input: str, config: Any = None, **kw) -> list:
            return self._get_relevant_documents(input, run_manager=None)

        def get_relevant_documents(self, query: str, **kw) -> list:   # legacy entry point (EnsembleRetriever 0.2.x)
            return self.invoke(query)

        def _get_relevant_documents(self, query: str, *, run_manager: Any = None) -> list:
            raise NotImplementedError

    class VectorStore(ABC):  # type: ignore[no-redef]
        """Only what our store inherits: `add_documents` -> `add_texts`, `from_documents` -> `from_texts`."""

        @property
        def embeddings(self):
            return None

        def add_documents(self, documents: list, **kw) -> list:
            return self.add_texts([d.page_content for d in documents], [d.metadata for d in documents], **kw)

        @classmethod
        def from_documents(cls, documents: list, embedding: Any, **kw):
            return cls.from_texts([d.page_content for d in documents], embedding,
                                  metadatas=[d.metadata for d in documents], **kw)

    class VectorStoreRetriever(BaseRetriever):  # type: ignore[no-redef]
        vectorstore: Any
        search_type: str = "similarity"
        search_kwargs: dict = {}
        tags: Optional[list] = None
        allowed_search_types: typing.ClassVar[tuple] = ("similarity", "similarity_score_threshold", "mmr")

        def __init__(self, **data):
            super().__init__(**data)
            if self.search_type not in self.allowed_search_types:
                raise ValueError(f"search_type of {self.search_type} not allowed. Valid values are: {self.allowed_search_types}")
            if self.search_type == "similarity_score_threshold":
                thr = self.search_kwargs.get("score_threshold")
                if thr is None or not isinstance(thr, float):
                    raise ValueError("`score_threshold` is not specified with a float value(0~1) in `search_kwargs`.")

    class BaseDocumentCompressor(_FieldModel, ABC):  # type: ignore[no-redef]
        @abstractmethod
        def compress_documents(self, documents: Sequence[Document], query: str, callbacks: Any = None) -> Sequence[Document]: ...

        async def acompress_documents(self, documents, query, callbacks=None):
            return self.compress_documents(documents, query, callbacks)


if not CROSS_ENCODER_BASES:

    class BaseCrossEncoder(ABC):
        """`score(text_pairs) -> list[float]`.  Without LangChain anything with a `score` method qualifies (the field
        check in the reranker is then duck-typed, as the reference's call site is)."""

        @abstractmethod
        def score(self, text_pairs: List[tuple]) -> List[float]: ...

        @classmethod
        def __subclasshook__(cls, other):
            if cls is BaseCrossEncoder:
                return callable(getattr(other, "score", None)) or NotImplemented
            return NotImplemented

    CROSS_ENCODER_BASES = (BaseCrossEncoder,)
else:
    BaseCrossEncoder = CROSS_ENCODER_BASES[0]

Callbacks = Any

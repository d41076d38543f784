"""Document + minimal Runnable shims.

The reference passes langchain_core Documents / Runnables around (server/RAGHelper.py,
server/RAGHelper_local.py:157-159,254-258).  When langchain_core is importable we use its classes so
our objects compose with the reference's LCEL chains; otherwise these duck-typed shims provide exactly
what the reference's call sites touch: ``page_content``, ``metadata``, ``copy(update=...)``,
``invoke(str)``, ``|`` composition and membership in an EnsembleRetriever-like list.
"""
from __future__ import annotations

from typing import Any, Callable

try:  # pragma: no cover - langchain is not installed in the build container
    from langchain_core.documents import Document  # type: ignore
    HAVE_LANGCHAIN = True
except Exception:  # noqa: BLE001
    HAVE_LANGCHAIN = False

    class Document:  # type: ignore[no-redef]
        """page_content + metadata, with the pydantic-v1 style ``copy(update=...)`` the reference's
        reranker uses (server/ScoredCrossEncoderReranker.py:45)."""

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


class RunnableShim:
    """``.invoke(x)`` and ``a | b`` -- enough for `retriever | RAGHelper.format_documents`
    (server/RAGHelper_local.py:158) when langchain_core is absent."""

    def invoke(self, x, config: Any = None, **kw):
        raise NotImplementedError

    def __or__(self, other: Callable | "RunnableShim"):
        first = self

        class _Seq(RunnableShim):
            def invoke(self, x, config=None, **kw):
                y = first.invoke(x)
                return other.invoke(y) if hasattr(other, "invoke") else other(y)

        return _Seq()

    # LangChain's BaseRetriever legacy entry point, still used by EnsembleRetriever in 0.2.x
    def get_relevant_documents(self, query: str, **kw):
        return self.invoke(query)

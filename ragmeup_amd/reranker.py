"""ScoredCrossEncoderReranker -- drop-in for server/ScoredCrossEncoderReranker.py:12-45.

Same fields (``model``, ``top_n=3``), same ``compress_documents(documents, query, callbacks=None)``
contract: score every (query, page_content) pair with ``model.score``, sort descending with Python's
stable ``sorted(..., reverse=True)`` (ties keep input order), keep ``top_n`` and return COPIES whose
metadata gains ``relevance_score``.  ``extra="forbid"`` / arbitrary model types as in the reference's
pydantic Config.  The model behind ``.score`` is ours (ragmeup_amd.cross_encoder.MI355XCrossEncoder) but
any object with ``score(list[tuple[str, str]]) -> list[float]`` works (that is all the reference requires
of langchain's BaseCrossEncoder).
"""
from __future__ import annotations

import operator
from typing import Any, Optional, Sequence

from .documents import Document


class ScoredCrossEncoderReranker:
    def __init__(self, model: Any = None, top_n: int = 3, **extra):
        if extra:  # Config.extra = "forbid"
            raise TypeError(f"extra fields not permitted: {sorted(extra)}")
        if model is None or not hasattr(model, "score"):
            raise TypeError("model must provide score(text_pairs) -> list[float]")
        self.model = model
        self.top_n = int(top_n)

    def compress_documents(self, documents: Sequence[Document], query: str,
                           callbacks: Optional[Any] = None) -> Sequence[Document]:
        scores = self.model.score([(query, doc.page_content) for doc in documents])
        docs_with_scores = list(zip(documents, scores))
        result = sorted(docs_with_scores, key=operator.itemgetter(1), reverse=True)
        return [doc.copy(update={"metadata": {**doc.metadata, "relevance_score": score}})
                for doc, score in result[:self.top_n]]

    # async twin that langchain's BaseDocumentCompressor exposes
    async def acompress_documents(self, documents, query, callbacks=None):
        return self.compress_documents(documents, query, callbacks)

"""ScoredCrossEncoderReranker -- drop-in for server/ScoredCrossEncoderReranker.py:12-45.

A `BaseDocumentCompressor` (LangChain's when installed, so `ContextualCompressionRetriever(base_compressor=...)`,
server/RAGHelper.py:487-490, accepts it) with the reference's fields (``model: BaseCrossEncoder``, ``top_n = 3``), its
pydantic Config (``arbitrary_types_allowed``, ``extra = "forbid"``) and its ``compress_documents(documents, query,
callbacks=None)`` contract: score every (query, page_content) pair with ``model.score``, sort descending with Python's
stable ``sorted(..., reverse=True)`` (ties keep input order), keep ``top_n`` and return COPIES whose metadata gains
``relevance_score``.  The model behind ``.score`` is ours (ragmeup_amd.embeddings.MI355XCrossEncoder, itself a
`BaseCrossEncoder`); the reference's own ScoredCrossEncoderReranker accepts that model unchanged as well.
"""
from __future__ import annotations

import operator
from typing import Optional, Sequence

from ._lc import BaseCrossEncoder, BaseDocumentCompressor, Callbacks, Document


class ScoredCrossEncoderReranker(BaseDocumentCompressor):
    model: BaseCrossEncoder
    top_n: int = 3

    class Config:
        arbitrary_types_allowed = True
        extra = "forbid"

    def compress_documents(self, documents: Sequence[Document], query: str,
                           callbacks: Optional[Callbacks] = None) -> Sequence[Document]:
        scores = self.model.score([(query, doc.page_content) for doc in documents])
        docs_with_scores = list(zip(documents, scores))
        result = sorted(docs_with_scores, key=operator.itemgetter(1), reverse=True)
        return [doc.copy(update={"metadata": {**doc.metadata, "relevance_score": score}})
                for doc, score in result[:self.top_n]]

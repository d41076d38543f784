"""Hybrid fusion next to the dense path: the reference wraps its dense and BM25 retrievers in
``EnsembleRetriever(retrievers=[sparse, dense], weights=[0.5, 0.5])`` (server/RAGHelper.py:500-503, :536-538), i.e.
weighted Reciprocal Rank Fusion with c = 60, de-duplicated on ``page_content``.  This is the same fusion as plain host
code so the hybrid retriever works with our dense retriever when langchain is not installed; with langchain present the
reference's own EnsembleRetriever accepts our retriever unchanged (it only calls ``invoke``)."""
from __future__ import annotations

from typing import Any, List, Sequence

from ._lc import BaseRetriever, Document


def weighted_reciprocal_rank(doc_lists: Sequence[Sequence[Document]], weights: Sequence[float], c: int = 60) -> list[Document]:
    """score(doc) = sum_i w_i / (rank_i(doc) + c), ranks from 1; documents are identified by page_content, the first
    occurrence (in list order) represents them; result sorted by score, ties keep first-seen order."""
    if len(doc_lists) != len(weights):
        raise ValueError("Number of rank lists must be equal to the number of weights.")
    score: dict[str, float] = {}
    first: dict[str, Document] = {}
    for docs, w in zip(doc_lists, weights):
        for rank, d in enumerate(docs, start=1):
            key = d.page_content
            if key not in first:
                first[key] = d
                score[key] = 0.0
            score[key] += w / (rank + c)
    return sorted(first.values(), key=lambda d: score[d.page_content], reverse=True)


class MI355XEnsembleRetriever(BaseRetriever):
    """`EnsembleRetriever(retrievers=[...], weights=[...])` with the same fields and fusion; a LangChain `BaseRetriever`
    when LangChain is installed.  `batch_invoke` runs every member once per batch where the member supports it."""

    retrievers: List[Any]
    weights: List[float] = []
    c: int = 60

    def _weights(self) -> list[float]:
        return list(self.weights) if self.weights else [1.0 / len(self.retrievers)] * len(self.retrievers)

    def _get_relevant_documents(self, query: str, *, run_manager: Any = None, **kw) -> list[Document]:
        lists = [r.invoke(query) if hasattr(r, "invoke") else r.get_relevant_documents(query) for r in self.retrievers]
        return weighted_reciprocal_rank(lists, self._weights(), self.c)

    def batch_invoke(self, queries: list[str]) -> list[list[Document]]:
        per = [r.batch_invoke(queries) if hasattr(r, "batch_invoke") else [r.invoke(q) for q in queries]
               for r in self.retrievers]
        return [weighted_reciprocal_rank([m[i] for m in per], self._weights(), self.c) for i in range(len(queries))]

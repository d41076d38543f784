"""ragmeup_amd -- MI355X-native retrieval hot path of RAGMeUp (embedding, dense top-k, rerank).

Python host code over hand-written gfx950 HIP kernels behind a C-ABI (include/rmu.h).
"""
from . import _native  # noqa: F401
from .index import FlatIndex, topk_merge  # noqa: F401
from .documents import Document  # noqa: F401
from .reranker import ScoredCrossEncoderReranker  # noqa: F401
from .vectorstore import MI355XVectorStore  # noqa: F401

from .vectorstore import MI355XRetriever  # noqa: F401
from . import factory  # noqa: F401

__all__ = ["FlatIndex", "topk_merge", "Document", "ScoredCrossEncoderReranker", "MI355XVectorStore", "MI355XRetriever",
           "BertEncoder", "MI355XEmbeddings", "MI355XCrossEncoder", "factory"]


def __getattr__(name):   # torch-dependent classes are imported lazily
    if name == "BertEncoder":
        from .bert import BertEncoder
        return BertEncoder
    if name in ("MI355XEmbeddings", "MI355XCrossEncoder"):
        from . import embeddings
        return getattr(embeddings, name)
    raise AttributeError(name)
__version__ = "0.1.0"

"""ragmeup_amd -- MI355X-native retrieval hot path of RAGMeUp (embedding, dense top-k, rerank).

Python host code over hand-written gfx950 HIP kernels behind a C-ABI (include/rmu.h).
"""
from . import _native  # noqa: F401
from .index import FlatIndex, topk_merge  # noqa: F401

__all__ = ["FlatIndex", "topk_merge"]
__version__ = "0.1.0"

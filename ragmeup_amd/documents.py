"""Document / Runnable types used by the host classes: LangChain's own when it is installed, shims otherwise
(resolved once in ragmeup_amd._lc).  The reference passes langchain_core Documents / Runnables around
(server/RAGHelper.py, server/RAGHelper_local.py:157-159,254-258)."""
from __future__ import annotations

from ._lc import HAVE_LANGCHAIN, Document, RunnableShim  # noqa: F401

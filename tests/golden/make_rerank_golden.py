"""Generates tests/golden/rerank_golden.json by RUNNING the reference's own
server/ScoredCrossEncoderReranker.py (imported from /root/reference, never copied).

langchain is not installed here, so the three names that file imports are provided as minimal stand-ins
(a keyword-initialised base class, a Document with pydantic-v1 ``copy(update=...)``); everything the
fixture records -- zip, ``sorted(key=itemgetter(1), reverse=True)``, ``[:top_n]``, the metadata copy -- is
the reference's code executing unmodified.  Run in the build container only:  python tests/golden/make_rerank_golden.py
"""
import importlib.util
import json
import os
import sys
import types

REF = "/root/reference/server/ScoredCrossEncoderReranker.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rerank_golden.json")


class _Doc:
    def __init__(self, page_content, metadata=None):
        self.page_content, self.metadata = page_content, dict(metadata or {})

    def copy(self, update=None):
        d = _Doc(self.page_content, dict(self.metadata))
        for k, v in (update or {}).items():
            setattr(d, k, v)
        return d


class _KwBase:
    def __init__(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


_stub("langchain_core")
_stub("langchain_core.callbacks", Callbacks=object)
_stub("langchain_core.documents", BaseDocumentCompressor=_KwBase, Document=_Doc)
_stub("langchain")
_stub("langchain.retrievers")
_stub("langchain.retrievers.document_compressors")
_stub("langchain.retrievers.document_compressors.cross_encoder", BaseCrossEncoder=object)

spec = importlib.util.spec_from_file_location("ref_reranker", REF)
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)


class _FixedModel:
    def __init__(self, scores):
        self.scores = scores

    def score(self, pairs):
        assert len(pairs) == len(self.scores)
        return list(self.scores)


cases = [
    {"name": "distinct", "scores": [0.1, 0.9, -0.3, 0.5, 0.7], "top_n": 3},
    {"name": "ties_keep_input_order", "scores": [0.5, 0.9, 0.5, 0.1, 0.9, 0.5], "top_n": 4},
    {"name": "top_n_exceeds_docs", "scores": [0.2, 0.4], "top_n": 5},
    {"name": "default_top_n", "scores": [3.0, 1.0, 2.0, 5.0, 4.0], "top_n": None},
    {"name": "empty", "scores": [], "top_n": 3},
    {"name": "negative_logits", "scores": [-7.25, -11.5, -7.25, -0.125, -3.0, -11.5, -8.0, -9.0, -0.125, -2.5, -4.0, -6.0, -5.0, -1.0],
     "top_n": 3},
]
out = []
for c in cases:
    docs = [_Doc(f"passage {i}", {"source": f"f{i % 3}.pdf", "id": f"id{i}", "pk": f"id{i}"}) for i in range(len(c["scores"]))]
    kw = {"model": _FixedModel(c["scores"])}
    if c["top_n"] is not None:
        kw["top_n"] = c["top_n"]
    rr = ref.ScoredCrossEncoderReranker(**kw)
    res = rr.compress_documents(docs, "the query")
    out.append({"name": c["name"], "scores": c["scores"], "top_n": c["top_n"],
                "default_top_n": ref.ScoredCrossEncoderReranker.top_n,
                "result": [{"page_content": d.page_content, "metadata": d.metadata} for d in res],
                "inputs_untouched": all("relevance_score" not in d.metadata for d in docs)})
json.dump({"generated_from": REF, "cases": out}, open(OUT, "w"), indent=1)
print("wrote", OUT, len(out), "cases")

"""What a checkpoint directory says about the model -- read, not assumed.

The reference names its models by hub id and lets the third-party loaders interpret the repository files:
  HuggingFaceEmbeddings(model_name=os.getenv('embedding_model'))      server/RAGHelper_local.py:107-117
      -> sentence_transformers.SentenceTransformer(model_name): `modules.json` lists the pipeline
         (Transformer -> Pooling -> [Normalize]), `sentence_bert_config.json` holds max_seq_length / do_lower_case,
         `1_Pooling/config.json` the pooling mode.  Without `modules.json` sentence-transformers builds
         Transformer + mean Pooling and no Normalize.
  HuggingFaceCrossEncoder(model_name=self.rerank_model)                server/RAGHelper.py:483-486
      -> sentence_transformers.CrossEncoder: AutoModelForSequenceClassification; activation =
         config.sbert_ce_default_activation_function, else Sigmoid when num_labels == 1, else Identity.
Here the same files of a LOCAL directory decide the same things; anything this build cannot run raises instead of
being embedded silently wrong (e.g. the reference's template default `avsolatorio/GIST-small-Embedding-v0`,
server/.env.template:3, is a 12-layer CLS-pooling model: CLS pooling is served, mean is not assumed).
"""
from __future__ import annotations

import json
import os
from dataclasses import dataclass, field
from typing import Optional

POOL_MEAN, POOL_CLS = "mean", "cls"


class UnsupportedCheckpoint(ValueError):
    """The directory describes a model this build does not execute (never a silent approximation)."""


def _load(path: str) -> Optional[dict]:
    if not os.path.exists(path):
        return None
    with open(path, encoding="utf-8") as f:
        return json.load(f)


@dataclass
class BertArch:
    layers: int
    heads: int
    hidden: int
    ffn: int
    max_pos: int
    type_vocab: int
    vocab_size: int
    ln_eps: float
    num_labels: int = 0


@dataclass
class CheckpointSpec:
    path: str
    arch: BertArch
    pooling: str = POOL_MEAN             # bi-encoder head
    normalize: bool = False              # a sentence-transformers Normalize module follows the pooling
    max_seq_length: int = 512            # truncation length incl. [CLS]/[SEP]
    st_lower_case: bool = False          # sentence_bert_config.json do_lower_case (text.lower() before tokenising)
    tok_lower_case: bool = True          # tokenizer_config.json do_lower_case (BertTokenizer's own normaliser)
    activation: str = "identity"         # cross-encoder: "identity" | "sigmoid"
    vocab_file: Optional[str] = None
    weights_file: Optional[str] = None
    notes: list = field(default_factory=list)
    config: dict = field(default_factory=dict, repr=False)   # the raw config.json (head-specific readers look things up in it)


def read_arch(cfg: dict) -> BertArch:
    mt = cfg.get("model_type", "bert")
    if mt != "bert":
        raise UnsupportedCheckpoint(f"model_type={mt!r}: only BERT encoders are built (all-MiniLM / ms-marco-MiniLM / bge-small family)")
    act = cfg.get("hidden_act", "gelu")
    if act != "gelu":
        raise UnsupportedCheckpoint(f"hidden_act={act!r}: the FFN kernels implement GELU(erf)")
    pet = cfg.get("position_embedding_type", "absolute")
    if pet != "absolute":
        raise UnsupportedCheckpoint(f"position_embedding_type={pet!r}: only absolute position embeddings")
    a = BertArch(layers=int(cfg["num_hidden_layers"]), heads=int(cfg["num_attention_heads"]), hidden=int(cfg["hidden_size"]),
                 ffn=int(cfg["intermediate_size"]), max_pos=int(cfg.get("max_position_embeddings", 512)),
                 type_vocab=int(cfg.get("type_vocab_size", 2)), vocab_size=int(cfg["vocab_size"]),
                 ln_eps=float(cfg.get("layer_norm_eps", 1e-12)))
    if (a.hidden, a.heads, a.ffn) != (384, 12, 1536):
        raise UnsupportedCheckpoint(f"hidden/heads/ffn = {a.hidden}/{a.heads}/{a.ffn}: this build's kernels are specialised for 384/12/1536")
    if not 1 <= a.max_pos <= 512:
        raise UnsupportedCheckpoint(f"max_position_embeddings={a.max_pos}: must be in [1, 512]")
    if "id2label" in cfg:
        a.num_labels = len(cfg["id2label"])
    elif "num_labels" in cfg:
        a.num_labels = int(cfg["num_labels"])
    return a


def _common(path: str) -> CheckpointSpec:
    if not os.path.isdir(path):
        raise FileNotFoundError(f"{path!r}: a local checkpoint directory is required (hub ids cannot be downloaded here)")
    cfg = _load(os.path.join(path, "config.json"))
    if cfg is None:
        raise FileNotFoundError(f"{path}/config.json is missing")
    spec = CheckpointSpec(path=path, arch=read_arch(cfg))
    for name in ("model.safetensors", "pytorch_model.bin"):
        if os.path.exists(os.path.join(path, name)):
            spec.weights_file = os.path.join(path, name)
            break
    if spec.weights_file is None:
        raise FileNotFoundError(f"{path}: neither model.safetensors nor pytorch_model.bin")
    vocab = os.path.join(path, "vocab.txt")
    spec.vocab_file = vocab if os.path.exists(vocab) else None
    tcfg = _load(os.path.join(path, "tokenizer_config.json")) or {}
    spec.tok_lower_case = bool(tcfg.get("do_lower_case", True))
    tmax = tcfg.get("model_max_length")
    spec.max_seq_length = spec.arch.max_pos if not isinstance(tmax, int) or tmax > 10 ** 6 else min(int(tmax), spec.arch.max_pos)
    spec.config = cfg
    return spec


def read_sentence_transformer(path: str) -> CheckpointSpec:
    """The bi-encoder pipeline sentence-transformers would build from this directory."""
    spec = _common(path)
    modules = _load(os.path.join(path, "modules.json"))
    sb = _load(os.path.join(path, "sentence_bert_config.json")) or {}
    if "max_seq_length" in sb and sb["max_seq_length"] is not None:
        spec.max_seq_length = min(int(sb["max_seq_length"]), spec.arch.max_pos)
    spec.st_lower_case = bool(sb.get("do_lower_case", False))
    if modules is None:                   # plain transformers checkpoint: ST adds mean pooling and nothing else
        spec.notes.append("no modules.json: Transformer + mean Pooling (sentence-transformers' default), no Normalize")
        return spec
    seen_pool = False
    for mod in sorted(modules, key=lambda m: m.get("idx", 0)):
        kind = mod.get("type", "").rsplit(".", 1)[-1]
        if kind == "Transformer":
            continue
        if kind == "Pooling":
            pc = _load(os.path.join(path, mod.get("path", "1_Pooling"), "config.json"))
            if pc is None:
                raise FileNotFoundError(f"{path}/{mod.get('path')}/config.json is missing")
            on = [k for k, v in pc.items() if k.startswith("pooling_mode_") and v is True]
            if on == ["pooling_mode_mean_tokens"]:
                spec.pooling = POOL_MEAN
            elif on == ["pooling_mode_cls_token"]:
                spec.pooling = POOL_CLS
            else:
                raise UnsupportedCheckpoint(f"pooling {on}: only mean-token and CLS-token pooling are built")
            if not pc.get("include_prompt", True):
                raise UnsupportedCheckpoint("pooling include_prompt=false is not built")
            dim = pc.get("word_embedding_dimension", spec.arch.hidden)
            if dim != spec.arch.hidden:
                raise UnsupportedCheckpoint(f"pooling dimension {dim} != hidden size {spec.arch.hidden}")
            seen_pool = True
        elif kind == "Normalize":
            if not seen_pool:
                raise UnsupportedCheckpoint("Normalize before Pooling")
            spec.normalize = True
        else:
            raise UnsupportedCheckpoint(f"sentence-transformers module {mod.get('type')!r} is not built (Transformer, Pooling, Normalize are)")
    if not seen_pool:
        raise UnsupportedCheckpoint("modules.json has no Pooling module")
    return spec


def read_cross_encoder(path: str) -> CheckpointSpec:
    """The CrossEncoder sentence-transformers would build from this directory (num_labels must be 1)."""
    spec = _common(path)
    if spec.arch.num_labels != 1:
        raise UnsupportedCheckpoint(f"num_labels={spec.arch.num_labels}: the reranker head is Linear(hidden, 1)")
    fn = spec.config.get("sbert_ce_default_activation_function")
    if fn is None:
        spec.activation = "sigmoid"       # CrossEncoder: Sigmoid when num_labels == 1 and the config names nothing
    else:
        leaf = str(fn).rsplit(".", 1)[-1].lower()
        if leaf == "identity":
            spec.activation = "identity"
        elif leaf == "sigmoid":
            spec.activation = "sigmoid"
        else:
            raise UnsupportedCheckpoint(f"sbert_ce_default_activation_function={fn!r}: Identity and Sigmoid are built")
    return spec


def load_state(spec: CheckpointSpec) -> dict:
    """name -> array/tensor of the checkpoint's weights."""
    if spec.weights_file.endswith(".safetensors"):
        from safetensors.numpy import load_file
        return load_file(spec.weights_file)
    import torch
    # weights_only: a checkpoint directory is data, never code (torch < 2.6 unpickles arbitrary objects otherwise)
    return torch.load(spec.weights_file, map_location="cpu", weights_only=True)

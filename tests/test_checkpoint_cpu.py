"""ragmeup_amd.checkpoint: what a local checkpoint directory declares (the files sentence-transformers / CrossEncoder would
interpret for the reference: server/RAGHelper_local.py:107-117, server/RAGHelper.py:483-486, server/.env.template:3) is read,
and anything this build cannot execute raises instead of being embedded silently wrong.  CPU only: directories are produced
with transformers' own save_pretrained."""
import json
import os

import pytest

from ragmeup_amd import checkpoint as C
from tests.helpers import write_ce_checkpoint, write_st_checkpoint


def test_all_minilm_style_directory(tmp_path):
    d = str(tmp_path / "minilm")
    write_st_checkpoint(d, pooling="mean", normalize=True, max_seq_length=256, layers=2)
    s = C.read_sentence_transformer(d)
    assert (s.pooling, s.normalize, s.max_seq_length) == ("mean", True, 256)
    assert (s.arch.layers, s.arch.hidden, s.arch.heads, s.arch.ffn) == (2, 384, 12, 1536)
    assert s.vocab_file.endswith("vocab.txt") and s.weights_file.endswith("model.safetensors") and s.tok_lower_case
    assert set(C.load_state(s)) >= {"embeddings.word_embeddings.weight", "encoder.layer.1.output.LayerNorm.bias"}


def test_gist_small_style_directory_cls_pooling_12_layers(tmp_path):
    """.env.template:3 names avsolatorio/GIST-small-Embedding-v0 (bge-small derivative): CLS pooling + Normalize, 512."""
    d = str(tmp_path / "gist")
    write_st_checkpoint(d, pooling="cls", normalize=True, max_seq_length=512, layers=1)
    s = C.read_sentence_transformer(d)
    assert (s.pooling, s.normalize, s.max_seq_length) == ("cls", True, 512)


def test_plain_transformers_directory_gets_st_defaults(tmp_path):
    d = str(tmp_path / "plain")
    write_st_checkpoint(d, layers=1, st_files=False)
    s = C.read_sentence_transformer(d)
    assert (s.pooling, s.normalize, s.max_seq_length) == ("mean", False, 512)      # ST: mean pooling, no Normalize, tokenizer max
    assert s.notes


def test_no_normalize_module(tmp_path):
    d = str(tmp_path / "nonorm")
    write_st_checkpoint(d, normalize=False, layers=1)
    assert C.read_sentence_transformer(d).normalize is False


@pytest.mark.parametrize("edit,msg", [
    (lambda c: c.update(hidden_size=768, intermediate_size=3072), "384/12/1536"),
    (lambda c: c.update(model_type="roberta"), "model_type"),
    (lambda c: c.update(hidden_act="relu"), "hidden_act"),
    (lambda c: c.update(position_embedding_type="relative_key"), "position_embedding_type"),
    (lambda c: c.update(max_position_embeddings=1024), "max_position_embeddings"),
])
def test_unsupported_architectures_raise(tmp_path, edit, msg):
    d = str(tmp_path / "bad")
    write_st_checkpoint(d, layers=1)
    p = os.path.join(d, "config.json")
    c = json.load(open(p))
    edit(c)
    json.dump(c, open(p, "w"))
    with pytest.raises(C.UnsupportedCheckpoint, match=msg):
        C.read_sentence_transformer(d)


def test_unsupported_pooling_and_modules_raise(tmp_path):
    d = str(tmp_path / "maxpool")
    write_st_checkpoint(d, pooling="max", layers=1)
    with pytest.raises(C.UnsupportedCheckpoint, match="pooling"):
        C.read_sentence_transformer(d)
    d2 = str(tmp_path / "dense")
    write_st_checkpoint(d2, layers=1)
    mods = json.load(open(os.path.join(d2, "modules.json")))
    mods.insert(2, {"idx": 2, "name": "2", "path": "2_Dense", "type": "sentence_transformers.models.Dense"})
    json.dump(mods, open(os.path.join(d2, "modules.json"), "w"))
    with pytest.raises(C.UnsupportedCheckpoint, match="Dense"):
        C.read_sentence_transformer(d2)


def test_cross_encoder_directory_and_activation(tmp_path):
    d = str(tmp_path / "ce")
    write_ce_checkpoint(d, layers=1, activation="identity")
    s = C.read_cross_encoder(d)
    assert s.arch.num_labels == 1 and s.activation == "identity" and s.max_seq_length == 512
    d2 = str(tmp_path / "ce_default")
    write_ce_checkpoint(d2, layers=1, activation=None)
    assert C.read_cross_encoder(d2).activation == "sigmoid"       # CrossEncoder's default for num_labels == 1
    d3 = str(tmp_path / "bi")
    write_st_checkpoint(d3, layers=1)
    with pytest.raises(C.UnsupportedCheckpoint, match="num_labels"):
        C.read_cross_encoder(d3)


def test_missing_directory_and_files(tmp_path):
    with pytest.raises(FileNotFoundError):
        C.read_sentence_transformer(str(tmp_path / "absent"))
    d = tmp_path / "empty"
    d.mkdir()
    with pytest.raises(FileNotFoundError):
        C.read_sentence_transformer(str(d))

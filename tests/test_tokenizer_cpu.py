"""CPU: the C++ WordPiece tokenizer against transformers' BertTokenizer (the implementation the reference's
dependencies use) on a synthetic vocabulary -- no real vocab.txt exists offline."""
import random

import numpy as np
import pytest

from ragmeup_amd.tokenizer import WordPieceTokenizer

WORDS = ["the", "quick", "brown", "fox", "jump", "##s", "##ed", "##ing", "over", "lazy", "dog", "retrieval", "augment",
         "##ation", "gen", "##era", "##tion", "vector", "store", "query", "docu", "##ment", "rank", "re", "##rank", "a",
         "b", "c", "##a", "##b", "##c", "un", "##aff", "##able", "cafe", "naive", "resume", "uber", "strasse", "hello",
         "world", "##ly", "1", "2", "##3", "2024", ",", ".", "!", "?", "(", ")", "-", "'", "\"", ":", ";", "/", "中", "文",
         "x", "##x", "y", "##y", "z", "##z"]


@pytest.fixture(scope="module")
def vocab(tmp_path_factory):
    p = tmp_path_factory.mktemp("tok") / "vocab.txt"
    toks = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"] + WORDS
    p.write_text("\n".join(toks) + "\n", encoding="utf-8")
    return str(p)


@pytest.fixture(scope="module")
def pair(vocab, librmu):
    from transformers import BertTokenizer
    toks = open(vocab, encoding="utf-8").read().split("\n")[:-1]
    # transformers 5.x: BertTokenizer is the `tokenizers`-backed implementation and takes the vocabulary as a dict
    return WordPieceTokenizer(vocab), BertTokenizer(vocab={t: i for i, t in enumerate(toks)}, do_lower_case=True)


def _rand_text(rng, n):
    parts = []
    for _ in range(n):
        r = rng.random()
        if r < 0.55:
            parts.append(rng.choice(["the", "quick", "brown", "fox", "jumps", "jumped", "jumping", "over", "lazy", "dog",
                                     "retrieval", "augmentation", "generation", "vector", "store", "query", "document",
                                     "rerank", "unaffable", "Hello", "WORLD", "worldly", "abcabc", "xyzzy", "2024", "123"]))
        elif r < 0.7:
            parts.append(rng.choice(["café", "naïve", "résumé", "Über", "ÀÉÎÕÜ", "straße"]))
        elif r < 0.8:
            parts.append(rng.choice([",", ".", "!?", "(a)", "b-c", "it's", "\"quoted\"", "a/b:c;"]))
        elif r < 0.88:
            parts.append(rng.choice(["中文", "a中b", "\tTab\n", "zero​width", "nb sp", "qqqq", "w" * 120]))
        else:
            parts.append("".join(rng.choice("abcxyz") for _ in range(rng.randint(1, 9))))
    return rng.choice([" ", "  ", "\n"]).join(parts)


def test_single_sentences_match_transformers(pair):
    mine, hf = pair
    rng = random.Random(0)
    texts = [_rand_text(rng, rng.randint(1, 40)) for _ in range(400)] + ["", "   ", "UNKNOWNWORD", "the", "a" * 300]
    ids, tt, lens = mine.encode(texts, max_len=64)
    for i, t in enumerate(texts):
        want = hf(t, truncation=True, max_length=64, padding=False)["input_ids"]
        assert ids[i, :lens[i]].tolist() == want, (t, ids[i, :lens[i]].tolist(), want)
        assert (ids[i, lens[i]:] == 0).all() and (tt[i] == 0).all()        # [PAD] = 0 in this vocab


def test_pairs_match_transformers(pair):
    mine, hf = pair
    rng = random.Random(1)
    qa = [_rand_text(rng, rng.randint(1, 12)) for _ in range(200)]
    pb = [_rand_text(rng, rng.randint(1, 60)) for _ in range(200)]
    for ml in (8, 17, 48, 49):
        ids, tt, lens = mine.encode(qa, pb, max_len=ml)
        for i in range(200):
            e = hf(qa[i], pb[i], truncation="longest_first", max_length=ml, padding=False, return_token_type_ids=True)
            assert ids[i, :lens[i]].tolist() == e["input_ids"], (ml, qa[i], pb[i])
            assert tt[i, :lens[i]].tolist() == e["token_type_ids"]
    # and the long-first / short-second orientation
    ids, tt, lens = mine.encode(pb, qa, max_len=24)
    for i in range(200):
        e = hf(pb[i], qa[i], truncation="longest_first", max_length=24, padding=False, return_token_type_ids=True)
        assert ids[i, :lens[i]].tolist() == e["input_ids"]


def test_hf_style_call_and_errors(pair, vocab, tmp_path):
    mine, hf = pair
    out = mine(["the quick fox", "lazy dog"], truncation=True, max_length=16)
    assert out["input_ids"] == hf(["the quick fox", "lazy dog"], truncation=True, max_length=16)["input_ids"]
    from ragmeup_amd._native import RmuError
    with pytest.raises(RmuError):
        WordPieceTokenizer(str(tmp_path / "nope.txt"))
    bad = tmp_path / "bad.txt"; bad.write_text("only\nwords\n")
    with pytest.raises(RmuError):
        WordPieceTokenizer(str(bad))

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


def test_unicode_normalisation_matches_the_tokenizers_library(tmp_path, librmu):
    """Whole-code-space behaviour (tables generated from `tokenizers`): every character of every normalised word is in the
    vocabulary (as 'c' and '##c'), so the id sequences expose the exact normalised text -- nothing hides behind [UNK]."""
    from tokenizers.normalizers import BertNormalizer
    from tokenizers.pre_tokenizers import BertPreTokenizer
    from transformers import BertTokenizer
    rng = random.Random(7)
    blocks = [(0x20, 0x7E), (0xA0, 0x17F), (0x180, 0x24F), (0x300, 0x36F), (0x370, 0x3FF), (0x400, 0x4FF), (0x590, 0x5FF),
              (0x600, 0x6FF), (0x900, 0x97F), (0xE00, 0xE7F), (0x1100, 0x11FF), (0x1E00, 0x1EFF), (0x2000, 0x206F),
              (0x2070, 0x20CF), (0x2100, 0x214F), (0x3000, 0x303F), (0x3040, 0x30FF), (0x4E00, 0x4E80), (0xAC00, 0xAD00),
              (0xF900, 0xF960), (0xFB00, 0xFB06), (0xFE50, 0xFE6F), (0xFF00, 0xFF5E), (0x1F600, 0x1F640), (0x1D400, 0x1D430),
              (0xE000, 0xE010), (0x200B, 0x200F), (0x80, 0x9F)]
    texts = []
    for _ in range(600):
        parts = []
        for _ in range(rng.randint(1, 12)):
            lo, hi = rng.choice(blocks)
            parts.append("".join(chr(rng.randint(lo, hi)) for _ in range(rng.randint(1, 6))))
        texts.append(rng.choice([" ", "", "  ", "\t"]).join(parts))
    texts += ["İstanbul ÇAĞRI", "ΑΒΓ δοκιμή Ωμέγα", "ПРИВЕТ мир Ёж", "한국어 테스트", "école Ångström", "ﬁnal ﬂow",
              "१२३ हिन्दी", "日本語のテキスト", "a­b soft­hyphen", "x﻿y⁠z", "Ǆ ǅ ǆ ß ẞ", "①②③ ½ ™"]
    nl = BertNormalizer(clean_text=True, handle_chinese_chars=True, strip_accents=True, lowercase=True)
    pt = BertPreTokenizer()
    alphabet = set()
    for t in texts:
        for w, _ in pt.pre_tokenize_str(nl.normalize_str(t)):
            alphabet.update(w)
    toks = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"] + sorted(alphabet) + ["##" + c for c in sorted(alphabet)]
    vp = tmp_path / "vocab.txt"
    vp.write_text("\n".join(toks) + "\n", encoding="utf-8")
    mine = WordPieceTokenizer(str(vp))
    hf = BertTokenizer(vocab={t: i for i, t in enumerate(toks)}, do_lower_case=True)
    ids, _, lens = mine.encode(texts, max_len=128)
    bad = 0
    for i, t in enumerate(texts):
        want = hf(t, truncation=True, max_length=128, padding=False)["input_ids"]
        if ids[i, :lens[i]].tolist() != want:
            bad += 1
            assert "Σ" in t or "σ" in t or "ς" in t, (t, ids[i, :lens[i]].tolist(), want)    # only the final-sigma rule may differ
    assert bad <= 3
    # cased checkpoints: no lower-casing, no accent stripping
    hfc = BertTokenizer(vocab={t: i for i, t in enumerate(toks)}, do_lower_case=False)
    minec = WordPieceTokenizer(str(vp), do_lower_case=False)
    sample = [t for t in texts if t.isascii() is False][:200]
    ids, _, lens = minec.encode(sample, max_len=128)
    for i, t in enumerate(sample):
        assert ids[i, :lens[i]].tolist() == hfc(t, truncation=True, max_length=128, padding=False)["input_ids"], t


def test_committed_unicode_tables_are_what_the_generator_produces(tmp_path):
    """csrc/wordpiece_tables.h is generated from the `tokenizers` library; a stale or hand-edited header fails here."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = tmp_path / "tables.h"
    subprocess.run([sys.executable, os.path.join(root, "tools", "gen_wordpiece_tables.py"), str(out)], check=True,
                   capture_output=True, timeout=600)
    assert out.read_text() == open(os.path.join(root, "ragmeup_amd", "csrc", "wordpiece_tables.h")).read()


def test_committed_golden_from_transformers(tmp_path, librmu):
    """tests/golden/tokenizer_golden.json was produced by transformers.BertTokenizer (script beside it): singles and pairs,
    several max lengths, lower-cased and cased -- checked without transformers at test time."""
    import json
    import os
    g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tokenizer_golden.json"), encoding="utf-8"))
    vp = tmp_path / "vocab.txt"
    vp.write_text("\n".join(g["vocab"]) + "\n", encoding="utf-8")
    tok = {True: WordPieceTokenizer(str(vp), do_lower_case=True), False: WordPieceTokenizer(str(vp), do_lower_case=False)}
    assert len(g["cases"]) == 300
    for c in g["cases"]:
        ids, tt, lens = tok[c["lower"]].encode([c["a"]], None if c["b"] is None else [c["b"]], max_len=c["max_len"])
        assert ids[0, :lens[0]].tolist() == c["ids"], c
        if c["b"] is not None:
            assert tt[0, :lens[0]].tolist() == c["type_ids"], c


def test_literal_special_tokens_match_transformers(pair):
    """literal '[SEP]' / '[CLS]' / '[MASK]' / '[UNK]' / '[PAD]' in the text are single ids in transformers'
    BertTokenizer (added tokens are cut out of the raw text before normalisation); lower-case look-alikes are not."""
    mine, hf = pair
    texts = ["BERT joins segments with [SEP] and starts with [CLS] .", "[MASK]", "a[SEP]b", "the [MASK]ed dog", "[sep] [Sep] [ SEP ]",
             "[UNK] [PAD] [SEP][SEP]", "x [CLS", "SEP] y", "[[SEP]]", "the quick [MASK] fox [UNK]", "[MASK][MASK] jumps[SEP]over"]
    ids, tt, lens = mine.encode(texts, max_len=48)
    for i, t in enumerate(texts):
        want = hf(t, truncation=True, max_length=48, padding=False)["input_ids"]
        assert ids[i, :lens[i]].tolist() == want, (t, ids[i, :lens[i]].tolist(), want)
    e = hf("what is [MASK] ?", "it is the [SEP] token", truncation="longest_first", max_length=32, return_token_type_ids=True)
    ids, tt, lens = mine.encode(["what is [MASK] ?"], ["it is the [SEP] token"], max_len=32)
    assert ids[0, :lens[0]].tolist() == e["input_ids"] and tt[0, :lens[0]].tolist() == e["token_type_ids"]


def test_blob_entry_matches_the_pointer_entry_and_rejects_miscounted_blobs(pair, librmu):
    """rmu_tok_encode_blob (one NUL-separated buffer per side) is the same tokenizer as rmu_tok_encode (n pointers); a text that
    itself contains NUL goes through the pointer form, where C strings end at the NUL on both paths of the reference too."""
    import ctypes
    from ragmeup_amd import _native as N
    mine, hf = pair
    rng = random.Random(7)
    a = [_rand_text(rng, rng.randint(0, 30)) for _ in range(300)] + ["", "x"]
    b = [_rand_text(rng, rng.randint(0, 30)) for _ in range(300)] + ["y", ""]
    n = len(a)
    ids, tt, lens = mine.encode(a, b, max_len=40)                       # blob path
    arr_a = (ctypes.c_char_p * n)(*[t.encode() for t in a]); arr_b = (ctypes.c_char_p * n)(*[t.encode() for t in b])
    ids2 = np.empty_like(ids); tt2 = np.empty_like(tt); lens2 = np.empty_like(lens)
    N.check(librmu.rmu_tok_encode(mine._h, arr_a, arr_b, n, 40, ids2.ctypes.data, tt2.ctypes.data, lens2.ctypes.data), "tok")
    assert (ids == ids2).all() and (tt == tt2).all() and (lens == lens2).all()
    with_nul = ["the quick\0brown fox", "lazy dog"]
    i3, _, l3 = mine.encode(with_nul, max_len=16)                        # falls back: the C string ends at the NUL
    assert i3[0, :l3[0]].tolist() == hf("the quick", truncation=True, max_length=16)["input_ids"]
    assert i3[1, :l3[1]].tolist() == hf("lazy dog", truncation=True, max_length=16)["input_ids"]
    RMU_E_INVALID = -1                                                   # include/rmu.h
    blob = b"one\0two\0"
    out = np.empty((3, 8), np.int32); ln = np.empty(3, np.int32)
    assert librmu.rmu_tok_encode_blob(mine._h, blob, len(blob), None, 0, 3, 8, out.ctypes.data, None, ln.ctypes.data) == RMU_E_INVALID
    assert librmu.rmu_tok_encode_blob(mine._h, blob, len(blob) - 1, None, 0, 2, 8, out.ctypes.data, None, ln.ctypes.data) == RMU_E_INVALID
    assert librmu.rmu_tok_encode_blob(mine._h, blob, len(blob), None, 0, 2, 8, out.ctypes.data, None, ln.ctypes.data) == 0

"""Generates tests/golden/tokenizer_golden.json: outputs of the real third-party tokenizer the reference's models use
(transformers.BertTokenizer, `tokenizers` backend) on a synthetic vocabulary -- singles and pairs, several max lengths,
lower-cased and cased.  Run where `transformers` is installed; the JSON is committed so the parity test needs nothing.

    python tests/golden/make_tokenizer_golden.py
"""
import json
import os
import random

import tokenizers
import transformers
from transformers import BertTokenizer

WORDS = ["the", "quick", "brown", "fox", "jump", "##s", "##ed", "##ing", "over", "lazy", "dog", "retrieval", "augment", "##ation",
         "gen", "##era", "##tion", "vector", "store", "query", "docu", "##ment", "rank", "re", "##rank", "a", "b", "c", "##a", "##b",
         "##c", "un", "##aff", "##able", "cafe", "naive", "resume", "uber", "strasse", "hello", "world", "##ly", "1", "2", "##3",
         "2024", ",", ".", "!", "?", "(", ")", "-", "'", "\"", ":", ";", "/", "中", "文", "x", "##x", "y", "##y", "z", "##z",
         "привет", "мир", "##ы", "αβγ", "한", "##국", "ᄒ", "##ᅡ", "##ᆫ", "Über", "Café", "Hello", "##O"]
SURFACE = ["the", "quick", "brown", "fox", "jumps", "jumped", "jumping", "over", "lazy", "dog", "retrieval", "augmentation",
           "generation", "vector", "store", "query", "document", "rerank", "unaffable", "Hello", "WORLD", "worldly", "abcabc",
           "xyzzy", "2024", "123", "café", "naïve", "résumé", "Über", "ÀÉÎÕÜ", "straße", ",", ".", "!?", "(a)", "b-c", "it's",
           "\"quoted\"", "a/b:c;", "中文", "a中b", "\tTab\n", "zero​width", "nb sp", "qqqq", "w" * 120, "Привет", "МИР", "миры",
           "ΑΒΓ", "한국", "한", "ｆｕｌｌ", "①", "ﬁx"]


def main():
    toks = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"] + WORDS
    vocab = {t: i for i, t in enumerate(toks)}
    rng = random.Random(20250925)
    cases = []
    for lower in (True, False):
        hf = BertTokenizer(vocab=vocab, do_lower_case=lower)
        for _ in range(90):
            a = rng.choice([" ", "  ", "\n"]).join(rng.choice(SURFACE) for _ in range(rng.randint(1, 40)))
            ml = rng.choice([8, 17, 32, 64])
            cases.append({"lower": lower, "a": a, "b": None, "max_len": ml,
                          "ids": hf(a, truncation=True, max_length=ml, padding=False)["input_ids"]})
        for _ in range(60):
            a = " ".join(rng.choice(SURFACE) for _ in range(rng.randint(1, 14)))
            b = " ".join(rng.choice(SURFACE) for _ in range(rng.randint(1, 50)))
            ml = rng.choice([8, 17, 48, 49])
            e = hf(a, b, truncation="longest_first", max_length=ml, padding=False, return_token_type_ids=True)
            cases.append({"lower": lower, "a": a, "b": b, "max_len": ml, "ids": e["input_ids"], "type_ids": e["token_type_ids"]})
    out = {"generator": f"transformers {transformers.__version__}, tokenizers {tokenizers.__version__}", "vocab": toks, "cases": cases}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tokenizer_golden.json")
    json.dump(out, open(path, "w"), ensure_ascii=False, indent=0)
    print(path, len(cases), "cases", os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()

// wordpiece.hip -- BERT BasicTokenizer + WordPiece on the host (plain C++, compiled into librmu.so; no device code).
//
// Serves the tokenisation step in front of both encoder forwards:
//   sentence-transformers `tokenize` inside HuggingFaceEmbeddings.embed_documents (server/RAGHelper.py:423-434 via
//   RAGHelper_local.py:107-117) and CrossEncoder pair tokenisation inside HuggingFaceCrossEncoder.score
//   (server/RAGHelper.py:483-486).  Restates transformers' BertTokenizer (BasicTokenizer + WordpieceTokenizer):
//   clean text -> pad CJK with spaces -> whitespace split -> lower-case + strip accents -> split punctuation ->
//   greedy longest-match-first WordPiece ("##" continuations, > 100 chars -> [UNK]); [CLS] a [SEP] (b [SEP]),
//   token types 0/1, single: keep the first max_len-2 tokens; pair: "longest_first" truncation as the `tokenizers`
//   backend computes it.
// Unicode coverage: ASCII exactly; Latin-1 / Latin Extended-A letters are lower-cased and de-accented through a small
// table; general/CJK punctuation and CJK ideograph ranges as in the original.  Other scripts pass through un-folded.
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/rmu.h"

extern "C" void rmu_set_error_(const char* msg);

struct rmu_tok {
    std::unordered_map<std::string, int> vocab;
    int unk = 0, cls = 0, sep = 0, pad = 0;
    bool lower = true;
};

namespace {

typedef uint32_t cp_t;

void decode_utf8(const char* s, std::vector<cp_t>& out) {
    const unsigned char* p = (const unsigned char*)s;
    while (*p) {
        cp_t c;
        int n;
        if (*p < 0x80) { c = *p; n = 1; }
        else if ((*p >> 5) == 6) { c = *p & 31; n = 2; }
        else if ((*p >> 4) == 14) { c = *p & 15; n = 3; }
        else if ((*p >> 3) == 30) { c = *p & 7; n = 4; }
        else { out.push_back(0xFFFD); ++p; continue; }
        bool ok = true;
        for (int i = 1; i < n; ++i) {
            if ((p[i] & 0xC0) != 0x80) { ok = false; break; }
            c = (c << 6) | (p[i] & 63);
        }
        if (!ok) { out.push_back(0xFFFD); ++p; continue; }
        out.push_back(c);
        p += n;
    }
}
void append_utf8(std::string& s, cp_t c) {
    if (c < 0x80) s += (char)c;
    else if (c < 0x800) { s += (char)(0xC0 | (c >> 6)); s += (char)(0x80 | (c & 63)); }
    else if (c < 0x10000) { s += (char)(0xE0 | (c >> 12)); s += (char)(0x80 | ((c >> 6) & 63)); s += (char)(0x80 | (c & 63)); }
    else { s += (char)(0xF0 | (c >> 18)); s += (char)(0x80 | ((c >> 12) & 63)); s += (char)(0x80 | ((c >> 6) & 63)); s += (char)(0x80 | (c & 63)); }
}
bool is_ws(cp_t c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == 0xA0 || c == 0x1680 || (c >= 0x2000 && c <= 0x200A) || c == 0x2028 || c == 0x2029 || c == 0x202F || c == 0x205F || c == 0x3000; }
bool is_control(cp_t c) {
    if (c == '\t' || c == '\n' || c == '\r') return false;
    return c < 0x20 || (c >= 0x7F && c < 0xA0) || c == 0xAD || (c >= 0x200B && c <= 0x200F) || (c >= 0x202A && c <= 0x202E) ||
           (c >= 0x2060 && c <= 0x2064) || c == 0xFEFF;
}
bool is_punct(cp_t c) {
    if ((c >= 33 && c <= 47) || (c >= 58 && c <= 64) || (c >= 91 && c <= 96) || (c >= 123 && c <= 126)) return true;
    if (c == 0xA1 || c == 0xA7 || c == 0xAB || c == 0xB6 || c == 0xB7 || c == 0xBB || c == 0xBF) return true;   // Latin-1 P*
    return (c >= 0x2010 && c <= 0x2027) || (c >= 0x2030 && c <= 0x205E) || (c >= 0x3001 && c <= 0x3003) ||
           (c >= 0x3008 && c <= 0x3011) || (c >= 0x3014 && c <= 0x301F) || (c >= 0xFF01 && c <= 0xFF0F) ||
           (c >= 0xFF1A && c <= 0xFF20) || (c >= 0xFF3B && c <= 0xFF40) || (c >= 0xFF5B && c <= 0xFF65);
}
bool is_cjk(cp_t c) {
    return (c >= 0x4E00 && c <= 0x9FFF) || (c >= 0x3400 && c <= 0x4DBF) || (c >= 0x20000 && c <= 0x2A6DF) ||
           (c >= 0x2A700 && c <= 0x2B73F) || (c >= 0x2B740 && c <= 0x2B81F) || (c >= 0x2B820 && c <= 0x2CEAF) ||
           (c >= 0xF900 && c <= 0xFAFF) || (c >= 0x2F800 && c <= 0x2FA1F);
}
// lower-case + strip accents (NFD, drop Mn) for ASCII, Latin-1 Supplement and Latin Extended-A
cp_t fold(cp_t c) {
    if (c < 0x80) return (c >= 'A' && c <= 'Z') ? c + 32 : c;
    static const char* l1 = "aaaaaa\0ceeeeiiii\0nooooo\0ouuuuy\0\0aaaaaa\0ceeeeiiii\0nooooo\0ouuuuy\0y";   // U+00C0..U+00FF
    if (c >= 0xC0 && c <= 0xFF) {
        const char b = l1[c - 0xC0];
        if (b) return (cp_t)b;
        if (c == 0xC6) return 0xE6;   // AE -> ae (no decomposition)
        if (c == 0xD0) return 0xF0;   // ETH
        if (c == 0xD8) return 0xF8;   // O-stroke
        if (c == 0xDE) return 0xFE;   // THORN
        return c;                      // x, /, ss, ae, eth, o-stroke, thorn stay
    }
    if (c >= 0x100 && c <= 0x17F) {
        static const char* la =
            "aaaaaaccccccccddddeeeeeeeeeegggggggghhhhiiiiiiiiii\0\0jjkk\0lllllllllnnnnnn\0\0\0oooooo\0\0rrrrrrssssssssttttttuuuuuuuuuuuuwwyyyzzzzzz\0";
        const char b = la[c - 0x100];
        if (b) return (cp_t)b;
        if (c == 0x132) return 0x133;
        if (c == 0x141) return 0x142;
        if (c == 0x14A) return 0x14B;
        if (c == 0x152) return 0x153;
        if (c == 0x110 || c == 0x126 || c == 0x166) return c + 1;   // stroked letters: lower-case only
        return c;
    }
    return c;
}

void basic_tokenize(const rmu_tok* tk, const char* text, std::vector<std::string>& out) {
    std::vector<cp_t> cps;
    decode_utf8(text, cps);
    std::vector<cp_t> cur;
    auto flush = [&]() {
        if (cur.empty()) return;
        std::string s;
        for (cp_t c : cur) append_utf8(s, c);
        out.push_back(std::move(s));
        cur.clear();
    };
    for (cp_t c : cps) {
        if (c == 0 || c == 0xFFFD || is_control(c)) continue;
        if (is_ws(c)) { flush(); continue; }
        if (is_cjk(c)) { flush(); cur.push_back(c); flush(); continue; }
        if (tk->lower) {
            if (c >= 0x300 && c <= 0x36F) continue;      // combining marks (Mn) are dropped by strip_accents
            c = fold(c);
        }
        if (is_punct(c)) { flush(); cur.push_back(c); flush(); continue; }
        cur.push_back(c);
    }
    flush();
}

void wordpiece(const rmu_tok* tk, const std::string& word, std::vector<int>& ids) {
    // length in characters
    size_t nchars = 0;
    for (unsigned char ch : word) if ((ch & 0xC0) != 0x80) ++nchars;
    if (nchars > 100) { ids.push_back(tk->unk); return; }
    std::vector<size_t> bounds;   // byte offsets of character starts (+ end)
    for (size_t i = 0; i < word.size(); ++i) if (((unsigned char)word[i] & 0xC0) != 0x80) bounds.push_back(i);
    bounds.push_back(word.size());
    std::vector<int> sub;
    size_t start = 0;
    const size_t n = bounds.size() - 1;
    while (start < n) {
        size_t end = n;
        int found = -1;
        while (start < end) {
            std::string piece = word.substr(bounds[start], bounds[end] - bounds[start]);
            if (start > 0) piece = "##" + piece;
            auto it = tk->vocab.find(piece);
            if (it != tk->vocab.end()) { found = it->second; break; }
            --end;
        }
        if (found < 0) { ids.push_back(tk->unk); return; }
        sub.push_back(found);
        start = end;
    }
    ids.insert(ids.end(), sub.begin(), sub.end());
}

void encode_text(const rmu_tok* tk, const char* text, std::vector<int>& ids) {
    std::vector<std::string> words;
    basic_tokenize(tk, text ? text : "", words);
    for (const std::string& wd : words) wordpiece(tk, wd, ids);
}

}  // namespace

extern "C" int rmu_tok_create(rmu_tok_t** out, const char* vocab_path, int do_lower_case) {
    if (!out || !vocab_path) { rmu_set_error_("rmu_tok_create: null argument"); return RMU_E_INVALID; }
    std::ifstream f(vocab_path);
    if (!f) { rmu_set_error_("rmu_tok_create: cannot open vocabulary file"); return RMU_E_INVALID; }
    auto* tk = new rmu_tok();
    std::string line;
    int idx = 0;
    while (std::getline(f, line)) {
        while (!line.empty() && (line.back() == '\r' || line.back() == '\n')) line.pop_back();
        tk->vocab.emplace(line, idx++);
    }
    auto need = [&](const char* t, int* dst) {
        auto it = tk->vocab.find(t);
        if (it == tk->vocab.end()) return false;
        *dst = it->second;
        return true;
    };
    if (!need("[UNK]", &tk->unk) || !need("[CLS]", &tk->cls) || !need("[SEP]", &tk->sep) || !need("[PAD]", &tk->pad)) {
        delete tk;
        rmu_set_error_("rmu_tok_create: vocabulary lacks [UNK]/[CLS]/[SEP]/[PAD]");
        return RMU_E_INVALID;
    }
    tk->lower = do_lower_case != 0;
    *out = tk;
    return RMU_OK;
}

extern "C" int rmu_tok_free(rmu_tok_t* tk) { delete tk; return RMU_OK; }
extern "C" int rmu_tok_vocab_size(rmu_tok_t* tk) { return tk ? (int)tk->vocab.size() : 0; }

extern "C" int rmu_tok_encode(rmu_tok_t* tk, const char* const* texts_a, const char* const* texts_b, int n, int max_len,
                              int32_t* ids, int32_t* type_ids, int32_t* lens) {
    if (!tk || !texts_a || !ids || !lens || n < 0 || max_len < 3) { rmu_set_error_("rmu_tok_encode: bad argument"); return RMU_E_INVALID; }
    const int nthreads = std::max(1, std::min<int>(n / 64, (int)std::thread::hardware_concurrency()));
    auto work = [&](int lo, int hi) {
        std::vector<int> a, b;
        for (int i = lo; i < hi; ++i) {
            a.clear(); b.clear();
            encode_text(tk, texts_a[i], a);
            const bool pair = texts_b && texts_b[i];
            if (pair) encode_text(tk, texts_b[i], b);
            const int special = pair ? 3 : 2;
            // truncation as the `tokenizers` library does it (AutoTokenizer's default fast backend): single -> keep the
            // head; pair "longest_first" -> only the longer side is cut when that suffices, else both to budget/2 with the
            // odd token going to the longer side (to the second on equal lengths)
            const size_t budget = (size_t)(max_len - special);
            if (a.size() + b.size() > budget) {
                if (!pair) a.resize(budget);
                else {
                    size_t n1 = a.size(), n2 = b.size();
                    const bool swap = n1 > n2;
                    if (swap) std::swap(n1, n2);
                    n2 = n1 > budget ? n1 : std::max(n1, budget - n1);
                    if (n1 + n2 > budget) { n1 = budget / 2; n2 = n1 + budget % 2; }
                    if (swap) std::swap(n1, n2);
                    a.resize(std::min(a.size(), n1));
                    b.resize(std::min(b.size(), n2));
                }
            }
            int32_t* row = ids + (size_t)i * max_len;
            int32_t* trow = type_ids ? type_ids + (size_t)i * max_len : nullptr;
            int p = 0;
            row[p] = tk->cls; if (trow) trow[p] = 0; ++p;
            for (int v : a) { row[p] = v; if (trow) trow[p] = 0; ++p; }
            row[p] = tk->sep; if (trow) trow[p] = 0; ++p;
            if (pair) {
                for (int v : b) { row[p] = v; if (trow) trow[p] = 1; ++p; }
                row[p] = tk->sep; if (trow) trow[p] = 1; ++p;
            }
            lens[i] = p;
            for (; p < max_len; ++p) { row[p] = tk->pad; if (trow) trow[p] = 0; }
        }
    };
    if (nthreads == 1) { work(0, n); return RMU_OK; }
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; ++t) th.emplace_back(work, (int)((int64_t)n * t / nthreads), (int)((int64_t)n * (t + 1) / nthreads));
    for (auto& x : th) x.join();
    return RMU_OK;
}

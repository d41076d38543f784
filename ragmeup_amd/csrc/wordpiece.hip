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
// Unicode: the per-codepoint behaviour of the normaliser and pre-tokeniser over the whole code space (deleted characters,
// spaces, CJK padding, lower-case + NFD + mark removal, punctuation) comes from wordpiece_tables.h, which
// tools/gen_wordpiece_tables.py records from the `tokenizers` library itself.  Not captured: the Greek final-sigma rule.
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>
#include <thread>
#include <pthread.h>
#include <mutex>
#include <memory>
#include <functional>
#include <condition_variable>
#include <unordered_map>
#include <vector>

#include "../../include/rmu.h"
#include "rmu_common.h"
#include "wordpiece_tables.h"

extern "C" void rmu_set_error_(const char* msg);

// Open-addressing table over the vocabulary for the greedy longest-match loop: keyed by (bytes, length) with FNV-1a, whose state
// after every byte is the hash of that PREFIX -- one pass over a word's remaining bytes yields the hashes of all its candidate
// pieces, so a probe never builds a std::string or re-hashes (the unordered_map<string> version spent its time in substr + "##"
// concatenation + hashing: 3.6k texts/s per core).  Whole-word pieces and "##" continuations live in separate tables.
struct PieceTable {
    struct Slot { uint64_t hash; uint32_t off, len; int32_t id; };
    std::vector<Slot> slots;
    std::string pool;
    uint64_t mask = 0;
    static constexpr uint64_t kOff = 1469598103934665603ull, kPrime = 1099511628211ull;
    void build(const std::vector<std::pair<std::string, int>>& items) {
        size_t cap = 64;
        while (cap < items.size() * 3) cap <<= 1;
        slots.assign(cap, Slot{0, 0, 0, -1});
        mask = cap - 1;
        for (const auto& it : items) {
            uint64_t h = kOff;
            for (unsigned char ch : it.first) h = (h ^ ch) * kPrime;
            size_t i = (size_t)(h & mask);
            bool dup = false;
            while (slots[i].id >= 0) {
                if (slots[i].hash == h && slots[i].len == it.first.size() && memcmp(pool.data() + slots[i].off, it.first.data(), it.first.size()) == 0) { dup = true; break; }
                i = (i + 1) & mask;
            }
            if (dup) continue;                      // a repeated vocabulary line keeps its first id (as the map did)
            slots[i] = Slot{h, (uint32_t)pool.size(), (uint32_t)it.first.size(), it.second};
            pool += it.first;
        }
    }
    int find(uint64_t h, const char* p, size_t n) const {
        if (slots.empty()) return -1;
        size_t i = (size_t)(h & mask);
        while (slots[i].id >= 0) {
            if (slots[i].hash == h && slots[i].len == n && memcmp(pool.data() + slots[i].off, p, n) == 0) return slots[i].id;
            i = (i + 1) & mask;
        }
        return -1;
    }
};

struct rmu_tok {
    std::unordered_map<std::string, int> vocab;
    PieceTable first, cont;                              // whole-word pieces / "##" continuations (stored without the prefix)
    int unk = 0, cls = 0, sep = 0, pad = 0;
    bool lower = true;
    std::vector<std::pair<std::string, int>> specials;   // literal special tokens present in the vocabulary
    unsigned char ascii_class[128];                      // 0 keep, 1 space, 2 removed, 3 punctuation (both normaliser variants agree on ASCII)
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
bool in_ranges(const uint32_t (*r)[2], int n, cp_t c) {
    int lo = 0, hi = n - 1;
    while (lo <= hi) {
        const int mid = (lo + hi) >> 1;
        if (c < r[mid][0]) hi = mid - 1;
        else if (c > r[mid][1]) lo = mid + 1;
        else return true;
    }
    return false;
}
// lower-case + NFD + mark removal of one codepoint (tables recorded from the `tokenizers` library): appends 0..n codepoints
void fold_into(cp_t c, std::vector<cp_t>& out) {
    if (c < 0x80) { out.push_back((c >= 'A' && c <= 'Z') ? c + 32 : c); return; }
    if (c >= 0xAC00 && c <= 0xD7A3) {                   // Hangul syllable: algorithmic canonical decomposition
        const cp_t s = c - 0xAC00;
        out.push_back(0x1100 + s / 588);
        out.push_back(0x1161 + (s % 588) / 28);
        if (s % 28) out.push_back(0x11A7 + s % 28);
        return;
    }
    int lo = 0, hi = kWpMap_n - 1;
    while (lo <= hi) {
        const int mid = (lo + hi) >> 1;
        if (c < kWpMap[mid].cp) hi = mid - 1;
        else if (c > kWpMap[mid].cp) lo = mid + 1;
        else {
            for (uint32_t i = 0; i < kWpMap[mid].len; ++i) out.push_back(kWpMapVals[kWpMap[mid].off + i]);
            return;
        }
    }
    out.push_back(c);
}

// greedy longest-match-first WordPiece of one pre-token (UTF-8 bytes [w, w + n)); > 100 characters -> [UNK]
void wordpiece(const rmu_tok* tk, const char* w, size_t n, std::vector<int>& ids, std::vector<uint64_t>& hs) {
    size_t nchars = 0;
    for (size_t i = 0; i < n; ++i) nchars += ((unsigned char)w[i] & 0xC0) != 0x80;
    if (nchars > 100) { ids.push_back(tk->unk); return; }
    const size_t mark = ids.size();
    size_t start = 0;
    hs.resize(n + 1);
    while (start < n) {
        // FNV-1a states after every byte of w[start ..): hs[e] = hash of w[start, e)
        uint64_t h = PieceTable::kOff;
        for (size_t e = start; e < n; ++e) { h = (h ^ (unsigned char)w[e]) * PieceTable::kPrime; hs[e + 1] = h; }
        const PieceTable& tab = start ? tk->cont : tk->first;
        int found = -1;
        size_t end = n;
        for (; end > start; --end) {
            if (end < n && ((unsigned char)w[end] & 0xC0) == 0x80) continue;      // only character boundaries
            found = tab.find(hs[end], w + start, end - start);
            if (found >= 0) break;
        }
        if (found < 0) { ids.resize(mark); ids.push_back(tk->unk); return; }
        ids.push_back(found);
        start = end;
    }
}

// BertNormalizer (clean text, pad CJK, strip accents, lower-case) + BertPreTokenizer (split on whitespace, isolate
// punctuation), streaming: every finished pre-token goes straight into WordPiece.  ASCII bytes take a 128-entry class table
// (the generated Unicode tables agree with it -- tests/test_tokenizer_cpu.py compares against the `tokenizers` library).
void encode_segment(const rmu_tok* tk, const char* text, size_t len, std::vector<int>& ids) {
    std::string cur;
    std::vector<uint64_t> hs;
    std::vector<cp_t> folded;
    auto flush = [&]() {
        if (cur.empty()) return;
        wordpiece(tk, cur.data(), cur.size(), ids, hs);
        cur.clear();
    };
    const unsigned char* p = (const unsigned char*)text;
    const unsigned char* e = p + len;
    while (p < e) {
        cp_t c;
        if (*p < 0x80) {
            c = *p++;
            if (c == 0) break;
            switch (tk->ascii_class[c]) {
                case 1: flush(); continue;
                case 2: continue;
                case 3: flush(); cur.push_back((char)c); flush(); continue;
                default: cur.push_back((char)((tk->lower && c >= 'A' && c <= 'Z') ? c + 32 : c)); continue;
            }
        }
        int n;
        if ((*p >> 5) == 6) { c = *p & 31; n = 2; }
        else if ((*p >> 4) == 14) { c = *p & 15; n = 3; }
        else if ((*p >> 3) == 30) { c = *p & 7; n = 4; }
        else { c = 0xFFFD; n = 0; }
        bool ok = n > 0 && p + n <= e;
        for (int i = 1; ok && i < n; ++i) {
            if ((p[i] & 0xC0) != 0x80) { ok = false; break; }
            c = (c << 6) | (p[i] & 63);
        }
        if (!ok) { c = 0xFFFD; p += 1; } else p += n;
        if (in_ranges(kWpSpace, kWpSpace_n, c)) { flush(); continue; }
        if (tk->lower ? in_ranges(kWpRemoved, kWpRemoved_n, c) : in_ranges(kWpRemovedCased, kWpRemovedCased_n, c)) continue;
        const bool cjk = in_ranges(kWpCjk, kWpCjk_n, c);   // padded with spaces: a token of its own
        if (cjk) flush();
        folded.clear();
        if (tk->lower) fold_into(c, folded);
        else folded.push_back(c);
        for (cp_t f : folded) {
            if (f == ' ') { flush(); continue; }        // compatibility ideographs decompose to a padded ideograph
            if (in_ranges(kWpPunct, kWpPunct_n, f)) { flush(); append_utf8(cur, f); flush(); continue; }
            append_utf8(cur, f);
        }
        if (cjk) flush();
    }
    flush();
}

// The special tokens are "added tokens" of the HF tokenizer: literal occurrences in the RAW text (case-sensitive, before
// normalisation) are cut out first and map to their single id; only the text between them is normalised and split
// (a chunk that talks about BERT's "[SEP]" keeps one id there, not '[', 'sep', ']').
void encode_text(const rmu_tok* tk, const char* text, std::vector<int>& ids) {
    if (!text) return;
    if (!strchr(text, '[')) { encode_segment(tk, text, strlen(text), ids); return; }   // every special token starts with '['
    const std::string t = text;
    size_t pos = 0;
    while (pos <= t.size()) {
        size_t best = std::string::npos, best_len = 0;
        int best_id = -1;
        for (const auto& sp : tk->specials) {
            const size_t f = t.find(sp.first, pos);
            if (f != std::string::npos && (f < best || (f == best && sp.first.size() > best_len))) {
                best = f; best_len = sp.first.size(); best_id = sp.second;
            }
        }
        if (best == std::string::npos) { encode_segment(tk, t.data() + pos, t.size() - pos, ids); break; }
        if (best > pos) encode_segment(tk, t.data() + pos, best - pos, ids);
        ids.push_back(best_id);
        pos = best + best_len;
    }
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
    {
        std::vector<std::pair<std::string, int>> fi, co;
        // ids in file order: a repeated line keeps its FIRST index in `vocab` (emplace) and the tables must agree
        for (const auto& kv : tk->vocab) {
            if (kv.first.size() > 2 && kv.first[0] == '#' && kv.first[1] == '#') co.emplace_back(kv.first.substr(2), kv.second);
            fi.emplace_back(kv.first, kv.second);       // a "##x" line is also a whole-word piece for the literal text "##x"
        }
        tk->first.build(fi);
        tk->cont.build(co);
        for (int c = 0; c < 128; ++c) {
            unsigned char cl = 0;
            if (c == ' ' || in_ranges(kWpSpace, kWpSpace_n, (cp_t)c)) cl = 1;
            else if (tk->lower ? in_ranges(kWpRemoved, kWpRemoved_n, (cp_t)c) : in_ranges(kWpRemovedCased, kWpRemovedCased_n, (cp_t)c)) cl = 2;
            else if (in_ranges(kWpPunct, kWpPunct_n, (cp_t)c)) cl = 3;
            tk->ascii_class[c] = cl;
        }
    }
    for (const char* sp : {"[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"}) {
        auto it = tk->vocab.find(sp);
        if (it != tk->vocab.end()) tk->specials.emplace_back(sp, it->second);
    }
    *out = tk;
    return RMU_OK;
}

// Persistent workers for rmu_tok_encode.  A job is a heap object the workers hold by shared_ptr, so one that wakes late only finds an
// exhausted counter; the caller works on its own job too, so a call completes even if every worker is busy elsewhere.  The pool is
// created on first use, never destroyed (no joins during process exit) and re-created in a forked child.
class TokPool {
    struct Job {
        std::function<void(int, int)> fn;
        int n = 0, grain = 1;
        std::atomic<int> next{0}, done{0};
        std::mutex mu;
        std::condition_variable cv;
    };
    std::vector<std::thread> workers_;
    std::mutex mu_;
    std::condition_variable cv_;
    std::shared_ptr<Job> current_;
    uint64_t gen_ = 0;

    static void drain(Job& j) {
        for (;;) {
            const int lo = j.next.fetch_add(j.grain);
            if (lo >= j.n) return;
            const int hi = std::min(j.n, lo + j.grain);
            j.fn(lo, hi);
            if (j.done.fetch_add(hi - lo) + (hi - lo) == j.n) {
                std::lock_guard<std::mutex> lk(j.mu);
                j.cv.notify_all();
            }
        }
    }
    void loop() {
        uint64_t seen = 0;
        std::unique_lock<std::mutex> lk(mu_);
        for (;;) {
            cv_.wait(lk, [&] { return gen_ != seen; });
            seen = gen_;
            std::shared_ptr<Job> j = current_;
            lk.unlock();
            if (j) drain(*j);
            j.reset();
            lk.lock();
        }
    }
    explicit TokPool(int n) {
        for (int i = 0; i < n; ++i) {
            try { workers_.emplace_back([this] { loop(); }); } catch (...) { break; }     // fewer workers is still correct
        }
    }
    static std::atomic<TokPool*>& slot() { static std::atomic<TokPool*> p{nullptr}; return p; }

  public:
    static TokPool& get() {
        static std::mutex create_mu;
        TokPool* p = slot().load(std::memory_order_acquire);
        if (p) return *p;
        std::lock_guard<std::mutex> lk(create_mu);
        p = slot().load(std::memory_order_acquire);
        if (!p) {
            static bool hooked = false;
            if (!hooked) { pthread_atfork(nullptr, nullptr, [] { slot().store(nullptr); }); hooked = true; }   // threads do not survive fork
            const int hw = (int)std::thread::hardware_concurrency();
            // 64 threads at most (round 5): the greedy longest-match loop is memory-latency bound -- 32 threads tokenise 1M texts no slower than
            // 256 do -- and a pool as wide as the machine starves the thread that feeds the GPU while the next block is tokenised (the
            // vector store's insert pipeline: 0.83-0.85 of the encoder-only rate with 255 helpers, 0.89 with 63; tools/idx_ab.sh)
            p = new TokPool(std::max(0, std::min(hw, 64) - 1));
            slot().store(p, std::memory_order_release);
        }
        return *p;
    }
    // fn(lo, hi) over [0, n) in grabs of `grain`; false = the job object could not be allocated
    bool run(int n, int grain, const std::function<void(int, int)>& fn) {
        std::shared_ptr<Job> j;
        try { j = std::make_shared<Job>(); j->fn = fn; } catch (...) { return false; }
        j->n = n; j->grain = grain;
        static const int cap = rmu_env("RMU_TOK_THREADS") ? std::max(1, atoi(rmu_env("RMU_TOK_THREADS"))) - 1 : 1 << 30;   // (tuning: helpers per job)
        const int helpers = std::min<int>(std::min<int>((int)workers_.size(), cap), (n + grain - 1) / grain - 1);
        if (helpers > 0) {
            { std::lock_guard<std::mutex> lk(mu_); current_ = j; ++gen_; }
            if (helpers * 2 >= (int)workers_.size()) cv_.notify_all();
            else for (int i = 0; i < helpers; ++i) cv_.notify_one();
        }
        drain(*j);
        std::unique_lock<std::mutex> lk(j->mu);
        j->cv.wait(lk, [&] { return j->done.load() >= j->n; });
        return true;
    }
};

extern "C" int rmu_tok_free(rmu_tok_t* tk) { delete tk; return RMU_OK; }
extern "C" int rmu_tok_vocab_size(rmu_tok_t* tk) { return tk ? (int)tk->vocab.size() : 0; }

extern "C" int rmu_tok_encode(rmu_tok_t* tk, const char* const* texts_a, const char* const* texts_b, int n, int max_len,
                              int32_t* ids, int32_t* type_ids, int32_t* lens) {
    if (!tk || !texts_a || !ids || !lens || n < 0 || max_len < 3) { rmu_set_error_("rmu_tok_encode: bad argument"); return RMU_E_INVALID; }
    std::atomic<int> failed{0};       // an exception (bad_alloc) inside a worker must not reach std::terminate
    auto work_body = [&](int lo, int hi) {
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
    auto work = [&](int lo, int hi) {
        try { work_body(lo, hi); } catch (...) { failed.store(1); }
    };
    // 8 texts (~0.25 ms) per grab: the reference's 1000-chunk add_documents call spreads over ~125 workers of the persistent pool
    // (spawning threads per call cost more than the tokenising: 2.0 ms per 1000 texts with a 64-text grain and fresh threads)
    if (n <= 8) work(0, n);
    else if (!TokPool::get().run(n, 8, work)) failed.store(1);
    if (failed.load()) { rmu_set_error_("rmu_tok_encode: out of memory while tokenising"); return RMU_E_OOM; }
    return RMU_OK;
}

extern "C" int rmu_tok_encode_blob(rmu_tok_t* tk, const char* blob_a, int64_t bytes_a, const char* blob_b, int64_t bytes_b, int n, int max_len,
                                   int32_t* ids, int32_t* type_ids, int32_t* lens) {
    if (!tk || !blob_a || n < 0 || bytes_a < n || (blob_b && bytes_b < n)) { rmu_set_error_("rmu_tok_encode_blob: bad argument"); return RMU_E_INVALID; }
    std::vector<const char*> pa, pb;
    try {
        auto split = [&](const char* blob, int64_t bytes, std::vector<const char*>& out) {
            out.reserve((size_t)n);
            const char* p = blob;
            const char* e = blob + bytes;
            while (p < e && (int)out.size() < n) {
                const char* z = (const char*)memchr(p, 0, (size_t)(e - p));
                if (!z) return false;
                out.push_back(p);
                p = z + 1;
            }
            return (int)out.size() == n && p == e;
        };
        if (!split(blob_a, bytes_a, pa) || (blob_b && !split(blob_b, bytes_b, pb))) {
            rmu_set_error_("rmu_tok_encode_blob: the blob does not hold exactly n NUL-terminated strings");
            return RMU_E_INVALID;
        }
    } catch (...) { rmu_set_error_("rmu_tok_encode_blob: out of memory"); return RMU_E_OOM; }
    return rmu_tok_encode(tk, pa.data(), blob_b ? pb.data() : nullptr, n, max_len, ids, type_ids, lens);
}

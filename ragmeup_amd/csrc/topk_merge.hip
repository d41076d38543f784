// topk_merge.hip -- k-way merge of per-chunk / per-shard top-k lists (gfx950 only).
//
// Serves: the last step of the dense search (server/RAGHelper.py:497-499 -> top-`fetch_k` per query)
// and the 8-GPU shard merge after the RCCL all-gather.  One wave per query streams
// the `parts*k` candidate keys in batches of 64, sorts each batch with a shuffle bitonic network and
// folds it into the running (sorted) top-K with one half-cleaner + bitonic merge.
#include <cstdlib>
#include "rmu_common.h"
#include "../../include/rmu.h"

namespace {

// NPL = 1 -> k <= 64, NPL = 2 -> k <= 128
template <int NPL, class LoadKey>
__device__ __forceinline__ void merge_stream(u64 (&top)[NPL], int64_t m, int lane, LoadKey load) {
#pragma unroll
    for (int p = 0; p < NPL; ++p) top[p] = 0ull;
    for (int64_t b0 = 0; b0 < m; b0 += 64) {
        u64 bk[1];
        const int64_t idx = b0 + lane;
        bk[0] = idx < m ? load(idx) : 0ull;
        rmu_bitonic_sort_desc<1>(bk, lane);
        // reversed batch against the LAST 64 slots of the running list -> bitonic sequence holding the
        // best 64*NPL of the union
        const u64 rev = __shfl(bk[0], 63 - lane);
        u64& tail = top[NPL - 1];
        tail = tail > rev ? tail : rev;
        rmu_bitonic_merge_desc<NPL>(top, lane);
    }
}

// Same fold, visiting the part lists rank slab by rank slab (ranks [4s, 4s+4) of every part): the lists are sorted with
// their zeros last, so the first slab that holds no key at all ends the merge.  The screening ladder's seeded launches
// leave one to three candidates per (chunk, query); this reads ~1/4 of the keys a part-major sweep would.
template <int NPL, class LoadKey>
__device__ __forceinline__ void merge_stream_slabs(u64 (&top)[NPL], int np, int k, int lane, LoadKey load /* (part, pos) */) {
#pragma unroll
    for (int p = 0; p < NPL; ++p) top[p] = 0ull;
    for (int s0 = 0; s0 < k; s0 += 4) {
        bool any = false;
        const int m = np * 4;
        for (int b0 = 0; b0 < m; b0 += 64) {
            u64 bk[1];
            const int i = b0 + lane;
            const int pos = s0 + (i & 3);
            bk[0] = (i < m && pos < k) ? load(i >> 2, pos) : 0ull;
            if (!__any(bk[0] != 0ull)) continue;
            any = true;
            rmu_bitonic_sort_desc<1>(bk, lane);
            const u64 rev = __shfl(bk[0], 63 - lane);
            u64& tail = top[NPL - 1];
            tail = tail > rev ? tail : rev;
            rmu_bitonic_merge_desc<NPL>(top, lane);
        }
        if (!any) break;
    }
}

// Software-pipelined form of merge_stream_slabs: the load of the next 64-entry batch is in flight while the current one
// is sorted and folded, so a wave pays one memory latency per merge instead of one per batch.
template <int NPL, class LoadKey>
__device__ __forceinline__ void merge_stream_slabs_pf(u64 (&top)[NPL], int np, int k, int lane, LoadKey load /* (part, pos) */) {
#pragma unroll
    for (int p = 0; p < NPL; ++p) top[p] = 0ull;
    const int m = np * 4;
    if (m <= 0) return;
    auto ld = [&](int s0, int b0) -> u64 {
        const int i = b0 + lane;
        const int pos = s0 + (i & 3);
        return (s0 < k && i < m && pos < k) ? load(i >> 2, pos) : 0ull;
    };
    int s0 = 0, b0 = 0;
    u64 cur = ld(0, 0);
    bool any = false;
    while (s0 < k) {
        int ns0 = s0, nb0 = b0 + 64;
        if (nb0 >= m) { nb0 = 0; ns0 = s0 + 4; }
        const u64 nxt = ld(ns0, nb0);
        if (__any(cur != 0ull)) {
            any = true;
            u64 bk[1] = {cur};
            rmu_bitonic_sort_desc<1>(bk, lane);
            const u64 rev = __shfl(bk[0], 63 - lane);
            u64& tail = top[NPL - 1];
            tail = tail > rev ? tail : rev;
            rmu_bitonic_merge_desc<NPL>(top, lane);
        }
        if (nb0 == 0) {            // the slab is done: lists are sorted with their zeros last, an empty slab ends the merge
            if (!any) break;
            any = false;
        }
        s0 = ns0; b0 = nb0; cur = nxt;
    }
}

// Merge of the [parts, nq, k] key lists of one scan launch, `wpq` waves per query inside one 1024-thread workgroup:
// wave `sub` of a query folds parts sub, sub + wpq, ... (rank-slab order, prefetched), parks its k best in LDS, and the
// query's first wave folds those wpq lists.  One launch replaces the former two-level pair; few-query batches (the
// HBM-bound regime, where a merge used to cost 40-60 us of dependent load -> sort rounds on a single wave) get 16 waves per
// query.  Output: keys (+ threshold seeding for the screening ladder) or final (score, row) lists.
// `cond`: device-side predicate of the conditional exact re-runs behind the screening path (see ScanLaunch::cond).
struct MergeOut {
    u64* keys;            // [nq, k] merged keys, or null
    u32* seed_thr;        // with keys: atomicMax of the merged k-th best (next ladder launch's shared threshold), or null
    float* scores;        // [nq, k] final scores, or null
    int64_t* rows;        // [nq, k] final rows (+ row_base)
    int64_t row_base;
    int l2_out;           // scores = max(qnorm2[qo] - s, 0), smaller = better (RMU_METRIC_L2SQ); qnorm2 is indexed like the OUTPUT rows
                          // (the batch's queries: a conditional re-run of the flagged queries scatters into them)
    const float* qnorm2;
    const int64_t* scatter;   // optional: query i writes output row scatter[i] (patching re-run queries into the batch)
    int unsorted;             // the part lists are compact (zeros last) but NOT sorted: the screening ladder's first launch (scan_screen.hip, share_thr bit 2)
};
template <int NPL>
__global__ __launch_bounds__(1024) void merge_wg_kernel(const u64* __restrict__ partial, int parts, int64_t nq, int k, int wpq,
                                                        MergeOut o, RmuCond cond) {
    __shared__ u64 lists[16][64 * NPL];
    int64_t nq_eff = nq;
    if (cond.p) {
        const int c = *cond.p;
        if (c < cond.lo || c > cond.hi) return;           // uniform over the grid
        if (cond.clamp && c < nq_eff) nq_eff = c;
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int qpb = 16 / wpq;
    const int64_t q = (int64_t)blockIdx.x * qpb + w / wpq;
    const int sub = w % wpq;
    const bool active = q < nq_eff;
    u64 top[NPL];
#pragma unroll
    for (int p = 0; p < NPL; ++p) top[p] = 0ull;
    if (active) {
        const int npw = parts > sub ? (parts - sub + wpq - 1) / wpq : 0;
        merge_stream_slabs_pf<NPL>(top, npw, k, lane, [&](int part, int pos) {
            return partial[((int64_t)(sub + part * wpq) * nq + q) * k + pos];
        });
        if (wpq > 1) {
#pragma unroll
            for (int p = 0; p < NPL; ++p) lists[w][lane + 64 * p] = top[p];
        }
    }
    if (wpq > 1) __syncthreads();
    if (!active || sub != 0) return;
    if (wpq > 1)
        merge_stream_slabs_pf<NPL>(top, wpq, k, lane, [&](int part, int pos) { return lists[w + part][pos]; });
    const int64_t qo = o.scatter ? o.scatter[q] : q;
#pragma unroll
    for (int p = 0; p < NPL; ++p) {
        const int e = lane + 64 * p;
        if (e >= k) continue;
        const u64 key = top[p];
        if (o.keys) {
            o.keys[qo * k + e] = key;
            if (o.seed_thr && e == k - 1 && key) atomicMax(o.seed_thr + qo, (u32)(key >> 32));
        }
        if (o.scores) {
            float sc;
            int64_t r;
            if (key == 0ull) {
                sc = o.l2_out ? INFINITY : -INFINITY;
                r = -1;
            } else {
                sc = rmu_key_score(key);
                if (o.l2_out) sc = fmaxf(o.qnorm2[qo] - sc, 0.f);
                r = (int64_t)rmu_key_row(key) + o.row_base;
            }
            o.scores[qo * k + e] = sc;
            o.rows[qo * k + e] = r;
        }
    }
}

// k <= 32: merge by SELECTION instead of sorting -- no shuffle network at all.  One workgroup per query:
//   A. heads[p] = best key of part p; tau = the k-th largest head (enumeration rank over LDS broadcasts).  k distinct parts
//      hold a key >= tau, so the final k-th best is >= tau and only keys >= tau can be in the result;
//   B. every wave walks its parts rank slab by rank slab (the lists are sorted, zeros last: a slab without a key >= tau
//      ends that batch) and appends the keys >= tau to an LDS array -- at most k parts x k keys = k^2 <= 1024 of them;
//   C. enumeration rank among those; rank < k -> output position rank.
// The wave-serial sort/merge rounds of merge_wg_kernel (27 dependent 64-bit shuffle steps per 64 keys) made a ladder merge
// cost 40 us -- six of them were 0.24 ms of a 1.7 ms HBM-bound batch; this form is a few LDS sweeps.
// k <= 128 (round 3: the deep-k ladder's merges and every final merge of a k > 32 search): the same selection with a 3072-key
// array -- the k parts whose heads reach tau rarely hold more than a few keys >= tau each.  If they do (count > CAPM: up to
// k x k = 16k keys in the worst case) the query's first wave folds the part lists with the bitonic stream merge instead.
template <int BLOCK, int CAPM>
__global__ __launch_bounds__(BLOCK) void merge_select_kernel(const u64* __restrict__ partial, int parts, int64_t nq, int k,
                                                             MergeOut o, RmuCond cond) {
    constexpr int MAXP = 1024;
    __shared__ u64 heads[MAXP];
    __shared__ u64 cand[CAPM];
    __shared__ u64 tau_s;
    __shared__ u32 count;
    int64_t nq_eff = nq;
    if (cond.p) {
        const int c = *cond.p;
        if (c < cond.lo || c > cond.hi) return;           // uniform over the grid
        if (cond.clamp && c < nq_eff) nq_eff = c;
    }
    const int64_t q = blockIdx.x;
    if (q >= nq_eff) return;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    constexpr int NW = BLOCK / 64;
    const u64* base = partial + q * k;
    const int64_t pstride = nq * k;
    if (o.unsorted) {            // heads = the maximum of each list (LDS atomics), not its first entry
        for (int p = tid; p < parts; p += BLOCK) heads[p] = 0ull;
        __syncthreads();
        const int total = parts * k;
        for (int i = tid; i < total; i += BLOCK) {
            const int part = i / k, pos = i - part * k;
            const u64 key = base[(int64_t)part * pstride + pos];
            if (key) atomicMax((unsigned long long*)&heads[part], (unsigned long long)key);
        }
    } else {
        for (int p = tid; p < parts; p += BLOCK) heads[p] = base[(int64_t)p * pstride];
    }
    if (tid == 0) { count = 0u; tau_s = 0ull; }
    __syncthreads();
    if (parts >= k) {
        for (int p = tid; p < parts; p += BLOCK) {
            const u64 mine = heads[p];
            int rank = 0;
            for (int j = 0; j < parts; ++j) rank += heads[j] > mine ? 1 : 0;
            if (mine != 0ull && rank == k - 1) tau_s = mine;      // keys are distinct: exactly one writer (or none)
        }
        __syncthreads();
    }
    const u64 tau = tau_s;
    for (int pb = w * 16; pb < parts; pb += NW * 16) {
        const int part = pb + (lane >> 2);
        for (int s0 = 0; s0 < k; s0 += 4) {
            const int pos = s0 + (lane & 3);
            u64 key = 0ull;
            if (part < parts && pos < k) key = base[(int64_t)part * pstride + pos];
            const bool take = key != 0ull && key >= tau;
            const u64 bal = __ballot(take);
            if (!bal) {
                if (o.unsorted && __ballot(key != 0ull)) continue;     // an unsorted list may hold its keys >= tau further back: only an EMPTY slab ends it
                break;
            }
            u32 at = 0;
            if (lane == 0) at = atomicAdd(&count, (u32)__builtin_popcountll(bal));
            at = __shfl(at, 0) + (u32)__builtin_popcountll(bal & ((1ull << lane) - 1ull));
            if (take && at < (u32)CAPM) cand[at] = key;
        }
    }
    __syncthreads();
    const u32 m = count < (u32)CAPM ? count : (u32)CAPM;
    const int64_t qo = o.scatter ? o.scatter[q] : q;
    auto emit = [&](int e, u64 key) {
        if (o.keys) {
            o.keys[qo * k + e] = key;
            if (o.seed_thr && e == k - 1 && key) atomicMax(o.seed_thr + qo, (u32)(key >> 32));
        }
        if (o.scores) {
            float sc;
            int64_t r;
            if (key == 0ull) {
                sc = o.l2_out ? INFINITY : -INFINITY;
                r = -1;
            } else {
                sc = rmu_key_score(key);
                if (o.l2_out) sc = fmaxf(o.qnorm2[qo] - sc, 0.f);
                r = (int64_t)rmu_key_row(key) + o.row_base;
            }
            o.scores[qo * k + e] = sc;
            o.rows[qo * k + e] = r;
        }
    };
    if (CAPM < 128 * 128 && count > (u32)CAPM) {    // (k <= 32: k x k <= 1024 keys, cannot happen)
        if (w == 0) {
            u64 top[2];
            merge_stream_slabs_pf<2>(top, parts, k, lane, [&](int part, int pos) { return base[(int64_t)part * pstride + pos]; });
#pragma unroll
            for (int p = 0; p < 2; ++p)
                if (lane + 64 * p < k) emit(lane + 64 * p, top[p]);
        }
        return;
    }
    for (u32 i = tid; i < m; i += BLOCK) {
        const u64 mine = cand[i];
        u32 rank = 0;
        for (u32 j = 0; j < m; ++j) rank += cand[j] > mine ? 1u : 0u;
        if (rank < (u32)k) emit((int)rank, mine);
    }
    for (int e = tid; e < k; e += BLOCK)
        if ((u32)e >= m) emit(e, 0ull);
}

// generic lists (scores fp32 + int64 rows; part p at scores + p*stride_s / rows + p*stride_r, each [nq, k]); ties resolve
// to the lower candidate index, i.e. the lower part, then the earlier position -- equal to (score, row) order when
// parts arrive in ascending row ranges.  smaller_better: the scores are distances (keys are built from -score).
template <int NPL>
__global__ __launch_bounds__(256) void merge_lists_kernel(const float* __restrict__ scores,
                                                          const int64_t* __restrict__ rows, int parts, int64_t stride_s,
                                                          int64_t stride_r, int64_t nq, int k, int smaller_better,
                                                          float* __restrict__ out_scores,
                                                          int64_t* __restrict__ out_rows) {
    const int lane = threadIdx.x & 63;
    const int64_t q = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= nq) return;
    const int64_t m = (int64_t)parts * k;
    u64 top[NPL];
    merge_stream<NPL>(top, m, lane, [&](int64_t idx) -> u64 {
        const int64_t part = idx / k, pos = idx % k;
        const float s = scores[part * stride_s + q * k + pos];
        if (rows[part * stride_r + q * k + pos] < 0 || !(s == s)) return 0ull;
        return rmu_make_key((smaller_better ? -s : s) + 0.0f, (u32)idx);
    });
#pragma unroll
    for (int p = 0; p < NPL; ++p) {
        const int e = lane + 64 * p;
        if (e < k) {
            const u64 key = top[p];
            if (key == 0ull) {
                out_scores[q * k + e] = smaller_better ? INFINITY : -INFINITY;
                out_rows[q * k + e] = -1;
            } else {
                const int64_t idx = rmu_key_row(key);
                const int64_t part = idx / k, pos = idx % k;
                out_scores[q * k + e] = smaller_better ? -rmu_key_score(key) : rmu_key_score(key);
                out_rows[q * k + e] = rows[part * stride_r + q * k + pos];
            }
        }
    }
}

}  // namespace

static int merge_wg_launch(const u64* partial, int parts, int64_t nq, int k, const MergeOut& o, const RmuCond& cond, hipStream_t s) {
    if (k < 1 || k > 128 || parts < 1 || nq < 1) return RMU_E_INVALID;
    static const int use_select = rmu_env("RMU_MERGE_SELECT") ? atoi(rmu_env("RMU_MERGE_SELECT")) : 1;
    if (use_select && k <= 32 && parts <= 1024) {          // selection merge: one workgroup per query
        if (nq <= 512) hipLaunchKernelGGL((merge_select_kernel<1024, 1024>), dim3((unsigned)nq), dim3(1024), 0, s, partial, parts, nq, k, o, cond);
        else hipLaunchKernelGGL((merge_select_kernel<256, 1024>), dim3((unsigned)nq), dim3(256), 0, s, partial, parts, nq, k, o, cond);
        return hipGetLastError() == hipSuccess ? RMU_OK : RMU_E_HIP;
    }
    if (use_select && parts <= 1024) {                     // 32 < k <= 128
        if (nq <= 512) hipLaunchKernelGGL((merge_select_kernel<1024, 3072>), dim3((unsigned)nq), dim3(1024), 0, s, partial, parts, nq, k, o, cond);
        else hipLaunchKernelGGL((merge_select_kernel<256, 3072>), dim3((unsigned)nq), dim3(256), 0, s, partial, parts, nq, k, o, cond);
        return hipGetLastError() == hipSuccess ? RMU_OK : RMU_E_HIP;
    }
    // waves per query: ~16 parts per wave, and at least ~2k waves in flight when the batch is small
    int wpq = 1;
    while (wpq < 16 && parts > 16 * wpq) wpq <<= 1;
    while (wpq < 16 && nq * wpq < 2048 && wpq < parts) wpq <<= 1;
    const int qpb = 16 / wpq;
    const dim3 grid((unsigned)((nq + qpb - 1) / qpb)), block(1024);
    if (k <= 64) hipLaunchKernelGGL(merge_wg_kernel<1>, grid, block, 0, s, partial, parts, nq, k, wpq, o, cond);
    else hipLaunchKernelGGL(merge_wg_kernel<2>, grid, block, 0, s, partial, parts, nq, k, wpq, o, cond);
    return hipGetLastError() == hipSuccess ? RMU_OK : RMU_E_HIP;
}

// final merge of a scan's part lists -> (scores, rows); `scatter` (optional) redirects query i to output row scatter[i]
int rmu_merge_final_launch(const u64* partial, int parts, int64_t nq, int k, int64_t row_base, int l2_out, const float* qnorm2,
                           float* out_scores, int64_t* out_rows, const int64_t* scatter, const RmuCond* cond, hipStream_t s) {
    MergeOut o{};
    o.scores = out_scores; o.rows = out_rows; o.row_base = row_base; o.l2_out = l2_out; o.qnorm2 = qnorm2; o.scatter = scatter;
    return merge_wg_launch(partial, parts, nq, k, o, cond ? *cond : RmuCond{}, s);
}

// merge to keys (screening ladder): out_keys [nq, k]; seed_thr (optional) receives the merged k-th best per query
int rmu_merge_to_keys_launch(const u64* partial, int parts, int64_t nq, int k, u64* out_keys, u32* seed_thr, hipStream_t s, int unsorted) {
    MergeOut o{};
    o.keys = out_keys; o.seed_thr = seed_thr; o.unsorted = unsorted;
    return merge_wg_launch(partial, parts, nq, k, o, RmuCond{}, s);
}

int rmu_merge_lists_launch(const float* scores, const int64_t* rows, int parts, int64_t stride_s, int64_t stride_r, int64_t nq, int k,
                           int smaller_better, float* out_scores, int64_t* out_rows, hipStream_t s) {
    if (k < 1 || k > 128 || parts < 1 || nq < 1) return RMU_E_INVALID;
    if ((int64_t)parts * k >= (1ll << 32)) return RMU_E_INVALID;
    const dim3 grid((unsigned)((nq + 3) / 4)), block(256);
    if (k <= 64)
        hipLaunchKernelGGL(merge_lists_kernel<1>, grid, block, 0, s, scores, rows, parts, stride_s, stride_r, nq, k, smaller_better,
                           out_scores, out_rows);
    else
        hipLaunchKernelGGL(merge_lists_kernel<2>, grid, block, 0, s, scores, rows, parts, stride_s, stride_r, nq, k, smaller_better,
                           out_scores, out_rows);
    return hipGetLastError() == hipSuccess ? RMU_OK : RMU_E_HIP;
}

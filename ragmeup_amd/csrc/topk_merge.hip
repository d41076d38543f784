// topk_merge.hip -- k-way merge of per-chunk / per-shard top-k lists (gfx950 only).
//
// Serves: the last step of the dense search (server/RAGHelper.py:497-499 -> top-`fetch_k` per query)
// and the 8-GPU shard merge after the RCCL all-gather.  One wave per query streams
// the `parts*k` candidate keys in batches of 64, sorts each batch with a shuffle bitonic network and
// folds it into the running (sorted) top-K with one half-cleaner + bitonic merge.
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

// level-1 of the two-level merge: wave (q, g) folds parts [g*ppg, (g+1)*ppg) into k keys -> scratch[g][q][k]
template <int NPL>
__global__ __launch_bounds__(256) void merge_keys_partial_kernel(const u64* __restrict__ partial, int parts, int64_t nq,
                                                                 int k, int groups, int ppg, u64* __restrict__ scratch,
                                                                 u32* __restrict__ seed_thr /* optional: [nq] <- max(k-th best) */) {
    const int lane = threadIdx.x & 63;
    const int64_t wid = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wid >= nq * groups) return;
    const int64_t q = wid / groups;
    const int g = (int)(wid % groups);
    const int p0 = g * ppg;
    const int np = min(ppg, parts - p0);
    u64 top[NPL];
    merge_stream_slabs<NPL>(top, np > 0 ? np : 0, k, lane, [&](int part, int pos) {
        return partial[((int64_t)(p0 + part) * nq + q) * k + pos];
    });
#pragma unroll
    for (int p = 0; p < NPL; ++p) {
        const int e = lane + 64 * p;
        if (e < k) scratch[((int64_t)g * nq + q) * k + e] = top[p];
        // screening ladder: the merged K'-th best is a lower bound of the final K'-th best -> next launch's shared threshold
        if (seed_thr && e == k - 1 && top[p]) atomicMax(seed_thr + q, (u32)(top[p] >> 32));
    }
}

template <int NPL>
__global__ __launch_bounds__(256) void merge_keys_kernel(const u64* __restrict__ partial, int parts, int64_t nq,
                                                         int k, int64_t row_base, int l2_out,
                                                         const float* __restrict__ qnorm2,
                                                         float* __restrict__ out_scores,
                                                         int64_t* __restrict__ out_rows) {
    const int lane = threadIdx.x & 63;
    const int64_t q = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= nq) return;
    const int64_t m = (int64_t)parts * k;
    u64 top[NPL];
    merge_stream<NPL>(top, m, lane, [&](int64_t idx) {
        const int64_t part = idx / k, pos = idx % k;
        return partial[(part * nq + q) * k + pos];
    });
#pragma unroll
    for (int p = 0; p < NPL; ++p) {
        const int e = lane + 64 * p;
        if (e < k) {
            const u64 key = top[p];
            float s;
            int64_t r;
            if (key == 0ull) {
                s = l2_out ? INFINITY : -INFINITY;
                r = -1;
            } else {
                s = rmu_key_score(key);
                if (l2_out) s = fmaxf(qnorm2[q] - s, 0.f);
                r = (int64_t)rmu_key_row(key) + row_base;
            }
            out_scores[q * k + e] = s;
            out_rows[q * k + e] = r;
        }
    }
}

// generic lists (scores fp32 + int64 rows, [parts, nq, k]); ties resolve to the lower candidate index,
// i.e. the lower part, then the earlier position -- equal to (score, row) order when parts arrive in
// ascending row ranges.
template <int NPL>
__global__ __launch_bounds__(256) void merge_lists_kernel(const float* __restrict__ scores,
                                                          const int64_t* __restrict__ rows, int parts,
                                                          int64_t nq, int k, float* __restrict__ out_scores,
                                                          int64_t* __restrict__ out_rows) {
    const int lane = threadIdx.x & 63;
    const int64_t q = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= nq) return;
    const int64_t m = (int64_t)parts * k;
    u64 top[NPL];
    merge_stream<NPL>(top, m, lane, [&](int64_t idx) -> u64 {
        const int64_t part = idx / k, pos = idx % k;
        const int64_t src = (part * nq + q) * k + pos;
        const float s = scores[src];
        if (rows[src] < 0 || !(s == s)) return 0ull;
        return rmu_make_key(s + 0.0f, (u32)idx);
    });
#pragma unroll
    for (int p = 0; p < NPL; ++p) {
        const int e = lane + 64 * p;
        if (e < k) {
            const u64 key = top[p];
            if (key == 0ull) {
                out_scores[q * k + e] = -INFINITY;
                out_rows[q * k + e] = -1;
            } else {
                const int64_t idx = rmu_key_row(key);
                const int64_t part = idx / k, pos = idx % k;
                out_scores[q * k + e] = rmu_key_score(key);
                out_rows[q * k + e] = rows[(part * nq + q) * k + pos];
            }
        }
    }
}

}  // namespace

int rmu_merge_keys_launch(const u64* partial, int parts, int64_t nq, int k, int64_t row_base, int l2_out,
                          const float* qnorm2, float* out_scores, int64_t* out_rows, hipStream_t s) {
    return rmu_merge_keys_launch2(partial, parts, nq, k, row_base, l2_out, qnorm2, out_scores, out_rows, nullptr, 0, s);
}

// scratch (optional, >= groups*nq*k keys): enables the two-level merge when few queries face many parts
// (nq = 1 over 1024 chunk lists is 10k keys for ONE wave: 326 us single-level, ~25 us two-level)
int rmu_merge_keys_launch2(const u64* partial, int parts, int64_t nq, int k, int64_t row_base, int l2_out,
                           const float* qnorm2, float* out_scores, int64_t* out_rows, u64* scratch, int64_t scratch_keys,
                           hipStream_t s) {
    if (k < 1 || k > 128 || parts < 1 || nq < 1) return RMU_E_INVALID;
    const dim3 block(256);
    const u64* src = partial;
    int src_parts = parts;
    if (scratch && parts >= 64 && nq * 4 <= 4096) {
        int groups = 32;
        while (groups > 1 && (int64_t)groups * nq > 8192) groups >>= 1;
        const int ppg = (parts + groups - 1) / groups;
        groups = (parts + ppg - 1) / ppg;
        if (groups > 1 && (int64_t)groups * nq * k <= scratch_keys) {
            const dim3 g1((unsigned)((nq * groups + 3) / 4));
            if (k <= 64) hipLaunchKernelGGL(merge_keys_partial_kernel<1>, g1, block, 0, s, partial, parts, nq, k, groups, ppg, scratch, (u32*)nullptr);
            else hipLaunchKernelGGL(merge_keys_partial_kernel<2>, g1, block, 0, s, partial, parts, nq, k, groups, ppg, scratch, (u32*)nullptr);
            src = scratch;
            src_parts = groups;
        }
    }
    const dim3 grid((unsigned)((nq + 3) / 4));
    if (k <= 64)
        hipLaunchKernelGGL(merge_keys_kernel<1>, grid, block, 0, s, src, src_parts, nq, k, row_base, l2_out, qnorm2,
                           out_scores, out_rows);
    else
        hipLaunchKernelGGL(merge_keys_kernel<2>, grid, block, 0, s, src, src_parts, nq, k, row_base, l2_out, qnorm2,
                           out_scores, out_rows);
    return hipGetLastError() == hipSuccess ? RMU_OK : RMU_E_HIP;
}

int rmu_merge_to_keys_launch(const u64* partial, int parts, int64_t nq, int k, u64* out_keys, u32* seed_thr, u64* scratch,
                             int64_t scratch_keys, hipStream_t s) {
    if (k < 1 || k > 128 || parts < 1 || nq < 1) return RMU_E_INVALID;
    const dim3 block(256);
    const u64* src = partial;
    int src_parts = parts;
    // few queries x many parts (one query tile scans 256 row chunks): one wave per query would fold thousands of keys
    // serially (58 us per merge at nq = 1); fold groups of 16 parts in parallel first
    if (scratch && nq <= 256 && parts >= 32) {
        const int ppg = 16, groups = (parts + ppg - 1) / ppg;
        if ((int64_t)groups * nq * k <= scratch_keys) {
            const dim3 g1((unsigned)((nq * groups + 3) / 4));
            if (k <= 64) hipLaunchKernelGGL(merge_keys_partial_kernel<1>, g1, block, 0, s, partial, parts, nq, k, groups, ppg, scratch, (u32*)nullptr);
            else hipLaunchKernelGGL(merge_keys_partial_kernel<2>, g1, block, 0, s, partial, parts, nq, k, groups, ppg, scratch, (u32*)nullptr);
            src = scratch;
            src_parts = groups;
        }
    }
    const dim3 grid((unsigned)((nq + 3) / 4));
    if (k <= 64) hipLaunchKernelGGL(merge_keys_partial_kernel<1>, grid, block, 0, s, src, src_parts, nq, k, 1, src_parts, out_keys, seed_thr);
    else hipLaunchKernelGGL(merge_keys_partial_kernel<2>, grid, block, 0, s, src, src_parts, nq, k, 1, src_parts, out_keys, seed_thr);
    return hipGetLastError() == hipSuccess ? RMU_OK : RMU_E_HIP;
}

int rmu_merge_lists_launch(const float* scores, const int64_t* rows, int parts, int64_t nq, int k,
                           float* out_scores, int64_t* out_rows, u64* /*scratch_keys*/, hipStream_t s) {
    if (k < 1 || k > 128 || parts < 1 || nq < 1) return RMU_E_INVALID;
    if ((int64_t)parts * k >= (1ll << 32)) return RMU_E_INVALID;
    const dim3 grid((unsigned)((nq + 3) / 4)), block(256);
    if (k <= 64)
        hipLaunchKernelGGL(merge_lists_kernel<1>, grid, block, 0, s, scores, rows, parts, nq, k, out_scores, out_rows);
    else
        hipLaunchKernelGGL(merge_lists_kernel<2>, grid, block, 0, s, scores, rows, parts, nq, k, out_scores, out_rows);
    return hipGetLastError() == hipSuccess ? RMU_OK : RMU_E_HIP;
}

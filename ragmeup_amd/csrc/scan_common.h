// scan_common.h -- device helpers shared by the exact fp32 scan (scan_topk.hip) and the fp16 screening
// scan (scan_screen.hip): LDS stores that do not drain the LDS-DMA ring, enumeration sort, slot compaction.
#pragma once
#include "rmu_common.h"

namespace {

// LDS byte address of a pointer into dynamic shared memory
__device__ __forceinline__ u32 lds_addr(const void* p) {
    return (u32)(uintptr_t)(__attribute__((address_space(3))) char*)(char*)p;
}
// LDS stores as inline asm: a compiler-generated LDS write is ordered behind the in-flight LDS-DMA with
// s_waitcnt vmcnt(0) and would drain the ring.
__device__ __forceinline__ void lds_store_b64(u32 addr, u64 v) {
    asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
// same, without the "memory" clobber: the compiler may move its own ring reads across it (the targets never
// alias the ring); a clobbering wait follows before anything reads the stored data back
__device__ __forceinline__ void lds_store_b64_nofence(u32 addr, u64 v) {
    asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v));
}
__device__ __forceinline__ void lds_store_b32(u32 addr, u32 v) {
    asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory");
}

// rank[p] = number of the wave's n keys that are larger than key[p]  (keys are distinct or 0).
// Enumeration sort: broadcast key i with v_readlane (no LDS traffic), every lane counts.  ~6 instructions
// per candidate; a compaction runs while the other three waves of the workgroup wait at the ring barrier,
// so its latency is paid four times -- the shuffle bitonic sort used here before cost 10% of the kernel.
template <int NPL>
__device__ __forceinline__ void rank_keys(const u64 (&key)[NPL], u32 n, u32 (&rank)[NPL]) {
#pragma unroll
    for (int p = 0; p < NPL; ++p) rank[p] = 0;
#pragma unroll
    for (int sp = 0; sp < NPL; ++sp) {
        const u32 lim = n > 64u * sp ? (n - 64u * sp < 64u ? n - 64u * sp : 64u) : 0u;
        const u32 lo = (u32)key[sp], hi = (u32)(key[sp] >> 32);
        // (round 6) four candidates per trip: one candidate is a chain readlane -> SGPR pair -> 64-bit compare -> VCC -> add, ~90 cycles when the
        // wave is alone on its SIMD (the emit of a cold range: 32 queries x 32 candidates = 1.6 us per query); four independent chains overlap
        u32 i = 0;
        for (; i + 4 <= lim; i += 4) {
            u64 ki[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                ki[u] = ((u64)(u32)__builtin_amdgcn_readlane((int)hi, (int)(i + u)) << 32) | (u64)(u32)__builtin_amdgcn_readlane((int)lo, (int)(i + u));
#pragma unroll
            for (int p = 0; p < NPL; ++p) {
                u32 c[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) c[u] = (ki[u] > key[p]) ? 1u : 0u;
                rank[p] += (c[0] + c[1]) + (c[2] + c[3]);
            }
        }
        for (; i < lim; ++i) {
            const u64 ki = ((u64)(u32)__builtin_amdgcn_readlane((int)hi, (int)i) << 32) |
                           (u64)(u32)__builtin_amdgcn_readlane((int)lo, (int)i);
#pragma unroll
            for (int p = 0; p < NPL; ++p) rank[p] += (ki > key[p]) ? 1u : 0u;
        }
    }
}

// keep the best k of slot j (sorted, best first), refresh its threshold and count
template <class C>
__device__ __forceinline__ void compact_slot(int j, u64* cand_w, u32* cnt_w, float* thr_w, int k, int lane,
                                             u32* gthr_w /* global, this wave's 32 queries */) {
    const u32 n_raw = cnt_w[j];
    const u32 n = n_raw < (u32)C::CAP ? n_raw : (u32)C::CAP;   // the screening scan lets the count run past a full slot
    u64 key[C::NPL];
    u32 rank[C::NPL];
#pragma unroll
    for (int p = 0; p < C::NPL; ++p) {
        const u32 e = lane + 64 * p;
        key[p] = (e < n) ? cand_w[j * C::CAP + e] : 0ull;
    }
    rank_keys<C::NPL>(key, n, rank);
    const u32 base = lds_addr(cand_w + j * C::CAP);
#pragma unroll
    for (int p = 0; p < C::NPL; ++p) {
        const u32 e = lane + 64 * p;
        if (e < n && rank[p] < (u32)k) {
            lds_store_b64(base + rank[p] * 8u, key[p]);
            if (rank[p] == (u32)(k - 1)) {
                lds_store_b32(lds_addr(thr_w + j), __float_as_uint(rmu_key_score(key[p])));
                // publish: this chunk's k-th best is a lower bound of the query's global k-th best
                __hip_atomic_fetch_max(gthr_w + j, (u32)(key[p] >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    if (lane == 0) lds_store_b32(lds_addr(cnt_w + j), n < (u32)k ? n : (u32)k);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}


}  // namespace

// scan_screen.hip -- fp16 hi/lo SCREENING scan + exact fp32 re-scoring (gfx950 / MI355X only).
//
// Same job as scan_topk.hip (the FLAT search behind server/RAGHelper.py:497-499), 5x cheaper in MFMA time, and
// still exact: the screen only PROPOSES candidates; the returned ids/scores come from an fp32 re-score in the
// exact kernel's own summation order, guarded by a sufficiency test with fallback to the exact scan.
//
// Split image (built at add time, same byte geometry as the fp32 matrix: 1536 B per 384-d row): for every 8
// consecutive k, [h0..h7 | l0..l7] with y = 64*x, h = fp16(y), l = fp16(y - h) (the 2^6 scale keeps l out of the fp16
// subnormal range for |x| >= ~1e-3; |x| must stay below ~1000).  Queries are split the same way.
//   4096 * s~ = sum (h.qh + h.ql + l.qh)                  (3 x v_mfma_f32_32x32x16_f16 per 16 k, ONE fp32 accumulator)
// Error vs the true dot product (unit scale): dropped l.ql and the residual of the 22-bit split are < 1e-6, the
// fp32 accumulation of 3*384 products is bounded by 1152 * 2^-24 * sum|x_i q_i| <= 6.9e-5; the exact fp32 kernel is
// itself within 384 * 2^-24 = 2.3e-5 of the truth.  EPS = 1e-4 * |x|max * |q| bounds |s~ - s_fp32| with margin.
// Sufficiency (per query): with the approximate top-K' (K' = 24) sorted, tau = k-th best s~.  If fewer than K'
// candidates exist, or s~[K'-1] < tau - 2*EPS, every row outside the candidate set has an exact score below k rows
// of the set, so the exact top-k is inside it.  Otherwise the query is flagged and the caller re-runs the batch on
// the exact scan.
#include <cstdlib>
#include <type_traits>
#include "rmu_common.h"
#include "scan_common.h"
#include "../../include/rmu.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int SD = 384;                 // floats per row (split image: same 1536 B)
constexpr int ROWB = SD * 4;
constexpr int S_RT = 32;                // rows per tile (4 waves share it, one 32-query group each)
constexpr int S_CKB = 768;              // bytes per row per chunk (12 steps of 64 B)
constexpr int S_U16 = S_CKB / 16;       // 48 units
constexpr int S_NCH = ROWB / S_CKB;     // 2 chunks per tile
constexpr int S_TS = S_CKB / 64;        // 12 steps per chunk, 3 MFMAs each
constexpr int S_RING = 4;                // 4 chunks (96 KiB) in flight: the f16 MFMAs are 5x shorter than the f32 ones and
                                         // two chunks no longer covered the ~4 us loaded L2/HBM latency (35 -> 25 ms)
constexpr int S_SLOT = S_RT * S_CKB;    // 24 KiB
constexpr int S_NI = S_RT * S_U16 / 256;  // 6 DMA wave-instructions per wave per chunk
struct ScreenCfg {
    // K' = 24 candidates per (chunk, query); the overflow check runs twice per tile (<= 16 appends per slot between
    // checks) and a compaction is due only after 16 further appends: CAP = 24 + 16 + 16
    static constexpr int CAP = 56, NPL = 1, A = 16;
    static constexpr int RING_BYTES = S_RING * S_SLOT;
    static constexpr int CAND_BYTES = 4 * 32 * CAP * 8;
    static constexpr int TRASH_OFF = RING_BYTES + CAND_BYTES + 4 * 32 * 4 + 4 * 32 * 4;
    static constexpr int GT_OFF = TRASH_OFF + 256 * 8;
    static constexpr int LDS_BYTES = GT_OFF + 4 * 256;
};
static_assert(ScreenCfg::LDS_BYTES <= 160 * 1024, "LDS");

extern __shared__ __attribute__((aligned(16))) char ssm[];

// fp32 rows [n, 384] -> split image [n, 1536 B]; one thread per group of 8 k
__global__ void k_split_rows(const float* __restrict__ src, char* __restrict__ dst, int64_t n_groups) {
    const int64_t gidx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gidx >= n_groups) return;
    const float* s = src + gidx * 8;
    f16x8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float y = s[e] * 64.0f;
        const _Float16 h = (_Float16)y;
        hi[e] = h;
        lo[e] = (_Float16)(y - (float)h);
    }
    *(f16x8*)(dst + gidx * 32) = hi;
    *(f16x8*)(dst + gidx * 32 + 16) = lo;
}

template <bool LA, int PF>   // PF = L2 prefetch distance in chunks (0 = off)
__global__ __launch_bounds__(256) void scan_screen_kernel(const ScanLaunch a) {
    using C = ScreenCfg;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave = query group, all waves read the same rows
    const int h = lane >> 5, j = lane & 31;
    int s_idx, qt;
    {
        const int b = blockIdx.x;
        if ((a.s_chunks & 7) == 0) {
            const int xcd = b & 7, m = b >> 3;
            qt = m % a.nqt;
            s_idx = (m / a.nqt) * 8 + xcd;
        } else {
            qt = b % a.nqt;
            s_idx = b / a.nqt;
        }
    }
    const int64_t tiles_total = (a.n_rows + S_RT - 1) / S_RT;
    const int64_t t0 = (int64_t)s_idx * a.tiles_per_chunk;
    int64_t t1 = t0 + a.tiles_per_chunk;
    if (t1 > tiles_total) t1 = tiles_total;
    const int ntiles = (int)(t1 > t0 ? t1 - t0 : 0);

    char* ring = ssm;
    u64* cand_w = (u64*)(ssm + C::RING_BYTES) + (size_t)w * 32 * C::CAP;
    u32* cnt_w = (u32*)(ssm + C::RING_BYTES + C::CAND_BYTES) + w * 32;
    float* thr_w = (float*)(ssm + C::RING_BYTES + C::CAND_BYTES + 4 * 32 * 4) + w * 32;
    const int q_idx = (qt * 4 + w) * 32 + j;
    const bool q_ok = q_idx < a.nq;
    ((u32*)(ssm + C::GT_OFF))[w * 64 + lane] = 0u;
    if (lane < 32) {
        cnt_w[lane] = 0;
        thr_w[lane] = q_ok ? -INFINITY : INFINITY;
    }
    float thr = q_ok ? -INFINITY : INFINITY, thr_loc = thr, thr_g = -INFINITY;
    float thr_s = thr;                                  // thr * 4096: filter threshold in the accumulator's scale (+-inf here)
    u32* gthr_w = a.gthr + (qt * 4 + w) * 32;
    const u32* gt_lds = (const u32*)(ssm + C::GT_OFF) + w * 64;

    // ---- query fragments (split image of the query batch): step T covers k [16T, 16T+16); lane half h owns 8 of them
    f16x8 qh[SD / 16], ql[SD / 16];
    {
        const char* qrow = (const char*)a.q + (size_t)(q_ok ? q_idx : 0) * ROWB + h * 32;
#pragma unroll
        for (int T = 0; T < SD / 16; ++T) {
            qh[T] = *(const f16x8*)(qrow + T * 64);
            ql[T] = *(const f16x8*)(qrow + T * 64 + 16);
        }
    }

    // ---- DMA source map: LDS unit f -> row f/48, physical unit f%48 holds logical unit p ^ (row & 15) ---------------
    u32 dma_off[S_NI];
#pragma unroll
    for (int n = 0; n < S_NI; ++n) {
        const int f = (n * 4 + w) * 64 + lane;
        const int i = f / S_U16, p = f % S_U16;
        dma_off[n] = (u32)(i * ROWB + (p ^ (i & 15)) * 16);
    }
    // L2 prefetch: the 8 query-tile workgroups of an XCD stream the same rows in lock step, so every chunk's first touch
    // is an HBM miss that all eight wait on.  Each wave therefore touches 48 of the 192 lines of the chunk PF chunks
    // ahead (one dword per 128-B line, result discarded) so that the LDS-DMA finds its lines in L2.  The load is part
    // of the chunk's VMEM group, i.e. it is covered by the same counted vmcnt; its destination is one dedicated VGPR.
    u32 pf_dummy = 0;
    const int pf_lane = (w * 64 + lane) < 192 ? (w * 64 + lane) : 191;
    const u32 pf_off = (u32)((pf_lane / 6) * ROWB + (pf_lane % 6) * 128);
    auto issue_chunk = [&](int cc) {
        int tl = cc / S_NCH;
        const int c = cc % S_NCH;
        if (tl >= ntiles) tl = ntiles - 1;
        const char* sbase = (const char*)a.x + ((t0 + tl) * S_RT) * (int64_t)ROWB + c * S_CKB;
        char* slot = ring + (cc % S_RING) * S_SLOT;
        if (PF > 0) {
            int tp = (cc + PF) / S_NCH;
            if (tp >= ntiles) tp = ntiles - 1;
            const char* pbase = (const char*)a.x + ((t0 + tp) * S_RT) * (int64_t)ROWB + ((cc + PF) % S_NCH) * S_CKB + pf_off;
            asm volatile("global_load_dword %0, %1, off" : "+v"(pf_dummy) : "v"(pbase));
        }
#pragma unroll
        for (int n = 0; n < S_NI; ++n)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(sbase + dma_off[n]),
                                             (__attribute__((address_space(3))) void*)(slot + (n * 4 + w) * 1024), 16, 0, 0);
    };
    // A fragments: row j, step t of the chunk: hi unit 4t + 2h, lo unit +1 (physical = logical ^ (row & 15))
    // (lo unit = hi unit ^ 1, i.e. byte ^ 16: all other address terms have bit 4 clear, so it folds into the base)
    int abase_hi[4], abase_lo[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        abase_hi[m] = j * S_CKB + (((4 * m + 2 * h) ^ (j & 15)) * 16);
        abase_lo[m] = abase_hi[m] ^ 16;
    }

    const u32 cnt_addr = lds_addr(cnt_w + j);
    const u32 cand_addr = lds_addr(cand_w + j * C::CAP);
    const u32 trash_addr = lds_addr(ssm + C::TRASH_OFF) + threadIdx.x * 8u;
    auto check_compact = [&]() {
        const u32 c = cnt_w[j];
        const u64 bal = __ballot(c > (u32)(C::CAP - C::A));
        u32 mask = (u32)bal | (u32)(bal >> 32);
        if (mask) {
            while (mask) {
                const int jj = __builtin_ctz(mask);
                mask &= mask - 1;
                compact_slot<C>(jj, cand_w, cnt_w, thr_w, a.k, lane, gthr_w);
            }
            thr_loc = thr_w[j];
            thr = fmaxf(thr_loc, thr_g);
            thr_s = thr * 4096.0f;
        }
    };
    struct Frag { f16x8 hi, lo; };
    // slot_off and t are compile-time at every call site: the address is a base VGPR + an immediate offset
    auto read_frag = [&](int slot_off, int t) -> Frag {
        Frag f;
        f.hi = *(const f16x8*)(ring + abase_hi[t & 3] + (slot_off + (t >> 2) * 256));
        f.lo = *(const f16x8*)(ring + abase_lo[t & 3] + (slot_off + (t >> 2) * 256));
        return f;
    };
    constexpr int GRP = S_NI + (PF > 0 ? 1 : 0);   // VMEM ops per chunk group
    constexpr int WAITN = LA ? GRP * (S_RING - 3) : GRP * (S_RING - 2);

    struct Acc { f32x16 a; };                          // 4096 * s~  (rows and queries are both scaled by 2^6)
    auto score = [](const Acc& p, int r) { return p.a[r] * (1.0f / 4096.0f); };
    u32 pmask = 0, wr_addr = 0, res_pos = 0;
    auto mask_slot = [&](const Acc& prev, int r) { pmask |= (prev.a[r] > thr_s) ? (1u << r) : 0u; };
    auto slow_begin = [&](u32 bits) {
        const u32 n = __builtin_popcount(pmask & bits);
        asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(res_pos) : "v"(cnt_addr), "v"(n) : "memory");
        wr_addr = cand_addr + res_pos * 8u;
    };
    auto slow_slot_r = [&](const Acc& prev, int r, int64_t rbase) {
        const u64 key = rmu_make_key(score(prev, r) + 0.0f, (u32)(rbase + (r & 3) + 8 * (r >> 2)));
        const bool pass = (pmask >> r) & 1u;
        lds_store_b64_nofence(pass ? wr_addr : trash_addr, key);
        wr_addr += pass ? 8u : 0u;
    };
    auto slow_end = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        check_compact();
    };

    int cc = 0;
    static_assert(LA && S_RING == 4 && S_NCH == 2, "static ring slots: tile parity p uses slots 2p, 2p+1");
    Frag a_cur;
    a_cur.hi = f16x8{}; a_cur.lo = f16x8{};
    // one tile (PAR = tile parity = which half of the ring it lives in): 72 MFMAs into `acc`; the previous tile's scores
    // are filtered in the first gaps (slot r behind MFMA r+1), then ONE branch (see scan_topk.hip for why)
    auto tile_body = [&](auto par_c, Acc& acc, Acc& prev, int64_t prev_rbase) {
        constexpr int PAR = decltype(par_c)::value;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc.a[r] = 0.f;
        pmask = 0;
#pragma unroll
        for (int c = 0; c < S_NCH; ++c, ++cc) {
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(WAITN) : "memory");
            __builtin_amdgcn_s_barrier();
            if (c == 0) {
                const u32 go = gt_lds[j];
                thr_g = (go && (a.share_thr & 1)) ? rmu_ord2f(go - 1u) : -INFINITY;
                thr = fmaxf(thr_loc, thr_g);
                thr_s = thr * 4096.0f;
            }
            if (c == S_NCH - 1)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gthr_w + j),
                                                 (__attribute__((address_space(3))) void*)(ssm + C::GT_OFF + w * 256), 4, 0, 16);
            issue_chunk(cc + S_RING - 1);
            constexpr int SL0 = 2 * PAR;                       // ring slot of chunk 0 of this tile
            const int slot_off = (SL0 + c) * S_SLOT;            // compile-time after unrolling
            const int next_off = ((SL0 + c + 1) % S_RING) * S_SLOT;
#pragma unroll
            for (int t = 0; t < S_TS; ++t) {
                const int gs = c * S_TS + t;
                const Frag a_nxt = (t + 1 < S_TS) ? read_frag(slot_off, t + 1) : read_frag(next_off, 0);
                if (gs == 6 && __builtin_expect(__any(pmask != 0), 0) && !(a.share_thr & 2)) {
#pragma unroll
                    for (int g2 = 0; g2 < 2; ++g2) {
                        slow_begin(0xFFu << (8 * g2));
#pragma unroll
                        for (int r = 8 * g2; r < 8 * g2 + 8; ++r) slow_slot_r(prev, r, prev_rbase);
                        slow_end();
                    }
                }
                const f16x8 bh = qh[gs], bl = ql[gs];
                acc.a = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_cur.hi, bh, acc.a, 0, 0, 0);
                if (gs < 6 && 3 * gs >= 1 && 3 * gs <= 16) mask_slot(prev, 3 * gs - 1);
                acc.a = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_cur.hi, bl, acc.a, 0, 0, 0);
                if (gs < 6 && 3 * gs + 1 <= 16) mask_slot(prev, 3 * gs);
                acc.a = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_cur.lo, bh, acc.a, 0, 0, 0);
                if (gs < 6 && 3 * gs + 2 <= 16) mask_slot(prev, 3 * gs + 1);
                a_cur = a_nxt;
                if (gs >= 6) {
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
                }
            }
        }
    };
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;

    if (ntiles > 0) {
#pragma unroll
        for (int c0 = 0; c0 < S_RING - 1; ++c0) issue_chunk(c0);
        if (LA) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GRP * (S_RING - 2)) : "memory");
            __builtin_amdgcn_s_barrier();
            a_cur = read_frag(0, 0);
        }
        Acc accA, accB;
#pragma unroll
        for (int r = 0; r < 16; ++r) accB.a[r] = -INFINITY;
        const int64_t lane_r0 = t0 * S_RT + 4 * h;
        auto rb = [&](int t) { return lane_r0 + (int64_t)t * S_RT; };
        tile_body(P0{}, accA, accB, rb(-1));              // tile 0 -> ring slots 0,1
        int tl = 1;
        for (; tl + 1 < ntiles; tl += 2) {
            tile_body(P1{}, accB, accA, rb(tl - 1));       // odd tile -> slots 2,3
            tile_body(P0{}, accA, accB, rb(tl));           // even tile -> slots 0,1
        }
        bool last_in_a = true;
        if (tl < ntiles) {
            tile_body(P1{}, accB, accA, rb(tl - 1));
            last_in_a = false;
        }
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(pf_dummy) : : "memory");
        {
            Acc last;
#pragma unroll
            for (int r = 0; r < 16; ++r) last.a[r] = last_in_a ? accA.a[r] : accB.a[r];
            const int64_t rbl = rb(ntiles - 1);
            pmask = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                mask_slot(last, r);
                if (rbl + (r & 3) + 8 * (r >> 2) >= a.n_rows) pmask &= ~(1u << r);
            }
            if (__any(pmask != 0)) {
#pragma unroll
                for (int g2 = 0; g2 < 2; ++g2) {
                    slow_begin(0xFFu << (8 * g2));
#pragma unroll
                    for (int r = 8 * g2; r < 8 * g2 + 8; ++r) slow_slot_r(last, r, rbl);
                    slow_end();
                }
            }
        }
    }
    // ---- emit: best K' approximate candidates of this (chunk, query), sorted ---------------------------------------------
    const int part = s_idx;
    for (int jj = 0; jj < 32; ++jj) {
        const int qq = (qt * 4 + w) * 32 + jj;
        if (qq >= a.nq) break;
        const u32 n = cnt_w[jj];
        u64 key[1];
        u32 rank[1];
        key[0] = ((u32)lane < n) ? cand_w[jj * C::CAP + lane] : 0ull;
        rank_keys<1>(key, n, rank);
        u64* dst = a.partial + ((size_t)part * a.nq + qq) * a.k;
        if ((u32)lane < n) {
            if (rank[0] < (u32)a.k) dst[rank[0]] = key[0];
        } else if (lane < a.k) {
            dst[lane] = 0ull;
        }
    }
}

// exact fp32 re-score of the K' candidates of each query, in the exact kernel's summation order:
// for t in 0..47, c in 0..3: acc = fma(x[8t+c], q[8t+c], acc); acc = fma(x[8t+4+c], q[8t+4+c], acc)
// (v_mfma_f32_32x32x2_f32 is a k-ordered fmaf chain; lanes < 32 hold k = 8t+c, lanes >= 32 hold k = 8t+4+c).
__global__ __launch_bounds__(256) void k_rescore(const u64* __restrict__ cand, int kp, const float* __restrict__ x,
                                                 const float* __restrict__ q, int64_t nq, int k, float eps_unit,
                                                 int64_t row_base, float* __restrict__ out_s, int64_t* __restrict__ out_r,
                                                 int* __restrict__ flagged) {
    const int lane = threadIdx.x & 63;
    const int64_t qi = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (qi >= nq) return;
    const u64 ck = lane < kp ? cand[qi * kp + lane] : 0ull;
    const bool valid = ck != 0ull;
    const float sa = valid ? rmu_key_score(ck) : -INFINITY;     // approximate score (sorted descending over lanes)
    const u32 row = valid ? rmu_key_row(ck) : 0u;
    const float* qv = q + qi * SD;
    const float* xv = x + (int64_t)row * SD;
    float acc = 0.f, qn2 = 0.f;
    for (int t = 0; t < SD / 8; ++t) {
        const f32x4 qa = *(const f32x4*)(qv + 8 * t), qb = *(const f32x4*)(qv + 8 * t + 4);
        const f32x4 xa = *(const f32x4*)(xv + 8 * t), xb = *(const f32x4*)(xv + 8 * t + 4);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            acc = fmaf(xa[c], qa[c], acc);
            acc = fmaf(xb[c], qb[c], acc);
            qn2 = fmaf(qa[c], qa[c], qn2);
            qn2 = fmaf(qb[c], qb[c], qn2);
        }
    }
    // sufficiency test on the approximate scores
    const int nvalid = __builtin_popcountll(__ballot(valid));
    const float tau = __shfl(sa, k - 1);                        // k-th best approximate score (or -inf)
    const float smin = __shfl(sa, kp - 1);                      // worst kept candidate
    const float eps = eps_unit * sqrtf(qn2);
    const bool complete = nvalid < kp;                           // every live row was a candidate
    const bool ok = complete || (smin < tau - 2.0f * eps);
    if (!ok && lane == 0) atomicAdd(flagged, 1);
    u64 key[1];
    u32 rank[1];
    key[0] = valid ? rmu_make_key(acc + 0.0f, row) : 0ull;
    rank_keys<1>(key, (u32)(kp < 64 ? kp : 64), rank);
    // keys of invalid lanes are 0 and rank below every valid one
    if (valid && rank[0] < (u32)k) {
        out_s[qi * k + rank[0]] = acc + 0.0f;
        out_r[qi * k + rank[0]] = (int64_t)row + row_base;
    }
    if (lane < k && lane >= nvalid) {
        out_s[qi * k + lane] = -INFINITY;
        out_r[qi * k + lane] = -1;
    }
}

}  // namespace

int rmu_split_launch(const float* src, void* dst, int64_t n_rows, hipStream_t s) {
    const int64_t groups = n_rows * (SD / 8);
    if (groups <= 0) return RMU_OK;
    hipLaunchKernelGGL(k_split_rows, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, s, src, (char*)dst, groups);
    return hipGetLastError() == hipSuccess ? RMU_OK : RMU_E_HIP;
}

template <bool LA, int PF>
static int screen_launch_cfg(const ScanLaunch* p, hipStream_t s) {
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute((const void*)scan_screen_kernel<LA, PF>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                ScreenCfg::LDS_BYTES) != hipSuccess)
            return RMU_E_HIP;
        attr = true;
    }
    hipLaunchKernelGGL((scan_screen_kernel<LA, PF>), dim3(p->grid), dim3(256), ScreenCfg::LDS_BYTES, s, *p);
    return hipGetLastError() == hipSuccess ? RMU_OK : RMU_E_HIP;
}

int rmu_screen_lds_bytes() { return ScreenCfg::LDS_BYTES; }

int rmu_screen_launch(const ScanLaunch* p, hipStream_t s) {
    // L2 software prefetch measured neutral-to-negative (26.3 -> 27.2 ms): off by default, RMU_SCREEN_PF=8 enables it
    static const int pf = getenv("RMU_SCREEN_PF") ? atoi(getenv("RMU_SCREEN_PF")) : 0;
    return pf > 0 ? screen_launch_cfg<true, 8>(p, s) : screen_launch_cfg<true, 0>(p, s);
}

int rmu_rescore_launch(const u64* cand, int kp, const float* x, const float* q, int64_t nq, int k, float eps_unit,
                       int64_t row_base, float* out_s, int64_t* out_r, int* flagged, hipStream_t s) {
    if (kp < k || kp > 64) return RMU_E_INVALID;
    hipLaunchKernelGGL(k_rescore, dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, s, cand, kp, x, q, nq, k, eps_unit, row_base,
                       out_s, out_r, flagged);
    return hipGetLastError() == hipSuccess ? RMU_OK : RMU_E_HIP;
}

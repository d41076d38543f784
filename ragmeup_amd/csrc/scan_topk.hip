// scan_topk.hip -- fused flat scan + top-k for an HBM-resident fp32 corpus (gfx950 / MI355X only).
//
// Replaces: the FLAT similarity search the reference reaches through
//   server/RAGHelper.py:497-499 (db.as_retriever -> Milvus col.search / pgvector `<=>`), SURVEY.md 8(a5).
//
// Shape of the work: S[n, q] = sum_k X[n, k] * Q[q, k] fused with a per-query running top-k; the N x B
// score matrix never exists.  One workgroup = 4 waves (one per SIMD, 512-register budget each):
//   * every wave keeps the fragments of 32 queries in REGISTERS (D/2 VGPRs) for the whole kernel,
//   * corpus rows stream HBM -> LDS through a ring of K-chunks filled by LDS-DMA
//     (global_load_lds_dwordx4, 1 KiB per wave-instruction, full 128-B lines, counted vmcnt so
//      RING-1 chunks stay in flight across the per-chunk s_barrier),
//   * v_mfma_f32_32x32x2_f32 (exact fp32, == a k-ordered fmaf chain) turns a 32-row x 32-query tile
//     into 16 accumulator registers per lane; lane l owns query (l & 31), so the top-k threshold is a
//     per-lane register compare,
//   * survivors are appended to a per-(wave,query) LDS candidate buffer (ds_add_rtn) and, when a
//     buffer could overflow, the wave sorts it with a shuffle bitonic network and keeps k
//     (new threshold = k-th key).
// WQ = number of distinct 32-query groups per workgroup: WQ=4 -> 128 queries/WG, every wave reads the
// same 32-row tile (f32-MFMA-bound regime); WQ=1 -> 32 queries/WG, the four waves take different
// 32-row slices of a 128-row tile (HBM-bound regime, small batches).
//
// k-permutation: lane (i, h=lane>>5) reads the float4 X[row i][8t+4h .. 8t+4h+3]; MFMA number c of
// that step multiplies component c, i.e. k = 8t+c (lanes<32) and k = 8t+4+c (lanes>=32).  The query
// fragments use the same map, so the product is the ordinary dot product with the k-order permuted.
//
// LDS image of a chunk: [RT rows][U16 16-byte units], unit index XOR-swizzled by the row so that the
// 16 lanes of every ds_read_b128 group hit 16 distinct 16-byte bank slots.  LDS-DMA writes
// lane-linear, so the swizzle is applied to the per-lane GLOBAL source address (guide rule 21).
#include <cstdlib>
#include <mutex>
#include "rmu_common.h"
#include "scan_common.h"
#include "../../include/rmu.h"

namespace {

template <int D_, int WQ_, int CKF_, int RING_, int CAP_, int NCHECK_, int EXP_ = 0, int LA_ = 1, int NT_ = 0>
struct Cfg {
    static constexpr bool NT = NT_ != 0;       // LDS-DMA cache policy of the corpus stream: nt (aux = 2) for read-once data
    static constexpr bool LA_ON = LA_ != 0;    // one-chunk look-ahead (costs one ring slot of in-flight data)
    static constexpr int EXP = EXP_;        // 0 = product; 1..3 = timing ablations (wrong results)
    static constexpr int D = D_;            // padded row length (floats)
    static constexpr int WQ = WQ_;          // query groups per workgroup
    static constexpr int RP = 4 / WQ_;      // row parts per tile
    static constexpr int RT = 32 * RP;      // rows per tile
    static constexpr int CKF = CKF_;        // floats per K-chunk
    static constexpr int U16 = CKF_ / 4;    // 16-byte units per row-chunk
    static constexpr int NCH = D_ / CKF_;   // chunks per tile
    static constexpr int TS = CKF_ / 8;     // ds_read_b128 steps per chunk
    static constexpr int RING = RING_;
    static constexpr int SLOT_BYTES = RT * CKF_ * 4;
    static constexpr int NI = RT * U16 / 256;  // DMA wave-instructions per wave per chunk
    static constexpr int CAP = CAP_;
    static constexpr int NPL = (CAP_ + 63) / 64;
    static constexpr int NCHECK = NCHECK_;
    static constexpr int A = 32 / NCHECK_;  // max appends per slot between overflow checks
    static constexpr int SWB = (U16 % 16 == 8) ? 8 : 4;  // swizzle block (units)
    static constexpr int RING_BYTES = RING_ * SLOT_BYTES;
    static constexpr int CAND_BYTES = 4 * 32 * CAP_ * 8;
    static constexpr int TRASH_OFF = RING_BYTES + CAND_BYTES + 4 * 32 * 4 + 4 * 32 * 4;
    static constexpr int GT_OFF = TRASH_OFF + 256 * 8;       // + one private 8-B trash slot per lane
    static constexpr int LDS_BYTES = GT_OFF + 4 * 256;       // + per-wave landing zone of the shared thresholds
    static_assert(D_ % CKF_ == 0 && CKF_ % 8 == 0, "chunking");
    static_assert((RT * U16) % 256 == 0, "DMA split");
    static_assert(U16 % 16 == 8 || U16 % 16 == 4 || U16 % 16 == 12, "swizzle classes");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
    static_assert(NI * (RING_ - 2) <= 63 && RING_ >= 2, "vmcnt field");
};

__device__ __forceinline__ int swz(int row, int swb) { return swb == 8 ? ((row >> 1) & 7) : ((row >> 2) & 3); }

extern __shared__ __attribute__((aligned(16))) char smem[];

template <class C>
__global__ __launch_bounds__(256) void scan_topk_kernel(const ScanLaunch a) {
    // device-side launch predicate (conditional re-runs behind the screening path): uniform over the grid
    int nq_eff = a.nq;
    if (a.cond.p) {
        const int c = *a.cond.p;
        if (c < a.cond.lo || c > a.cond.hi) return;
        if (a.cond.clamp && c < nq_eff) nq_eff = c;
    }
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = w % C::WQ;   // query group of this wave
    const int rp = w / C::WQ;  // row part of this wave
    const int h = lane >> 5;
    const int j = lane & 31;

    // ---- block -> (corpus chunk, query tile); query tiles of one chunk share an XCD's L2 ----------
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
    const int64_t tiles_total = (a.n_rows + C::RT - 1) / C::RT;
    const int64_t t0 = (int64_t)s_idx * a.tiles_per_chunk;
    int64_t t1 = t0 + a.tiles_per_chunk;
    if (t1 > tiles_total) t1 = tiles_total;
    const int ntiles = (int)(t1 > t0 ? t1 - t0 : 0);

    // ---- LDS carve (one object: see guide "three .s-level traps" (a)) -----------------------------
    char* ring = smem;
    u64* cand_w = (u64*)(smem + C::RING_BYTES) + (size_t)w * 32 * C::CAP;
    u32* cnt_w = (u32*)(smem + C::RING_BYTES + C::CAND_BYTES) + w * 32;
    float* thr_w = (float*)(smem + C::RING_BYTES + C::CAND_BYTES + 4 * 32 * 4) + w * 32;

    const int q_idx = (qt * C::WQ + g) * 32 + j;
    const bool q_ok = q_idx < nq_eff;
    ((u32*)(smem + C::GT_OFF))[w * 64 + lane] = 0u;   // landing zone of the shared thresholds: 0 = no bound
    if (lane < 32) {
        cnt_w[lane] = 0;
        thr_w[lane] = q_ok ? -INFINITY : INFINITY;
    }
    float thr = q_ok ? -INFINITY : INFINITY;       // effective filter = max(local k-th best, shared bound)
    float thr_loc = thr, thr_g = -INFINITY;
    // shared thresholds of this wave's 32 queries (global) and their LDS landing zone
    u32* gthr_w = a.gthr + (qt * C::WQ + g) * 32;
    const u32* gt_lds = (const u32*)(smem + C::GT_OFF) + w * 64;

    // ---- query fragments -> registers --------------------------------------------------------------
    f32x4 qf[C::D / 8];
    {
        // padded query columns read row 0 (valid memory): their threshold is +inf and nothing is emitted.
        // No select here: pure load -> MFMA operand lets the allocator park fragments in AGPRs.
        const float* qrow = a.q + (size_t)(q_ok ? q_idx : 0) * C::D + 4 * h;
#pragma unroll
        for (int t = 0; t < C::D / 8; ++t) qf[t] = *(const f32x4*)(qrow + 8 * t);
    }

    // ---- per-lane DMA source map (constant over the kernel) ----------------------------------------
    // byte offset of this lane's 16-B unit inside a [RT x D] tile (de-swizzled); the per-chunk base is a
    // scalar, so a DMA issue is one global_load_lds with saddr + 32-bit voffset and no VALU at all.
    // Rows past n_rows are read (the index keeps >= 128 slack rows) and dropped by the epilogue.
    u32 dma_off[C::NI];
#pragma unroll
    for (int n = 0; n < C::NI; ++n) {
        const int f = (n * 4 + w) * 64 + lane;
        const int i = f / C::U16, p = f % C::U16;
        dma_off[n] = (u32)(i * C::D + 4 * (p ^ swz(i, C::SWB))) * 4u;
    }

    auto issue_chunk = [&](int cc) {   // cc = running chunk number inside this workgroup
        int tl = cc / C::NCH;
        const int c = cc % C::NCH;
        if (tl >= ntiles) tl = ntiles - 1;   // tail: harmless reloads keep the vmcnt bookkeeping uniform
        const char* sbase = (const char*)(a.x + ((t0 + tl) * C::RT) * (int64_t)C::D + c * C::CKF);
        char* slot = ring + (cc % C::RING) * C::SLOT_BYTES;
#pragma unroll
        for (int n = 0; n < C::NI; ++n) {
            // (the aux operand must be a literal: a dependent constant here silently drops the kernel's host-side handle)
            if (C::NT)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(sbase + dma_off[n]),
                                                 (__attribute__((address_space(3))) void*)(slot + (n * 4 + w) * 1024), 16, 0, 2);
            else
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(sbase + dma_off[n]),
                                                 (__attribute__((address_space(3))) void*)(slot + (n * 4 + w) * 1024), 16, 0, 0);
        }
    };

    // A-fragment read offsets: row (32*rp + j), unit (2t+h) ^ swz
    const int rowi = 32 * rp + j;
    int abase[C::SWB / 2];
#pragma unroll
    for (int m = 0; m < C::SWB / 2; ++m)
        abase[m] = (rowi * C::U16 + ((2 * m + h) ^ swz(rowi, C::SWB))) * 16;

    const u32 cnt_addr = (u32)(uintptr_t)(__attribute__((address_space(3))) char*)(char*)(cnt_w + j);
    const u32 cand_addr = (u32)(uintptr_t)(__attribute__((address_space(3))) char*)(char*)(cand_w + j * C::CAP);

    auto check_compact = [&]() {
        const u32 c = cnt_w[j];
        const u64 bal = __ballot(c > (u32)(C::CAP - C::A));
        u32 mask = (u32)bal | (u32)(bal >> 32);
        if (mask) {
            const long long tc0 = (C::EXP == 7) ? clock64() : 0;
            while (mask) {
                const int jj = __builtin_ctz(mask);
                mask &= mask - 1;
                compact_slot<C>(jj, cand_w, cnt_w, thr_w, a.k, lane, gthr_w);
                if (C::EXP == 7 && lane == 0 && a.dbg) atomicAdd((unsigned long long*)a.dbg + 1, 1ull);
            }
            thr_loc = thr_w[j];
            thr = fmaxf(thr_loc, thr_g);
            if (C::EXP == 7 && lane == 0 && a.dbg) atomicAdd((unsigned long long*)a.dbg + 5, (unsigned long long)(clock64() - tc0));
        }
    };

    // A-fragment of step t of the chunk living in ring slot `slot_off` (bytes)
    auto read_frag = [&](int slot_off, int t) -> f32x4 {
        const int off = abase[t % (C::SWB / 2)] + (t / (C::SWB / 2)) * (C::SWB * 16);
        return *(const f32x4*)(ring + slot_off + off);
    };
    // LA (look-ahead): chunk cc+1 has landed when barrier cc is passed, so the first fragment of the
    // next chunk is read BEFORE its barrier and the MFMA pipe never waits on an LDS round trip.
    constexpr bool LA = C::RING >= 3 && C::LA_ON;
    constexpr int WAITN = LA ? C::NI * (C::RING - 3) : C::NI * (C::RING - 2);

    // ---- threshold filter of the PREVIOUS tile, hidden behind this tile's MFMAs ------------------------
    // Two accumulators alternate.  While the first 16 MFMAs of a tile issue, each gap builds one bit of a
    // per-lane pass mask (prev[r] > thr; tombstoned rows are NaN and never pass).  ONE uniform branch per
    // tile then picks a plain continuation or a "slow" one whose MFMA gaps carry the appends:
    // one ds_add_rtn by popcount reserves slots, 16 exec-predicated ds_write_b64 store the keys.
    // (On a lone wave per SIMD every skip-branch costs ~100 cycles -- 16 per tile was 13% of the kernel.)
    // All LDS writes are inline asm: a compiler-generated LDS write is ordered behind the in-flight
    // LDS-DMA with s_waitcnt vmcnt(0) and would drain the ring.
    u32 pmask = 0;
    auto mask_slot = [&](const f32x16& prev, int r) { pmask |= (prev[r] > thr) ? (1u << r) : 0u; };
    u32 wr_addr = 0;
    u32 res_pos = 0;
    auto slow_issue = [&](u32 bits) {   // reserve popc(pmask & bits) entries of this lane's query slot
        const u32 n = __builtin_popcount(pmask & bits);
        asm volatile("ds_add_rtn_u32 %0, %1, %2" : "=v"(res_pos) : "v"(cnt_addr), "v"(n));
    };
    auto slow_wait = [&]() {            // ... one K-step later the returned position is consumed
        const long long tw0 = (C::EXP == 7) ? clock64() : 0;
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(res_pos));
        if (C::EXP == 7 && lane == 0 && a.dbg) atomicAdd((unsigned long long*)a.dbg + 6, (unsigned long long)(clock64() - tw0));
        if (C::EXP == 5) res_pos &= 31u;   // ablation: appends every tile, never compacted
        wr_addr = cand_addr + res_pos * 8u;
    };
    auto slow_begin = [&](u32 bits) { slow_issue(bits); slow_wait(); };
    // store prev[r]'s key at wr_addr (and advance it) in the lanes whose pass bit r is set.  No branch and
    // no EXEC games (32 EXEC rewrites per tile stalled the MFMA stream for thousands of cycles): every
    // lane stores, the non-passing ones into their private trash slot.
    const u32 trash_addr = lds_addr(smem + C::TRASH_OFF) + threadIdx.x * 8u;
    auto slow_slot_r = [&](const f32x16& prev, int r, int64_t rbase) {
        const u64 key = rmu_make_key(prev[r] + 0.0f, (u32)(a.row0 + rbase + (r & 3) + 8 * (r >> 2)));
        const bool pass = (pmask >> r) & 1u;   // r is a compile-time constant after unrolling
        lds_store_b64_nofence(pass ? wr_addr : trash_addr, key);
        wr_addr += pass ? 8u : 0u;
    };
    auto slow_end = [&]() {
        const long long te0 = (C::EXP == 7) ? clock64() : 0;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const long long te1 = (C::EXP == 7) ? clock64() : 0;
        if (C::EXP != 5) check_compact();
        if (C::EXP == 7 && lane == 0 && a.dbg) {
            atomicAdd((unsigned long long*)a.dbg + 7, (unsigned long long)(te1 - te0));
            atomicAdd((unsigned long long*)a.dbg + 8, (unsigned long long)(clock64() - te1));
        }
    };
    int cc = 0;
    f32x4 a_cur = {0.f, 0.f, 0.f, 0.f};
    // one K-step: prefetch the next fragment, 4 MFMAs, hook(i) after MFMA i (i = 0..3)
    auto do_step = [&](f32x16& acc, int c, int t, int slot_off, int next_off, bool pin, auto&& hook) {
        f32x4 a_nxt = a_cur;
        if (t + 1 < C::TS) a_nxt = read_frag(slot_off, t + 1);
        else if (LA) a_nxt = read_frag(next_off, 0);
        const f32x4 qv = qf[c * C::TS + t];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur.x, qv.x, acc, 0, 0, 0);
        hook(0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur.y, qv.y, acc, 0, 0, 0);
        hook(1);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur.z, qv.z, acc, 0, 0, 0);
        hook(2);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur.w, qv.w, acc, 0, 0, 0);
        hook(3);
        a_cur = a_nxt;
        if (pin) {
            // pin the software pipeline: the NEXT fragment's ds_read issues ahead of this step's four
            // MFMAs (hipcc otherwise sinks it behind them and exposes the LDS round trip)
            if (t + 1 < C::TS || LA) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        }
    };
    auto no_hook = [](int) {};

    // One tile: a single MFMA chain into `acc`; the previous tile's accumulator `prev` is filtered in its
    // gaps, keyed on the global step gs = c*TS + t:
    //   gs 0..3  pass-mask bits (slot r behind MFMA r+1, when prev's last MFMA has retired)
    //   gs 4     the ONE branch of the tile: if any lane passed, an out-of-line block (no MFMAs inside)
    //            reserves slots with one ds_add_rtn by popcount and stores the keys; ~7% of tiles at 10M.
    // Measured alternatives on MI355X (10M x 384, B=1024): 16 skip-branches/tile 128 TF; a fast and a slow copy
    // of the MFMA chain (appends in MFMA gaps) 135.6 TF but hipcc then copies 16 AGPRs + drains the pipe
    // at every tile end; ~20 not-taken scalar branches/tile 132 TF (every branch stalls a lone in-order wave).
    auto tile_body = [&](f32x16& acc, f32x16& prev, int64_t prev_rbase) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        pmask = 0;
        bool anyp = false;
#pragma unroll
        for (int c = 0; c < C::NCH; ++c, ++cc) {
            // own DMA of chunk cc (+1 with look-ahead) has landed; own LDS reads have returned
            if (C::EXP == 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(WAITN) : "memory");
            if (C::EXP != 2) __builtin_amdgcn_s_barrier();
            if (c == 0) {
                // shared threshold fetched by last tile's DMA: pass iff v >= bound  <=>  v > nextbelow(bound)
                const u32 go = gt_lds[j];
                thr_g = (go && a.share_thr) ? rmu_ord2f(go - 1u) : -INFINITY;
                thr = fmaxf(thr_loc, thr_g);
            }
            if (c == C::NCH - 1) {
                // refresh for the next tile: one 4-B-per-lane LDS-DMA, issued BEFORE this chunk's group so the
                // next counted vmcnt covers it; sc1 = skip the (never refreshed) per-CU L1
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gthr_w + j),
                                                 (__attribute__((address_space(3))) void*)(smem + C::GT_OFF + w * 256), 4, 0, 16);
            }
            issue_chunk(cc + C::RING - 1);   // refills the slot every wave finished reading last round
            const int slot_off = (cc % C::RING) * C::SLOT_BYTES;
            const int next_off = ((cc + 1) % C::RING) * C::SLOT_BYTES;
            if (!LA) a_cur = read_frag(slot_off, 0);
#pragma unroll
            for (int t = 0; t < C::TS; ++t) {
                const int gs = c * C::TS + t;
                if (C::EXP == 3 || gs > 4) {
                    do_step(acc, c, t, slot_off, next_off, true, no_hook);
                } else if (gs < 4) {
                    do_step(acc, c, t, slot_off, next_off, false, [&](int i) {
                        const int m = 4 * gs + i;
                        if (m >= 1) mask_slot(prev, m - 1);
                        if (m == 15) mask_slot(prev, 15);
                    });
                } else {   // gs == 4: the one branch of the tile
                    anyp = __any(pmask != 0) && C::EXP != 4;
                    if (C::EXP == 7 && lane == 0 && a.dbg) atomicAdd((unsigned long long*)a.dbg + 3, 1ull);
                    if (__builtin_expect(anyp, 0)) {
                        if (C::EXP == 7 && a.dbg) {
                            if (lane == 0) atomicAdd((unsigned long long*)a.dbg + 0, 1ull);
                            atomicAdd((unsigned long long*)a.dbg + 2, (unsigned long long)__builtin_popcount(pmask));
                        }
                        slow_begin(C::NCHECK == 2 ? 0x00FFu : 0xFFFFu);
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            if (C::NCHECK == 2 && r == 8) { slow_end(); slow_begin(0xFF00u); }
                            slow_slot_r(prev, r, prev_rbase);
                        }
                        slow_end();
                    }
                    do_step(acc, c, t, slot_off, next_off, true, no_hook);
                }
            }
        }
    };

    if (ntiles > 0) {
        // ---- prologue: RING-1 chunks in flight -----------------------------------------------------
#pragma unroll
        for (int c0 = 0; c0 < C::RING - 1; ++c0) issue_chunk(c0);
        if (LA) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::NI * (C::RING - 2)) : "memory");
            __builtin_amdgcn_s_barrier();
            a_cur = read_frag(0, 0);
        }
        f32x16 accA, accB;
#pragma unroll
        for (int r = 0; r < 16; ++r) accB[r] = -INFINITY;   // "previous tile" of tile 0: nothing passes
        const int64_t lane_r0 = t0 * C::RT + 32 * rp + 4 * h;   // this lane's first row in tile 0
        auto rb = [&](int t) { return lane_r0 + (int64_t)t * C::RT; };
        // peel + two-tile loop: at the loop head accA always holds the tile to be filtered next
        tile_body(accA, accB, rb(-1));
        int tl = 1;
        for (; tl + 1 < ntiles; tl += 2) {
            tile_body(accB, accA, rb(tl - 1));
            tile_body(accA, accB, rb(tl));
        }
        bool last_in_a = true;
        if (tl < ntiles) {
            tile_body(accB, accA, rb(tl - 1));
            last_in_a = false;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // ---- filter of the chunk's last tile (not overlapped).  Rows >= n_rows exist only here (the
        // globally last tile is some chunk's last tile): blank them with a per-lane validity mask.
        {
            f32x16 last;
#pragma unroll
            for (int r = 0; r < 16; ++r) last[r] = last_in_a ? accA[r] : accB[r];
            if (C::EXP == 3) asm volatile("" ::"v"(last[0]), "v"(last[15]));
            const int64_t rbl = rb(ntiles - 1);
            pmask = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                mask_slot(last, r);
                if (rbl + (r & 3) + 8 * (r >> 2) >= a.n_rows) pmask &= ~(1u << r);
            }
            if (__any(pmask != 0) && C::EXP != 3) {
                slow_begin(C::NCHECK == 2 ? 0x00FFu : 0xFFFFu);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if (C::NCHECK == 2 && r == 8) { slow_end(); slow_begin(0xFF00u); }
                    slow_slot_r(last, r, rbl);
                }
                slow_end();
            }
        }
    }

    // ---- final: sort every slot, emit k keys per (part, query) ---------------------------------------
    const int part = s_idx * C::RP + rp;
    for (int jj = 0; jj < 32; ++jj) {
        const int qq = (qt * C::WQ + g) * 32 + jj;
        if (qq >= nq_eff) break;
        const u32 n = cnt_w[jj];
        u64 key[C::NPL];
        u32 rank[C::NPL];
#pragma unroll
        for (int p = 0; p < C::NPL; ++p) {
            const u32 e = lane + 64 * p;
            key[p] = (e < n) ? cand_w[jj * C::CAP + e] : 0ull;
        }
        rank_keys<C::NPL>(key, n, rank);
        u64* dst = a.partial + ((size_t)part * a.nq + qq) * a.k;
#pragma unroll
        for (int p = 0; p < C::NPL; ++p) {
            const u32 e = lane + 64 * p;
            if (e < n) {
                if (rank[p] < (u32)a.k) dst[rank[p]] = key[p];
            } else if (e < (u32)a.k) {
                dst[e] = 0ull;   // fewer than k candidates: pad (e >= n are exactly the unfilled ranks)
            }
        }
    }
}

template <class C>
int launch_cfg(const ScanLaunch* p, hipStream_t s) {
    // function-local static: initialised exactly once, thread-safe (C++11)
    static const hipError_t attr_rc =
        hipFuncSetAttribute((const void*)scan_topk_kernel<C>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
    if (attr_rc != hipSuccess) return RMU_E_HIP;
    hipLaunchKernelGGL(scan_topk_kernel<C>, dim3(p->grid), dim3(256), C::LDS_BYTES, s, *p);
    return hipGetLastError() == hipSuccess ? RMU_OK : RMU_E_HIP;
}

// geometry table: (WQ) x (k class).  kv 0: k <= 32 (CAP 64, one check per tile); kv 1: k <= 112.
//                                 D    WQ CKF RING CAP NCHECK
template <int D> using C_w4_k0 = Cfg<D, 4, 96, 4, 64, 1>;    // 48 KiB ring + 64 KiB candidates
template <int D> using C_w4_k1 = Cfg<D, 4, 96, 2, 128, 2>;   // 24 KiB ring + 128 KiB candidates
template <int D> using C_w2_k0 = Cfg<D, 2, 96, 3, 64, 1, 0, 0>;   // 72 KiB ring, no look-ahead (2 chunks in flight)
template <int D> using C_w2_k1 = Cfg<D, 2, 48, 2, 128, 2>;   // 24 KiB ring
template <int D> using C_w1_k0 = Cfg<D, 1, 48, 3, 64, 1, 0, 0>;   // 72 KiB ring; HBM-bound: no look-ahead -> 2 chunks in flight (5.0 -> 5.8 TB/s)
// one query tile (<= 64 queries): every corpus byte is read by exactly one workgroup -> non-temporal stream
template <int D> using C_w2_k0_nt = Cfg<D, 2, 96, 3, 64, 1, 0, 0, 1>;
template <int D> using C_w1_k0_nt = Cfg<D, 1, 48, 3, 64, 1, 0, 0, 1>;

template <int D>
int launch_d(const ScanLaunch* p, hipStream_t s) {
    switch (p->wq * 2 + p->kv) {
        case 8: {
#ifdef RMU_DEBUG_KERNELS      // timing ablations / cycle counters (wrong results by design): python -m ragmeup_amd.build --debug-kernels
            static const int exp = rmu_env("RMU_SCAN_EXP") ? atoi(rmu_env("RMU_SCAN_EXP")) : 0;
            if (D == 384 && exp == 1) return launch_cfg<Cfg<384, 4, 96, 4, 64, 1, 1>>(p, s);
            if (D == 384 && exp == 2) return launch_cfg<Cfg<384, 4, 96, 4, 64, 1, 2>>(p, s);
            if (D == 384 && exp == 3) return launch_cfg<Cfg<384, 4, 96, 4, 64, 1, 3>>(p, s);
            if (D == 384 && exp == 4) return launch_cfg<Cfg<384, 4, 96, 4, 64, 1, 4>>(p, s);
            if (D == 384 && exp == 5) return launch_cfg<Cfg<384, 4, 96, 4, 64, 1, 5>>(p, s);
            if (D == 384 && exp == 6) return launch_cfg<Cfg<384, 4, 96, 4, 64, 1, 6>>(p, s);
            if (D == 384 && exp == 7) return launch_cfg<Cfg<384, 4, 96, 4, 64, 1, 7>>(p, s);
#endif
            return launch_cfg<C_w4_k0<D>>(p, s);
        }
        case 9: return launch_cfg<C_w4_k1<D>>(p, s);
        case 4: return p->nt ? launch_cfg<C_w2_k0_nt<D>>(p, s) : launch_cfg<C_w2_k0<D>>(p, s);
        case 5: return launch_cfg<C_w2_k1<D>>(p, s);
        case 2: return p->nt ? launch_cfg<C_w1_k0_nt<D>>(p, s) : launch_cfg<C_w1_k0<D>>(p, s);
        default: return RMU_E_INVALID;
    }
}

template <int D>
int lds_d(int wq, int kv) {
    switch (wq * 2 + kv) {
        case 8: return C_w4_k0<D>::LDS_BYTES;
        case 9: return C_w4_k1<D>::LDS_BYTES;
        case 4: return C_w2_k0<D>::LDS_BYTES;
        case 5: return C_w2_k1<D>::LDS_BYTES;
        case 2: return C_w1_k0<D>::LDS_BYTES;
        default: return -1;
    }
}

}  // namespace

int rmu_scan_plan(ScanLaunch* p) {
    if (p->k < 1 || p->k > RMU_MAX_K || p->nq < 1 || p->n_rows < 0) return RMU_E_INVALID;
    if (p->dpad != 192 && p->dpad != 384 && p->dpad != 768) return RMU_E_INVALID;
    p->kv = p->k <= 32 ? 0 : 1;
    p->wq = p->nq <= 32 ? 1 : (p->nq <= 64 ? 2 : 4);
    if (p->kv == 1 && p->wq == 1) p->wq = 2;   // no LDS room for a 128-row tile next to 128-deep buffers
    const int rt = 32 * (4 / p->wq);
    p->nqt = (p->nq + 32 * p->wq - 1) / (32 * p->wq);
    const int64_t tiles_total = (p->n_rows + rt - 1) / rt;
    // corpus chunks: a multiple of 8 (XCD-aware block map) that makes grid = S*nqt fill 256 CUs evenly
    int best_s = 8;
    double best_eff = -1.0;
    for (int s = 8; s <= 256; s += 8) {
        const int64_t total = (int64_t)s * p->nqt;
        const double eff = (double)total / (double)(((total + 255) / 256) * 256);
        if (eff > best_eff + 1e-9) { best_eff = eff; best_s = s; }
        if (total >= 256 && eff > 0.999) break;
    }
    int s = best_s;
    if (tiles_total < s) s = tiles_total > 0 ? (int)tiles_total : 1;
    p->tiles_per_chunk = (int)((tiles_total + s - 1) / s);
    if (p->tiles_per_chunk < 1) p->tiles_per_chunk = 1;
    // drop empty trailing chunks (keeps the multiple-of-8 property only when nothing is dropped)
    const int64_t used = (tiles_total + p->tiles_per_chunk - 1) / p->tiles_per_chunk;
    if (used > 0 && used < s) s = (int)used;
    p->s_chunks = s;
    p->grid = s * p->nqt;
    p->parts = s * (4 / p->wq);
    static const int nt_env = rmu_env("RMU_NT") ? atoi(rmu_env("RMU_NT")) : 1;
    p->nt = (nt_env && p->nqt == 1 && p->kv == 0 && p->wq <= 2) ? 1 : 0;
    p->lds_bytes = p->dpad == 384 ? lds_d<384>(p->wq, p->kv)
                 : p->dpad == 768 ? lds_d<768>(p->wq, p->kv) : lds_d<192>(p->wq, p->kv);
    return p->lds_bytes > 0 ? RMU_OK : RMU_E_INVALID;
}

int rmu_scan_launch(const ScanLaunch* p, hipStream_t s) {
    switch (p->dpad) {
        case 384: return launch_d<384>(p, s);
        case 768: return launch_d<768>(p, s);
        case 192: return launch_d<192>(p, s);
        default: return RMU_E_INVALID;
    }
}

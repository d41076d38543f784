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
#include "rmu_common.h"
#include "../../include/rmu.h"

namespace {

template <int D_, int WQ_, int CKF_, int RING_, int CAP_, int NCHECK_>
struct Cfg {
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
    static constexpr int NPL = CAP_ / 64;
    static constexpr int NCHECK = NCHECK_;
    static constexpr int A = 32 / NCHECK_;  // max appends per slot between overflow checks
    static constexpr int SWB = (U16 % 16 == 8) ? 8 : 4;  // swizzle block (units)
    static constexpr int RING_BYTES = RING_ * SLOT_BYTES;
    static constexpr int CAND_BYTES = 4 * 32 * CAP_ * 8;
    static constexpr int LDS_BYTES = RING_BYTES + CAND_BYTES + 4 * 32 * 4 + 4 * 32 * 4;
    static_assert(D_ % CKF_ == 0 && CKF_ % 8 == 0, "chunking");
    static_assert((RT * U16) % 256 == 0, "DMA split");
    static_assert(U16 % 16 == 8 || U16 % 16 == 4 || U16 % 16 == 12, "swizzle classes");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
    static_assert(NI * (RING_ - 2) <= 63, "vmcnt field");
};

__device__ __forceinline__ int swz(int row, int swb) { return swb == 8 ? ((row >> 1) & 7) : ((row >> 2) & 3); }

extern __shared__ __attribute__((aligned(16))) char smem[];

template <class C>
__device__ __forceinline__ void compact_slot(int j, u64* cand_w, u32* cnt_w, float* thr_w, int k, int lane) {
    const u32 n = cnt_w[j];
    u64 key[C::NPL];
#pragma unroll
    for (int p = 0; p < C::NPL; ++p) {
        const u32 e = lane + 64 * p;
        key[p] = (e < n) ? cand_w[j * C::CAP + e] : 0ull;
    }
    rmu_bitonic_sort_desc<C::NPL>(key, lane);
    const u32 nn = n < (u32)k ? n : (u32)k;
#pragma unroll
    for (int p = 0; p < C::NPL; ++p) {
        const u32 e = lane + 64 * p;
        if (e < nn) cand_w[j * C::CAP + e] = key[p];
    }
    if (n >= (u32)k) {
        u64 sel = key[0];
        if (C::NPL > 1 && ((k - 1) >> 6)) sel = key[C::NPL > 1 ? 1 : 0];
        const u64 kth = __shfl(sel, (k - 1) & 63);
        if (lane == 0) thr_w[j] = rmu_key_score(kth);
    }
    if (lane == 0) cnt_w[j] = nn;
}

template <class C>
__global__ __launch_bounds__(256) void scan_topk_kernel(const ScanLaunch a) {
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
    const bool q_ok = q_idx < a.nq;
    if (lane < 32) {
        cnt_w[lane] = 0;
        thr_w[lane] = q_ok ? -INFINITY : INFINITY;
    }
    float thr = q_ok ? -INFINITY : INFINITY;

    // ---- query fragments -> registers --------------------------------------------------------------
    f32x4 qf[C::D / 8];
    {
        const float* qrow = a.q + (size_t)(q_ok ? q_idx : 0) * C::D + 4 * h;
#pragma unroll
        for (int t = 0; t < C::D / 8; ++t) {
            f32x4 v = *(const f32x4*)(qrow + 8 * t);
            if (!q_ok) v = f32x4{0.f, 0.f, 0.f, 0.f};
            qf[t] = v;
        }
    }

    // ---- per-lane DMA source map (constant over the kernel) ----------------------------------------
    int dma_row[C::NI];   // row inside the tile
    int dma_col[C::NI];   // float offset inside the chunk (already de-swizzled)
#pragma unroll
    for (int n = 0; n < C::NI; ++n) {
        const int f = (n * 4 + w) * 64 + lane;
        const int i = f / C::U16, p = f % C::U16;
        dma_row[n] = i;
        dma_col[n] = 4 * (p ^ swz(i, C::SWB));
    }
    const int64_t last_row = a.n_rows - 1;

    auto issue_chunk = [&](int cc) {   // cc = running chunk number inside this workgroup
        int tl = cc / C::NCH;
        const int c = cc % C::NCH;
        if (tl >= ntiles) tl = ntiles - 1;   // tail: harmless reloads keep the vmcnt bookkeeping uniform
        const int64_t row0 = (t0 + tl) * C::RT;
        char* slot = ring + (cc % C::RING) * C::SLOT_BYTES;
#pragma unroll
        for (int n = 0; n < C::NI; ++n) {
            int64_t r = row0 + dma_row[n];
            r = r > last_row ? last_row : r;
            const float* src = a.x + r * (int64_t)C::D + c * C::CKF + dma_col[n];
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(slot + (n * 4 + w) * 1024),
                                             16, 0, 0);
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
            while (mask) {
                const int jj = __builtin_ctz(mask);
                mask &= mask - 1;
                compact_slot<C>(jj, cand_w, cnt_w, thr_w, a.k, lane);
            }
            thr = thr_w[j];
        }
    };

    if (ntiles > 0) {
        // ---- prologue: RING-1 chunks in flight -----------------------------------------------------
#pragma unroll
        for (int cc = 0; cc < C::RING - 1; ++cc) issue_chunk(cc);

        int cc = 0;
        for (int tl = 0; tl < ntiles; ++tl) {
            f32x16 acc = {0.f};
#pragma unroll
            for (int c = 0; c < C::NCH; ++c, ++cc) {
                // chunk cc has landed once at most (RING-2) younger groups are outstanding
                // lgkmcnt(0): this wave's LDS reads of the slot about to be refilled have returned too
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(C::NI * (C::RING - 2)) : "memory");
                __builtin_amdgcn_s_barrier();
                issue_chunk(cc + C::RING - 1);   // overwrites the slot every wave finished last round
                const char* slot = ring + (cc % C::RING) * C::SLOT_BYTES;
#pragma unroll
                for (int t = 0; t < C::TS; ++t) {
                    const int off = abase[t % (C::SWB / 2)] + (t / (C::SWB / 2)) * (C::SWB * 16);
                    const f32x4 av = *(const f32x4*)(slot + off);
                    const f32x4 qv = qf[c * C::TS + t];
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, qv.x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, qv.y, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, qv.z, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, qv.w, acc, 0, 0, 0);
                }
            }
            // ---- epilogue: threshold filter + LDS append ---------------------------------------------
            const int64_t rbase = (t0 + tl) * C::RT + 32 * rp + 4 * h;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t row = rbase + (r & 3) + 8 * (r >> 2);
                const float v = acc[r] + 0.0f;   // canonicalise -0
                // tombstoned rows are NaN-poisoned in HBM (rmu_index_remove_rows): NaN > thr is false
                if (v > thr && row < a.n_rows) {
                    // inline asm: a compiler-generated LDS write would be ordered behind the in-flight
                    // LDS-DMA with s_waitcnt vmcnt(0), draining the ring on most tiles
                    const u64 key = rmu_make_key(v, (u32)row);
                    u32 pos;
                    asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)"
                                 : "=v"(pos) : "v"(cnt_addr), "v"(1u) : "memory");
                    const u32 dst = cand_addr + pos * 8u;
                    asm volatile("ds_write_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" ::"v"(dst), "v"(key) : "memory");
                }
                if (C::NCHECK == 2 && r == 7) check_compact();
            }
            check_compact();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }

    // ---- final: sort every slot, emit k keys per (part, query) ---------------------------------------
    const int part = s_idx * C::RP + rp;
    for (int jj = 0; jj < 32; ++jj) {
        const int qq = (qt * C::WQ + g) * 32 + jj;
        if (qq >= a.nq) break;
        const u32 n = cnt_w[jj];
        u64 key[C::NPL];
#pragma unroll
        for (int p = 0; p < C::NPL; ++p) {
            const u32 e = lane + 64 * p;
            key[p] = (e < n) ? cand_w[jj * C::CAP + e] : 0ull;
        }
        rmu_bitonic_sort_desc<C::NPL>(key, lane);
        u64* dst = a.partial + ((size_t)part * a.nq + qq) * a.k;
#pragma unroll
        for (int p = 0; p < C::NPL; ++p) {
            const int e = lane + 64 * p;
            if (e < a.k) dst[e] = key[p];
        }
    }
}

template <class C>
int launch_cfg(const ScanLaunch* p, hipStream_t s) {
    static bool attr_done = false;   // benign race: idempotent
    if (!attr_done) {
        if (hipFuncSetAttribute((const void*)scan_topk_kernel<C>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                C::LDS_BYTES) != hipSuccess)
            return RMU_E_HIP;
        attr_done = true;
    }
    hipLaunchKernelGGL(scan_topk_kernel<C>, dim3(p->grid), dim3(256), C::LDS_BYTES, s, *p);
    return hipGetLastError() == hipSuccess ? RMU_OK : RMU_E_HIP;
}

// geometry table: (WQ) x (k class).  kv 0: k <= 32 (CAP 64, one check per tile); kv 1: k <= 112.
//                                 D    WQ CKF RING CAP NCHECK
template <int D> using C_w4_k0 = Cfg<D, 4, 96, 4, 64, 1>;    // 48 KiB ring + 64 KiB candidates
template <int D> using C_w4_k1 = Cfg<D, 4, 96, 2, 128, 2>;   // 24 KiB ring + 128 KiB candidates
template <int D> using C_w2_k0 = Cfg<D, 2, 96, 3, 64, 1>;    // 72 KiB ring
template <int D> using C_w2_k1 = Cfg<D, 2, 48, 2, 128, 2>;   // 24 KiB ring
template <int D> using C_w1_k0 = Cfg<D, 1, 48, 3, 64, 1>;    // 72 KiB ring

template <int D>
int launch_d(const ScanLaunch* p, hipStream_t s) {
    switch (p->wq * 2 + p->kv) {
        case 8: return launch_cfg<C_w4_k0<D>>(p, s);
        case 9: return launch_cfg<C_w4_k1<D>>(p, s);
        case 4: return launch_cfg<C_w2_k0<D>>(p, s);
        case 5: return launch_cfg<C_w2_k1<D>>(p, s);
        case 2: return launch_cfg<C_w1_k0<D>>(p, s);
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

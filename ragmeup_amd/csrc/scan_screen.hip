// scan_screen.hip -- fp16 SCREENING scan + exact fp32 re-scoring (gfx950 / MI355X only).
//
// Same job as scan_topk.hip (the FLAT search behind server/RAGHelper.py:497-499) at a fraction of the MFMA time, and
// still exact: the screen only PROPOSES candidates; the returned ids/scores come from an fp32 re-score in the
// exact kernel's own summation order, guarded by a per-query sufficiency test; queries that fail it are answered
// by the exact scan.
//
// Screening image (built at add time, HALF the bytes of the fp32 matrix: 768 B per 384-d row): h = fp16(64 * x) in
// natural k order (the 2^6 scale keeps |x| >= 1e-6 out of the fp16 subnormal range; |x| must stay below ~1000).
// Queries are converted the same way per call.
//   4096 * s~ = sum h_x . h_q                    (ONE v_mfma_f32_32x32x16_f16 per 16 k, one fp32 accumulator)
// Error vs the exact kernel's fp32 score, with dx = x - h_x/64 (row), dq = q - h_q/64 (query), by Cauchy-Schwarz:
//   |sum (x q - h_x h_q)/4096| <= |dx| |q| + |x| |dq| + |dx| |dq|.
// |dx|max is MEASURED when rows are added (k_img_err; ~1.7e-4 |x| for real data, 4.9e-4 |x| worst case) and |dq| is
// measured per query in the re-score kernel, so subnormal flushes and odd value ranges are covered by construction.
// The fp32 accumulation of 384 exact fp16 products adds <= 408 * 2^-24 = 2.5e-5 |x||q| and the exact kernel is itself
// within 384 * 2^-24 = 2.3e-5 |x||q| of the true dot product:
//   EPS(q) = |dx|max |q| + |x|max |dq| + |dx|max |dq| + 5e-5 |x|max |q|          (~4e-4 for unit vectors)
// Sufficiency (per query): with the approximate top-K' (K' = 32) sorted, tau = k-th best s~.  If fewer than K'
// candidates exist, or s~[K'-1] < tau - 2*EPS, every row outside the candidate set has an exact score below k rows
// of the set, so the exact top-k is inside it.  Otherwise the query is flagged and re-run on the exact scan.
// (An earlier hi/lo split variant, 3 MFMAs per 16 k with EPS = 1e-4, ran at 25 ms for the 10M x 1024 headline; its
// ablations showed the LDS/L2 path, not the MFMA pipe, setting the time, which is what halving the bytes attacks.)
#include <cstdlib>
#include <mutex>
#include <type_traits>
#include "rmu_common.h"
#include "scan_common.h"
#include "../../include/rmu.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int SD = 384;                 // floats per row
constexpr int IMGB = RMU_IMG_ROW_BYTES; // 768: screening-image bytes per row
static_assert(IMGB == SD * 2, "image geometry");
constexpr int S_RT = 32;                // rows per tile (the 4 waves of a workgroup share it)
constexpr int S_CKB = IMGB / 2;         // 384 B per row per chunk: a tile is streamed as two half-k chunks
constexpr int S_U16 = S_CKB / 16;       // 24 16-byte units per row per chunk
constexpr int S_TS = SD / 16;           // 24 MFMA steps per tile ...
constexpr int S_CS = S_TS / 2;          // ... 12 per chunk
constexpr int S_SLOT = S_RT * S_CKB;    // 12 KiB ring slot
// G = 32-query groups per wave.  The v2 kernel (G = 1) was bound by LDS bandwidth: four waves each reading the whole tile
// need 4 x 1 KiB per 32-cycle MFMA = all 128 B/clk of the LDS before the DMA writes are counted (ablations in
// DESIGN.md 4.3).  With G = 2 every A fragment feeds two MFMAs (64 queries per wave, 256 per workgroup), halving LDS
// and L2 bytes per MFMA; the price is 80 KiB of candidate slots, hence the smaller ring slots and CAP.
// NW = waves per workgroup.  8 (round 3, G = 1): two waves per SIMD, 32 queries each, still 256 queries per workgroup -- a lone
// wave per SIMD issues roughly one instruction per 6-7 cycles in a wait / MFMA / ds_read / VALU mix (tools/ubench/mfma_issue.hip:
// the same skeleton runs 46 cycles per MFMA bare and 65 with five VALU fillers), so the 4-wave kernel's ~4.5 fillers per MFMA keep
// the matrix pipe under half busy; two interleaved instruction streams per SIMD hide each other's issue gaps.  The price is
// one A-fragment read per MFMA instead of one per two (128 of the LDS's 256 B/clk).
template <int G, int NRV = 0, int NW = 4>
struct ScreenCfg {
    static_assert(NW == 4 || (NW == 8 && G == 1), "8 waves carry one 32-query group each");
    // K' = 32 candidates per (chunk, query).  Appends reserve their position with ds_add_rtn; a position past the slot
    // is retried after the compaction (back to K' entries) that it triggers, and a slot is compacted early once an
    // append lands in its last A entries.
    static constexpr int QW = 32 * G;                       // queries per wave
    static constexpr bool BIG = G == 2 || NW == 8;          // 256 queries per workgroup: 80 KiB of candidate slots
    static constexpr int CAP = BIG ? 40 : 56, NPL = 1, A = 4;
    static constexpr int NR = NRV ? NRV : (BIG ? 6 : 8);      // ring slots (NRV: ring-depth experiments)
    static constexpr int NDW = NW == 8 ? 6 : 4;             // waves that issue the corpus DMA (12 one-KiB pieces per chunk) ...
    static constexpr int NIW = 12 / NDW;                    // ... and how many pieces each of them issues per chunk
    static constexpr int RING_BYTES = NR * S_SLOT;
    static constexpr int CAND_BYTES = NW * QW * CAP * 8;
    static constexpr int CNT_OFF = RING_BYTES + CAND_BYTES;
    static constexpr int THR_OFF = CNT_OFF + NW * QW * 4;
    static constexpr int TRASH_OFF = THR_OFF + NW * QW * 4;
    static constexpr int GT_OFF = TRASH_OFF + NW * 64 * 8;
    static constexpr int LDS_BYTES = GT_OFF + NW * 256;
};
static_assert(ScreenCfg<1>::LDS_BYTES <= 160 * 1024 && ScreenCfg<2>::LDS_BYTES <= 160 * 1024 && ScreenCfg<1, 0, 8>::LDS_BYTES <= 160 * 1024, "LDS");

extern __shared__ __attribute__((aligned(16))) char ssm[];

// fp32 rows [n, 384 of `stride` floats] -> screening image [n, 768 B] = fp16(scale * x); one thread per group of 8 k.  scale = 64, or 32 for the
// L2 index's queries, which are stored doubled (rmu_api.hip: k_l2_aug_queries)
// (round 6, second session) zero_a / zero_b: words the FIRST workgroup zeroes on the way -- the query conversion opens every screened search, and
// the ladder's shared thresholds and the re-run count used to be two memsets in front of it (~4.5 us of kernel boundary each)
__global__ void k_split_rows(const float* __restrict__ src, char* __restrict__ dst, int64_t n_groups, int stride, float scale,
                             u32* __restrict__ zero_a, int n_zero_a, u32* __restrict__ zero_b, int n_zero_b) {
    if (blockIdx.x == 0) {
        for (int i = threadIdx.x; i < n_zero_a; i += blockDim.x) zero_a[i] = 0u;
        for (int i = threadIdx.x; i < n_zero_b; i += blockDim.x) zero_b[i] = 0u;
    }
    const int64_t gidx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gidx >= n_groups) return;
    const float* s = src + (gidx / (SD / 8)) * stride + (gidx % (SD / 8)) * 8;
    f16x8 hi;
#pragma unroll
    for (int e = 0; e < 8; ++e) hi[e] = (_Float16)(s[e] * scale);
    *(f16x8*)(dst + gidx * 16) = hi;
}

// err2[r] = |x_r - h_r / 64|^2: the measured rounding error of the screening image, one wave per row
__global__ __launch_bounds__(256) void k_img_err(const float* __restrict__ x, int64_t n, float* __restrict__ err2, int stride) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n) return;
    const float* row = x + r * stride;
    float s = 0.f;
    for (int c = lane; c < SD; c += 64) {
        const float d = row[c] - (float)(_Float16)(row[c] * 64.0f) * (1.0f / 64.0f);
        s = fmaf(d, d, s);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) err2[r] = s * 1.0001f;   // summation slack
}

// EXP = timing ablations (wrong results): bit 0 no corpus DMA, bit 1 no LDS fragment reads, bit 3 no filter VALU; bit 2 = debug counters
// S_PRE = A-fragment prefetch depth in steps
template <int G, int EXP = 0, int S_PRE = 4, int NRV = 0, int NT = 0, int NW = 4>
__global__ __launch_bounds__(64 * NW) void scan_screen_kernel(const ScanLaunch a) {
    using C = ScreenCfg<G, NRV, NW>;
    constexpr bool DBG = (EXP & 4) != 0;   // cycle / event counters into a.dbg (RMU_SCAN_EXP=7)
    static_assert(S_CS % S_PRE == 0, "fragment register ring must close over a chunk");
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave = query groups, all waves read the same rows
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
    const char* img = (const char*)a.x + a.row0 * (int64_t)IMGB;      // rows [row0, row0 + n_rows) of the image

    char* ring = ssm;
    u64* cand_w = (u64*)(ssm + C::RING_BYTES) + (size_t)w * C::QW * C::CAP;
    u32* cnt_w = (u32*)(ssm + C::CNT_OFF) + w * C::QW;
    float* thr_w = (float*)(ssm + C::THR_OFF) + w * C::QW;
    const int q_base = (qt * NW + w) * C::QW;            // this wave's first query; group g, lane j owns q_base + 32 g + j
    bool q_ok[G];
#pragma unroll
    for (int g = 0; g < G; ++g) q_ok[g] = q_base + 32 * g + j < a.nq;
    if (lane < C::QW) {
        cnt_w[lane] = 0;
        thr_w[lane] = (q_base + lane < a.nq) ? -INFINITY : INFINITY;
    }
    float thr_loc[G], thr_g[G], thr_s[G];                // thr_s = max(local, shared) * 4096: the accumulator's scale
#pragma unroll
    for (int g = 0; g < G; ++g) {
        thr_loc[g] = q_ok[g] ? -INFINITY : INFINITY;
        thr_g[g] = -INFINITY;
        thr_s[g] = thr_loc[g];
    }
    u32* gthr_w = a.gthr + q_base;
    const u32* gt_lds = (const u32*)(ssm + C::GT_OFF) + w * 64;
    // ---- sibling pacing (G = 1, <= 4 query tiles, one workgroup per CU: rmu_screen_plan decides) ----------------------------
    // The nqt workgroups that scan the SAME row chunk for different query tiles sit on one XCD (block map above) so that the
    // chunk's image bytes come from HBM once and from that XCD's L2 nqt - 1 times -- which only works while the siblings stay
    // within an L2 window of each other, and left alone they drift (slow tiles, compactions): round 3 measured 1.97x the image
    // in HBM fetches, L2 hit 0.53 of an ideal 0.75.  The whole grid is resident at once (one workgroup per CU, no second wave
    // of blocks), so the kernel lasts as long as its slowest workgroup and a leader that waits loses nothing: every workgroup
    // publishes ~tile (0 = not started / finished = "ignore me") once per tile, reads its siblings' words one tile stale through
    // the same 4-byte LDS-DMA that refreshes the shared thresholds (lanes 32..35: no extra VMEM instruction), and the pacing wave
    // holds the workgroup at the ring barrier while it is more than a.pace tiles ahead of the slowest sibling.  A HINT only: the
    // wait is bounded (a sibling that is not resident -- another kernel on the device -- switches pacing off for this workgroup).
    // The pacing wave is the LAST one (with 8 waves it carries no corpus DMA): vmcnt retires in order, so a store issued by a DMA
    // wave sits in front of that wave's counted ring waits until the write is acknowledged -- the first version (wave 0, agent-scope
    // store) doubled the kernel's time.  The store is a plain one: the siblings share this XCD's L2, which is where it lands, and
    // they read with sc1 (past their vector L1).
    const bool pace_on = G == 1 && a.prog != nullptr;
    u32* prog_w = a.prog + (size_t)s_idx * 4;
    bool pace_live = pace_on;
    const u32* gsrc = gthr_w + (lane & (C::QW - 1));
    if (G == 1 && pace_on && lane >= 32 && lane < 36) gsrc = prog_w + (lane - 32);
    auto refresh_gthr = [&]() {   // 4-byte LDS-DMA of this wave's shared thresholds (G = 1: both lane halves load the same)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                         (__attribute__((address_space(3))) void*)(ssm + C::GT_OFF + w * 256), 4, 0, 16);
    };
    constexpr int PW = NW - 1;         // the pacing wave
    auto pace_step = [&](int tile) {   // once per tile, after the barrier (the DMA'd words may be a few tiles stale: harmless)
        if (lane == 0) __hip_atomic_store(prog_w + qt, ~(u32)tile, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const u32 m4 = max(max(gt_lds[32], gt_lds[33]), max(gt_lds[34], gt_lds[35]));   // uniform LDS reads (words of this wave's area)
        int lead = m4 ? tile - (int)~m4 : -1;                     // all zero: nobody to wait for
        if (__builtin_expect(__builtin_amdgcn_readfirstlane(lead) > a.pace, 0)) {
            int spins = 0;
            do {
                __builtin_amdgcn_s_sleep(24);
                u32 v = 0;
                if (lane < 4) v = __hip_atomic_load(prog_w + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                v = max(v, (u32)__shfl_xor((int)v, 1));
                v = max(v, (u32)__shfl_xor((int)v, 2));
                const u32 vm = (u32)__builtin_amdgcn_readfirstlane((int)v);
                lead = vm ? tile - (int)~vm : -1;
            } while (lead > a.pace && ++spins < 400);
            if (spins >= 400) pace_live = false;
        }
    };

    // ---- query fragments: step T covers k [16T, 16T+16); lane half h owns 8 of them -----------------------------------
    f16x8 qh[G][S_TS];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const char* qrow = (const char*)a.q + (size_t)(q_ok[g] ? q_base + 32 * g + j : 0) * IMGB + h * 16;
#pragma unroll
        for (int T = 0; T < S_TS; ++T) qh[g][T] = *(const f16x8*)(qrow + T * 32);
    }
    // The loads must be COMPLETE, as far as the compiler's wait-count pass can tell, before the first LDS-DMA is issued: it cannot count
    // through the ring's inline-asm waits, so a query fragment still "pending" at the loop head gets an s_waitcnt vmcnt(0) in front of its
    // first MFMA -- inside the loop, draining the DMA ring once per trip (found in round 4: both tile-loop kernels had it).
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int T = 0; T < S_TS; ++T) asm volatile("" : "+v"(qh[g][T]));

    // ---- DMA source map: LDS unit f -> row f/24, physical unit f%24 holds logical unit p ^ ((row >> 1) & 7).  LDS rows are
    // 384 B = 96 banks apart, so rows alternate between two bank halves; the XOR spreads 8 row pairs over the 8 units of an
    // aligned block: any 16 consecutive rows reading one logical unit touch 16 distinct 4-bank groups (conflict free).
    u32 dma_off[C::NIW];
#pragma unroll
    for (int n = 0; n < C::NIW; ++n) {
        const int f = (n * C::NDW + (w < C::NDW ? w : 0)) * 64 + lane;
        const int i = f / S_U16, p = f % S_U16;
        dma_off[n] = (u32)(i * IMGB + (p ^ ((i >> 1) & 7)) * 16);
    }
    const int nchunks = 2 * ntiles;
    // one of the LDS-DMA instructions of chunk cc (twelve 1-KiB pieces per chunk and workgroup).  They are issued one at a time between MFMA steps, not as a burst
    // behind the barrier: a wave that cannot hand its VMEM instruction to the (busy) address unit cannot issue MFMAs either,
    // and twelve back-to-back 1-KiB DMA instructions per chunk and workgroup cost ~700 cycles per tile that way.
    auto issue_part = [&](int cc, int n) {
        if (EXP & 1) return;   // ablation: no corpus DMA at all
        if (NW > C::NDW && w >= C::NDW) return;   // (uniform) 8 waves: six of them carry the twelve pieces
        int ce = cc;
        if (ce >= nchunks) ce = nchunks - 1;
        const char* sbase = img + ((t0 + (ce >> 1)) * S_RT) * (int64_t)IMGB + (ce & 1) * S_CKB;
        char* slot = ring + (cc % C::NR) * S_SLOT;
        if (NT)       // literal aux operands only (see scan_topk.hip)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(sbase + dma_off[n]),
                                             (__attribute__((address_space(3))) void*)(slot + (n * C::NDW + w) * 1024), 16, 0, 2);
        else
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(sbase + dma_off[n]),
                                             (__attribute__((address_space(3))) void*)(slot + (n * C::NDW + w) * 1024), 16, 0, 0);
    };
    auto issue_chunk = [&](int cc) {
#pragma unroll
        for (int n = 0; n < C::NIW; ++n) issue_part(cc, n);
    };
    // A fragment of (row j, chunk step t): logical unit 2t + h = 8 (t >> 2) + (2 (t & 3) + h); the XOR touches the low three
    // bits only, so four per-lane bases + an immediate (t >> 2) * 128 address a whole chunk
    int abase[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) abase[m] = j * S_CKB + (((2 * m + h) ^ ((j >> 1) & 7)) * 16);
    f16x8 fr[S_PRE];
#pragma unroll
    for (int m = 0; m < S_PRE; ++m) fr[m] = f16x8{};
    // Fragment reads and the waits on them are inline asm: left to itself hipcc sinks every ds_read to just before the MFMA
    // that consumes it and waits lgkmcnt(0) there (a 32-cycle MFMA cannot hide an LDS round trip).  The reads keep their
    // program order (volatile), S_PRE of them are in flight, and `frag_wait` ties the counted wait to the register the MFMA
    // reads, so the MFMA cannot be scheduled above it.
    const u32 ring_addr = lds_addr(ring);
    auto read_frag = [&](f16x8& dst, int slot_off, int t) {
        if (EXP & 2) {         // ablation: no LDS fragment reads (keep the register live and opaque)
            asm volatile("" : "+v"(dst));
            return;
        }
        const u32 addr = ring_addr + (u32)(abase[t & 3] + slot_off);
        if ((t >> 2) == 0) asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr));
        else if ((t >> 2) == 1) asm volatile("ds_read_b128 %0, %1 offset:128" : "=v"(dst) : "v"(addr));
        else asm volatile("ds_read_b128 %0, %1 offset:256" : "=v"(dst) : "v"(addr));
    };
    auto frag_wait = [&](f16x8& f) {
        if (EXP & 2) return;
        asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(f) : "n"(S_PRE - 1));
    };

    u32 cnt_addr[G], cand_addr[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        cnt_addr[g] = lds_addr(cnt_w + 32 * g + j);
        cand_addr[g] = lds_addr(cand_w + (32 * g + j) * C::CAP);
    }
    const u32 trash_addr = lds_addr(ssm + C::TRASH_OFF) + threadIdx.x * 8u;
    auto set_thr = [&]() {
#pragma unroll
        for (int g = 0; g < G; ++g) thr_s[g] = fmaxf(thr_loc[g], thr_g[g]) * 4096.0f;
    };
    auto check_compact = [&]() {
        const u32 c = cnt_w[lane & (C::QW - 1)];
        u64 mask = __ballot(c > (u32)(C::CAP - C::A));
        if (G == 1) mask = (u32)mask | (u32)(mask >> 32);
        if (mask) {
            while (mask) {
                const int jj = __builtin_ctzll(mask);
                mask &= mask - 1;
                compact_slot<C>(jj, cand_w, cnt_w, thr_w, a.k, lane, gthr_w);
            }
#pragma unroll
            for (int g = 0; g < G; ++g) thr_loc[g] = thr_w[32 * g + j];
            set_thr();
        }
    };
    constexpr int GRP = C::NIW;                            // corpus VMEM ops per chunk group (of a wave that issues any)
    constexpr int WAITN = GRP * (C::NR - 3);               // at a chunk's barrier only chunks >= cc + 2 may be in flight

    struct Acc { f32x16 a; };                          // 4096 * s~  (rows and queries are both scaled by 2^6)
    auto score = [](const Acc& p, int r) { return p.a[r] * (1.0f / 4096.0f); };
    u32 pmask[G], res_pos = 0;
    // Append the passing scores of one tile (bits of pmask) to this lane's query slots.  In a seeded launch an event is
    // almost always a single score in a single lane, so the accumulator slots are visited under a wave-uniform branch each
    // (one ballot per slot) and only slots with a passing lane pay for key + ds_add_rtn + store.  A position past the slot
    // means "full": the compaction this triggers frees room and the score is retried.
    u32 d_slow = 0, d_rounds = 0, d_comp = 0, d_app = 0;
    unsigned long long d_clk_slow = 0, d_clk_bar = 0, d_clk_vm = 0, d_clk_all = DBG ? clock64() : 0;
    auto slow_path = [&](const Acc* p, int64_t rbase, bool recompute) {
        unsigned long long c0 = 0;
        if (recompute) {      // the hot loop only kept a wave-wide "any lane passed" flag (scalar unit): per-lane bits are made here
#pragma unroll
            for (int g = 0; g < G; ++g) {
                pmask[g] = 0;
#pragma unroll
                for (int r = 0; r < 16; ++r) pmask[g] |= (p[g].a[r] > thr_s[g]) ? (1u << r) : 0u;
            }
        }
        if (DBG) {
            ++d_slow;
#pragma unroll
            for (int g = 0; g < G; ++g) d_app += __builtin_popcount(pmask[g]);
            c0 = clock64();
        }
        u32 todo[G];
#pragma unroll
        for (int g = 0; g < G; ++g) todo[g] = pmask[g];
        bool again;
        do {
            bool nearly_full = false;
#pragma unroll
            for (int g = 0; g < G; ++g) {
                // wave-uniform union of the lanes' bit masks: one ballot, then one v_readlane per lane with work (usually one)
                u32 uni = 0;
                for (u64 bl = __ballot(todo[g] != 0); bl; bl &= bl - 1)
                    uni |= (u32)__builtin_amdgcn_readlane((int)todo[g], __builtin_ctzll(bl));
                u32 left = 0;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if ((uni >> r) & 1u) {                              // scalar test: most slots are skipped
                        const bool has = (todo[g] >> r) & 1u;
                        if (DBG) ++d_rounds;
                        const u64 key = rmu_make_key(score(p[g], r) + 0.0f, (u32)(rbase + (r & 3) + 8 * (r >> 2)));
                        asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(res_pos) : "v"(cnt_addr[g]), "v"(has ? 1u : 0u) : "memory");
                        const bool fits = has && res_pos < (u32)C::CAP;
                        lds_store_b64_nofence(fits ? cand_addr[g] + res_pos * 8u : trash_addr, key);
                        nearly_full |= has && res_pos >= (u32)(C::CAP - C::A);
                        left |= (has && !fits) ? (1u << r) : 0u;
                    }
                }
                todo[g] = left;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (__any(nearly_full)) {                               // some slot is (nearly) full: compact, then retry what did not fit
                check_compact();
                if (DBG) ++d_comp;
            }
            again = false;
#pragma unroll
            for (int g = 0; g < G; ++g) again |= todo[g] != 0;
        } while (__any(again));
        if (DBG) d_clk_slow += clock64() - c0;
    };

    int cc = 0;
    // one tile = two chunks of 12 steps, G MFMAs per step into acc[g]; the previous tile's 16 scores per lane and group are
    // filtered in the first gaps (slot r behind step r+1), then ONE branch (see scan_topk.hip for why).  Fragments run
    // S_PRE steps ahead of the MFMAs that eat them, across the chunk boundary.
    auto tile_body = [&](Acc* acc, Acc* prev, int64_t prev_rbase) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[g].a[r] = 0.f;
        }
        u64 any_pass = 0;     // OR of the v_cmp lane masks: lives in an SGPR pair, costs one VALU op per score (the compare)
#pragma unroll
        for (int c = 0; c < 2; ++c, ++cc) {
            unsigned long long cb = 0;
            if (DBG) cb = clock64();
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAITN) : "memory");
            if (DBG) { const unsigned long long cv = clock64(); d_clk_vm += cv - cb; cb = cv; }      // own DMA pieces landed | everybody arrived
            __builtin_amdgcn_s_barrier();
            if (DBG) d_clk_bar += clock64() - cb;
            if (c == 0) {
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const u32 go = gt_lds[32 * g + j];
                    thr_g[g] = (go && (a.share_thr & 1)) ? rmu_ord2f(go - 1u) : -INFINITY;
                }
                set_thr();
                if (G == 1 && w == PW && pace_live) pace_step(cc >> 1);
            } else {
                refresh_gthr();
            }
            const int cur_off = (cc % C::NR) * S_SLOT, nxt_off = ((cc + 1) % C::NR) * S_SLOT;
#pragma unroll
            for (int t = 0; t < S_CS; ++t) {
                const int gs = c * S_CS + t;
                if (gs == 18 && !(a.share_thr & 2) && __builtin_expect(any_pass != 0, 0)) slow_path(prev, prev_rbase, true);
                frag_wait(fr[gs % S_PRE]);
#pragma unroll
                for (int g = 0; g < G; ++g)
                    acc[g].a = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[gs % S_PRE], qh[g][gs], acc[g].a, 0, 0, 0);
                if (gs >= 1 && gs <= 16 && !(EXP & 8)) {
#pragma unroll
                    for (int g = 0; g < G; ++g) any_pass |= __ballot(prev[g].a[gs - 1] > thr_s[g]);
                }
                if (t + S_PRE < S_CS) read_frag(fr[gs % S_PRE], cur_off, t + S_PRE);
                else read_frag(fr[gs % S_PRE], nxt_off, t + S_PRE - S_CS);
                if (t % 4 == 1 && t / 4 < C::NIW) issue_part(cc + C::NR - 1, t / 4);      // steps 1, 5 (, 9): the wave's DMA instructions of the chunk
            }
        }
    };

    constexpr bool PP = (EXP & 16) != 0;
    if constexpr (PP) {
        // ---- PING-PONG form (round 4, 8 waves): the two waves of a SIMD never do the same thing at the same time.  Round 3's loop lets both
        // interleave wait / MFMA / compare / ds_read / DMA step by step; the matrix pipe is per SIMD and an in-order wave cannot slip an MFMA
        // into a gap shorter than 32 cycles (MI355X_MICROARCH.md, "Two waves per SIMD"): 0.59 MFMA busy.  Here a wave alternates a LOAD
        // segment -- the 12 fragment reads of a chunk into 48 registers, the previous tile's 16 filter compares, its share of the ring's DMA --
        // with a COMPUTE segment of 12 back-to-back MFMAs, and waves w and w + 4 (one SIMD) run half a period apart: group X = waves 0-3
        // loads chunk c in phase 2c and computes it in phase 2c + 1, group Y = waves 4-7 one phase later.  One s_barrier per phase.
        //   ring: chunk c is read by X in phase 2c and by Y in phase 2c + 1; its slot is refilled with chunk c + 6 - 1 + ... = c + 5's
        //   successor: the pieces of chunk c + 5 go into the slot of chunk c - 1 (free since the barrier that ended phase 2c - 1), the X
        //   half (pieces 0-5) issued in X's load of chunk c, the Y half (6-11) in Y's; a wave's pieces of chunk k have landed when at most
        //   3 chunks' worth of its own pieces are still in flight: waited for at the end of EVERY load segment, i.e. one barrier before
        //   anybody reads chunk k.
        // MEASURED (debug counters, RMU_SCAN_EXP=7 + RMU_SCREEN_PP=1, largest range): a load segment = 12 ds_read_b128 + the wait = 481 cycles
        // with four waves of a CU loading at once -- 8 cycles per KiB and CU, i.e. the LDS delivers 128 B/clk per CU = 32 B/clk to each SIMD, and
        // ONE 1-KiB A fragment per 32-cycle MFMA is exactly that rate.  The interleaved form's 0.59 MFMA busy (1300 cycles per chunk for 768 of
        // MFMA and 96 KiB of fragment reads = 74 B/clk) is the same wall from the other side: with one LDS fragment per MFMA the matrix pipe
        // and the LDS return path both have to run at 100 % at the same time.  Ping-pong does not move it (8.3 ms vs 7.75): debug builds only.
        // What would: every fragment feeding two MFMAs (64 queries per wave = 192 registers of query fragments: one wave per SIMD, the
        // 4-wave form, which is issue-bound instead: 0.35).
        static_assert(NW == 8 && G == 1, "ping-pong: 8 waves x 32 queries");
        if (ntiles > 0) {
            const bool grpY = w >= 4;
            const int wi = w & 3;
            const bool heavy = wi < 2;                           // two pieces per chunk (pieces wi and 4 + wi of the group's six), else one
            const int p1 = (grpY ? 6 : 0) + wi, p2 = (grpY ? 6 : 0) + 4 + wi;
            u32 poff1, poff2;
            {
                const int f1 = p1 * 64 + lane, f2 = p2 * 64 + lane;
                const int i1 = f1 / S_U16, q1 = f1 % S_U16, i2 = f2 / S_U16, q2 = f2 % S_U16;
                poff1 = (u32)(i1 * IMGB + (q1 ^ ((i1 >> 1) & 7)) * 16);
                poff2 = (u32)(i2 * IMGB + (q2 ^ ((i2 >> 1) & 7)) * 16);
            }
            auto issue_pp = [&](int cc2) {
                int ce = cc2 < nchunks ? cc2 : nchunks - 1;
                const char* sbase = img + ((t0 + (ce >> 1)) * S_RT) * (int64_t)IMGB + (ce & 1) * S_CKB;
                char* slot = ring + (cc2 % C::NR) * S_SLOT;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(sbase + poff1),
                                                 (__attribute__((address_space(3))) void*)(slot + p1 * 1024), 16, 0, 0);
                if (heavy)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(sbase + poff2),
                                                     (__attribute__((address_space(3))) void*)(slot + p2 * 1024), 16, 0, 0);
            };
            // in flight when a load segment ends: this wave's pieces of the three youngest chunks (issued in its last three compute segments)
            auto wait_own = [&]() {
                if (heavy) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");
            };
            refresh_gthr();
#pragma unroll
            for (int c0 = 0; c0 < C::NR - 1; ++c0) issue_pp(c0);
            if (heavy) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // chunk 0
            __builtin_amdgcn_s_barrier();
            if (grpY) __builtin_amdgcn_s_barrier();              // Y runs one phase behind X
            f16x8 fq[S_CS];
            // two accumulation chains per tile (even / odd k steps: a 32-cycle MFMA never waits for the one before it), summed -- and
            // filtered -- in the load segment that follows the tile
            Acc acc[1], acc2[1];
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[0].a[r] = -INFINITY; acc2[0].a[r] = 0.f; }
            const int64_t lane_r0 = a.row0 + t0 * S_RT + 4 * h;
            for (int tl = 0; tl < ntiles; ++tl) {
#pragma unroll
              for (int c = 0; c < 2; ++c) {                          // (unrolled: the chunk parity selects the query fragments at compile time)
                const int cc2 = 2 * tl + c;
                const int soff = (cc2 % C::NR) * S_SLOT;
                unsigned long long ck0 = 0, ck1 = 0, ck2 = 0, ck3 = 0;
                if (DBG) ck0 = clock64();
                // ---- LOAD segment: the chunk's 12 fragments; behind a tile's second chunk also that tile's sum, its 16 compares, ONE branch
#pragma unroll
                for (int t = 0; t < S_CS; ++t) read_frag(fq[t], soff, t);
                if (c == 0) {
                    if (tl > 0) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[0].a[r] += acc2[0].a[r];
                        if (!(a.share_thr & 2) && !(EXP & 8)) {
                            u64 any_pass = 0;
#pragma unroll
                            for (int r = 0; r < 16; ++r) any_pass |= __ballot(acc[0].a[r] > thr_s[0]);
                            if (__builtin_expect(any_pass != 0, 0)) slow_path(acc, lane_r0 + (int64_t)(tl - 1) * S_RT, true);
                        }
                    }
                    const u32 go = gt_lds[j];
                    thr_g[0] = (go && (a.share_thr & 1)) ? rmu_ord2f(go - 1u) : -INFINITY;
                    set_thr();
                    if (w == NW - 1 && pace_live) pace_step(tl);
                }
                wait_own();                                          // (the 12 fragment reads have landed as well)
#pragma unroll
                for (int t = 0; t < S_CS; ++t) asm volatile("" : "+v"(fq[t]));
                if (DBG) ck1 = clock64();
                __builtin_amdgcn_s_barrier();
                if (DBG) ck2 = clock64();
                // ---- COMPUTE segment: 12 MFMAs on two chains; this wave's DMA pieces of chunk cc2 + 5 and the threshold refresh in their shadow
                if (c == 0) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) { acc[0].a[r] = 0.f; acc2[0].a[r] = 0.f; }
                }
#pragma unroll
                for (int t = 0; t < S_CS; ++t) {
                    if (t & 1) acc2[0].a = __builtin_amdgcn_mfma_f32_32x32x16_f16(fq[t], qh[0][c * S_CS + t], acc2[0].a, 0, 0, 0);
                    else acc[0].a = __builtin_amdgcn_mfma_f32_32x32x16_f16(fq[t], qh[0][c * S_CS + t], acc[0].a, 0, 0, 0);
                    if (t == 1 || (t == 6 && heavy)) {
                        int ce = cc2 + C::NR - 1;
                        const int cs = ce % C::NR;
                        if (ce >= nchunks) ce = nchunks - 1;
                        const char* sbase = img + ((t0 + (ce >> 1)) * S_RT) * (int64_t)IMGB + (ce & 1) * S_CKB;
                        if (t == 1)
                            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(sbase + poff1),
                                                             (__attribute__((address_space(3))) void*)(ring + cs * S_SLOT + p1 * 1024), 16, 0, 0);
                        else
                            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(sbase + poff2),
                                                             (__attribute__((address_space(3))) void*)(ring + cs * S_SLOT + p2 * 1024), 16, 0, 0);
                    }
                    if (c == 1 && t == 9) refresh_gthr();
                }
                if (DBG) { asm volatile("" : "+v"(acc[0].a), "+v"(acc2[0].a)); ck3 = clock64(); }
                if (!(grpY && cc2 + 1 == nchunks)) __builtin_amdgcn_s_barrier();
                if (DBG) { d_clk_vm += ck1 - ck0; d_clk_bar += ck2 - ck1; d_clk_slow += ck3 - ck2; d_rounds += (u32)((clock64() - ck3) >> 4); }
              }
            }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            {
                const int64_t rbl = lane_r0 + (int64_t)(ntiles - 1) * S_RT;
                const int64_t row_end = a.row0 + a.n_rows;
                pmask[0] = 0;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    acc[0].a[r] += acc2[0].a[r];
                    const bool in_range = rbl + (r & 3) + 8 * (r >> 2) < row_end;
                    pmask[0] |= (in_range && acc[0].a[r] > thr_s[0]) ? (1u << r) : 0u;
                }
                if (__any(pmask[0] != 0)) slow_path(acc, rbl, false);
            }
        }
    } else
    if (ntiles > 0) {
        refresh_gthr();                                    // oldest VMEM op: seeded / already published thresholds
#pragma unroll
        for (int c0 = 0; c0 < C::NR - 1; ++c0) issue_chunk(c0);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GRP * (C::NR - 2)) : "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int m = 0; m < S_PRE; ++m) read_frag(fr[m], 0, m);
        Acc accA[G], accB[G];
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int r = 0; r < 16; ++r) accB[g].a[r] = -INFINITY;
        const int64_t lane_r0 = a.row0 + t0 * S_RT + 4 * h;
        auto rb = [&](int t) { return lane_r0 + (int64_t)t * S_RT; };
        const bool last_in_a = ((ntiles - 1) & 1) == 0;   // even tiles accumulate in A
        for (int tl = 0; tl < ntiles; tl += 2) {          // two copies of the body: accumulator parity
            tile_body(accA, accB, rb(tl - 1));              // (tile -1 = the -inf accumulators: nothing passes)
            if (tl + 1 < ntiles) tile_body(accB, accA, rb(tl));
        }
        if (pace_on && w == PW && lane == 0) __hip_atomic_store(prog_w + qt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // finished: ignore me
        // the fragment reads issued for a chunk that does not exist are still in flight: their registers must stay
        // allocated until the data has landed (the compiler sees dead values and would reuse the registers under them)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int m = 0; m < S_PRE; ++m) asm volatile("" : "+v"(fr[m]));
        {
            Acc last[G];
            const int64_t rbl = rb(ntiles - 1);
            const int64_t row_end = a.row0 + a.n_rows;
            bool any_pass = false;
#pragma unroll
            for (int g = 0; g < G; ++g) {
                pmask[g] = 0;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    last[g].a[r] = last_in_a ? accA[g].a[r] : accB[g].a[r];
                    const bool in_range = rbl + (r & 3) + 8 * (r >> 2) < row_end;
                    pmask[g] |= (in_range && last[g].a[r] > thr_s[g]) ? (1u << r) : 0u;
                }
                any_pass |= pmask[g] != 0;
            }
            if (__any(any_pass)) slow_path(last, rbl, false);
        }
    }
    if (DBG) {
        u32 app = d_app;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) app += __shfl_xor(app, o);
        if (lane == 0) {
            atomicAdd((unsigned long long*)a.dbg + 0, (unsigned long long)d_slow);
            atomicAdd((unsigned long long*)a.dbg + 1, (unsigned long long)d_comp);
            atomicAdd((unsigned long long*)a.dbg + 2, (unsigned long long)app);
            atomicAdd((unsigned long long*)a.dbg + 3, (unsigned long long)ntiles);
            atomicAdd((unsigned long long*)a.dbg + 4, (unsigned long long)d_rounds);
            atomicAdd((unsigned long long*)a.dbg + 5, d_clk_slow);
            atomicAdd((unsigned long long*)a.dbg + 6, d_clk_bar);
            atomicAdd((unsigned long long*)a.dbg + 8, d_clk_vm);
            atomicAdd((unsigned long long*)a.dbg + 7, (unsigned long long)(clock64() - d_clk_all));
        }
    }
    // ---- emit: best K' approximate candidates of this (chunk, query), sorted ---------------------------------------------
    const int part = s_idx;
    for (int jj = 0; jj < C::QW; ++jj) {
        const int qq = q_base + jj;
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

// ---- the product kernel for full query tiles (8 waves x 32 queries), on an instruction diet (round 4) ------------------------------------
// Same algorithm, layout, ring and filter semantics as scan_screen_kernel<1, .., NW = 8>; what changes is how many instructions a tile costs.
// The counters of this round's experiments fit "32 cycles per MFMA + 6-11 per OTHER instruction of the wave" for every form tried, and
// the round-3 kernel spends ~300 other instructions on 24 MFMAs -- half of them scalar arithmetic on the ring slot (cc % 6 by multiply-high,
// 64-bit source pointers per DMA piece, clamps) and one v_add per fragment read.  Here the tile loop is unrolled over the ring's period
// (6 chunks = 3 tiles, x 2 for the accumulator parity: six bodies) so that every ring slot is a compile-time constant:
//   fragment reads   ds_read_b128 with the slot folded into the 16-bit offset field: no address arithmetic at all
//   DMA pieces       LDS target = per-wave base + constant; source = ONE running 64-bit tile pointer (2 SALU per tile) + a per-lane 32-bit
//                    offset that already contains the piece's look-ahead distance; no clamp -- the ring's look-ahead past the last tile
//                    reads the image's slack rows (rmu_api.hip: kSlackRows, zero-filled) or the next chunk's rows, and is never consumed
//   filter           a running v_max3 over the previous tile's 16 scores (8 VALU) and ONE compare per tile instead of 16 v_cmp + 16 s_or
struct LeanCfg : ScreenCfg<1, 0, 8> {};

template <int EXP = 0>
__global__ __launch_bounds__(512) void scan_screen_lean_kernel(const ScanLaunch a) {
    using C = LeanCfg;
    constexpr bool DBG = (EXP & 4) != 0;
    constexpr int NW = 8, S_PRE = 4;
    static_assert(C::NR == 6 && C::NDW == 6 && C::NIW == 2, "the unrolled ring below is written for 6 slots, 6 DMA waves x 2 pieces");
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
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
    const char* img = (const char*)a.x + a.row0 * (int64_t)IMGB;
    char* ring = ssm;
    u64* cand_w = (u64*)(ssm + C::RING_BYTES) + (size_t)w * C::QW * C::CAP;
    u32* cnt_w = (u32*)(ssm + C::CNT_OFF) + w * C::QW;
    float* thr_w = (float*)(ssm + C::THR_OFF) + w * C::QW;
    const int q_base = (qt * NW + w) * C::QW;
    const bool q_ok = q_base + j < a.nq;
    if (lane < C::QW) {
        cnt_w[lane] = 0;
        thr_w[lane] = (q_base + lane < a.nq) ? -INFINITY : INFINITY;
    }
    float thr_loc = q_ok ? -INFINITY : INFINITY, thr_g = -INFINITY, thr_s = thr_loc;
    u32* gthr_w = a.gthr + q_base;
    const u32* gt_lds = (const u32*)(ssm + C::GT_OFF) + w * 64;
    const bool pace_on = a.prog != nullptr;               // sibling pacing: see scan_screen_kernel
    u32* prog_w = a.prog + (size_t)s_idx * 4;
    bool pace_live = pace_on;
    const u32* gsrc = gthr_w + (lane & (C::QW - 1));
    if (pace_on && lane >= 32 && lane < 36) gsrc = prog_w + (lane - 32);
    auto refresh_gthr = [&]() {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                         (__attribute__((address_space(3))) void*)(ssm + C::GT_OFF + w * 256), 4, 0, 16);
    };
    constexpr int PW = NW - 1;
    auto pace_step = [&](int tile) {
        if (lane == 0) __hip_atomic_store(prog_w + qt, ~(u32)tile, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const u32 m4 = max(max(gt_lds[32], gt_lds[33]), max(gt_lds[34], gt_lds[35]));
        int lead = m4 ? tile - (int)~m4 : -1;
        if (__builtin_expect(__builtin_amdgcn_readfirstlane(lead) > a.pace, 0)) {
            int spins = 0;
            do {
                __builtin_amdgcn_s_sleep(24);
                u32 v = 0;
                if (lane < 4) v = __hip_atomic_load(prog_w + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                v = max(v, (u32)__shfl_xor((int)v, 1));
                v = max(v, (u32)__shfl_xor((int)v, 2));
                const u32 vm = (u32)__builtin_amdgcn_readfirstlane((int)v);
                lead = vm ? tile - (int)~vm : -1;
            } while (lead > a.pace && ++spins < 400);
            if (spins >= 400) pace_live = false;
        }
    };
    f16x8 qh[S_TS];
    {
        const char* qrow = (const char*)a.q + (size_t)(q_ok ? q_base + j : 0) * IMGB + h * 16;
#pragma unroll
        for (int T = 0; T < S_TS; ++T) qh[T] = *(const f16x8*)(qrow + T * 32);
#pragma unroll
        for (int T = 0; T < S_TS; ++T) asm volatile("" : "+v"(qh[T]));     // complete before any LDS-DMA (see scan_screen_kernel)
    }
    // DMA: wave w < 6 carries pieces w and 6 + w of every chunk.  A piece issued during chunk (tile t, half c) belongs to chunk 2t + c + 5:
    // half c' = 1 - c of tile t + 2 + c, so its per-lane source offset from the CURRENT tile's base is a constant of (c, n)
    u32 dma_off[2][C::NIW];
#pragma unroll
    for (int n = 0; n < C::NIW; ++n) {
        const int f = (n * C::NDW + (w < C::NDW ? w : 0)) * 64 + lane;
        const int i = f / S_U16, p = f % S_U16;
        const u32 o = (u32)(i * IMGB + (p ^ ((i >> 1) & 7)) * 16);
        dma_off[0][n] = o + 2u * S_RT * IMGB + S_CKB;      // issued in a first half: second half of tile t + 2
        dma_off[1][n] = o + 3u * S_RT * IMGB;              // issued in a second half: first half of tile t + 3
    }
    char* const lds_w = ring + (w < C::NDW ? w : 0) * 1024;                 // this wave's piece inside a slot (+ 6 KiB for its second piece)
    const char* tp = img + (t0 * S_RT) * (int64_t)IMGB;                     // the current tile's rows (uniform)
    // SLOT = ring slot the piece lands in (compile time)
    auto issue_part = [&](auto SLOT, int c, int n) {
        if (EXP & 1) return;
        if (w >= C::NDW) return;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(tp + dma_off[c][n]),
                                         (__attribute__((address_space(3))) void*)(lds_w + decltype(SLOT)::value * S_SLOT + n * C::NDW * 1024), 16, 0, 0);
    };
    u32 ab[4];                                            // fragment (row j, step t) of slot s: ab[t & 3] + s * S_SLOT + (t >> 2) * 128
#pragma unroll
    for (int m = 0; m < 4; ++m) ab[m] = lds_addr(ring) + (u32)(j * S_CKB + (((2 * m + h) ^ ((j >> 1) & 7)) * 16));
    f16x8 fr[S_PRE];
#pragma unroll
    for (int m = 0; m < S_PRE; ++m) fr[m] = f16x8{};
    auto read_frag = [&](f16x8& dst, auto OFF, int t) {
        if (EXP & 2) { asm volatile("" : "+v"(dst)); return; }
        const u32 ad = ab[t & 3];
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(ad), "n"(decltype(OFF)::value));
    };
    auto frag_wait = [&](f16x8& f) {
        if (EXP & 2) return;
        asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(f) : "n"(S_PRE - 1));
    };
    const u32 cnt_addr = lds_addr(cnt_w + j), cand_addr = lds_addr(cand_w + j * C::CAP);
    const u32 trash_addr = lds_addr(ssm + C::TRASH_OFF) + threadIdx.x * 8u;
    auto set_thr = [&]() { thr_s = fmaxf(thr_loc, thr_g) * 4096.0f; };
    auto check_compact = [&]() {
        const u32 c = cnt_w[lane & (C::QW - 1)];
        u64 mask = __ballot(c > (u32)(C::CAP - C::A));
        mask = (u32)mask | (u32)(mask >> 32);
        if (mask) {
            while (mask) {
                const int jj = __builtin_ctzll(mask);
                mask &= mask - 1;
                compact_slot<C>(jj, cand_w, cnt_w, thr_w, a.k, lane, gthr_w);
            }
            thr_loc = thr_w[j];
            set_thr();
        }
    };
    constexpr int GRP = C::NIW;
    constexpr int WAITN = GRP * (C::NR - 3);
    u32 res_pos = 0;
    u32 d_slow = 0, d_comp = 0, d_app = 0;
    unsigned long long d_clk_slow = 0, d_clk_bar = 0, d_clk_vm = 0, d_clk_all = DBG ? clock64() : 0;
    // append the scores of one tile that pass (p = 4096 * s~; bits of `inmask` = accumulator slots inside the row range): as scan_screen_kernel
    auto slow_path = [&](const f32x16& p, int64_t rbase, u32 inmask) {
        unsigned long long c0 = 0;
        u32 todo = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) todo |= (p[r] > thr_s) ? (1u << r) : 0u;
        todo &= inmask;
        if (DBG) { ++d_slow; d_app += __builtin_popcount(todo); c0 = clock64(); }
        bool again;
        do {
            bool nearly_full = false;
            u32 uni = 0;
            for (u64 bl = __ballot(todo != 0); bl; bl &= bl - 1) uni |= (u32)__builtin_amdgcn_readlane((int)todo, __builtin_ctzll(bl));
            u32 left = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if ((uni >> r) & 1u) {
                    const bool has = (todo >> r) & 1u;
                    const u64 key = rmu_make_key(p[r] * (1.0f / 4096.0f) + 0.0f, (u32)(rbase + (r & 3) + 8 * (r >> 2)));
                    asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(res_pos) : "v"(cnt_addr), "v"(has ? 1u : 0u) : "memory");
                    const bool fits = has && res_pos < (u32)C::CAP;
                    lds_store_b64_nofence(fits ? cand_addr + res_pos * 8u : trash_addr, key);
                    nearly_full |= has && res_pos >= (u32)(C::CAP - C::A);
                    left |= (has && !fits) ? (1u << r) : 0u;
                }
            }
            todo = left;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (__any(nearly_full)) {
                check_compact();
                if (DBG) ++d_comp;
            }
            again = todo != 0;
        } while (__any(again));
        if (DBG) d_clk_slow += clock64() - c0;
    };
    if (ntiles > 0) {
        refresh_gthr();
        {   // chunks 0..4 -> slots 0..4: the prologue's pieces sit at 0, 1 | 2 tile offsets that the per-lane constants do not cover
            const char* tp0 = tp - (2 * S_RT * IMGB + S_CKB);          // so that dma_off[0] addresses (tile 0, half 0)
            auto pro = [&](auto SLOT, const char* base, int c) {
#pragma unroll
                for (int n = 0; n < C::NIW; ++n) {
                    if (w < C::NDW)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + dma_off[c][n]),
                                                         (__attribute__((address_space(3))) void*)(lds_w + decltype(SLOT)::value * S_SLOT + n * C::NDW * 1024), 16, 0, 0);
                }
            };
            pro(std::integral_constant<int, 0>{}, tp0, 0);                                     // (tile 0, half 0)
            pro(std::integral_constant<int, 1>{}, tp0 + S_CKB, 0);                             // (0, 1)
            pro(std::integral_constant<int, 2>{}, tp0 + S_RT * IMGB, 0);                       // (1, 0)
            pro(std::integral_constant<int, 3>{}, tp0 + S_RT * IMGB + S_CKB, 0);               // (1, 1)
            pro(std::integral_constant<int, 4>{}, tp0 + 2 * S_RT * IMGB, 0);                   // (2, 0)
        }
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GRP * (C::NR - 2)) : "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int m = 0; m < S_PRE; ++m) {
            if (!(EXP & 2)) asm volatile("ds_read_b128 %0, %1" : "=v"(fr[m]) : "v"(ab[m]));
        }
        f32x16 accA, accB;
#pragma unroll
        for (int r = 0; r < 16; ++r) { accB[r] = -INFINITY; accA[r] = 0.f; }
        const int64_t lane_r0 = a.row0 + t0 * S_RT + 4 * h;
        // one tile; P = its position in the ring's three-tile period (compile time): its chunks sit in slots 2P and 2P + 1
        auto tile_body = [&](auto PI, f32x16& acc, const f32x16& prev, int tl) {
            constexpr int P = decltype(PI)::value;
            float mx = -INFINITY;
            auto half = [&](auto CI) {
                constexpr int c = decltype(CI)::value;
                unsigned long long cb = 0;
                if (DBG) cb = clock64();
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAITN) : "memory");
                if (DBG) { const unsigned long long cv = clock64(); d_clk_vm += cv - cb; cb = cv; }
                __builtin_amdgcn_s_barrier();
                if (DBG) d_clk_bar += clock64() - cb;
                if (c == 0) {
                    const u32 go = gt_lds[j];
                    thr_g = (go && (a.share_thr & 1)) ? rmu_ord2f(go - 1u) : -INFINITY;
                    set_thr();
                    if (w == PW && pace_live) pace_step(tl);
                } else {
                    refresh_gthr();
                }
                auto step = [&](auto TI) {
                    constexpr int t = decltype(TI)::value % S_CS, cch = decltype(TI)::value / S_CS, gs = decltype(TI)::value;
                    constexpr int cur = 2 * P + cch, nxt = (cur + 1) % C::NR, tgt = (cur + C::NR - 1) % C::NR;
                    if (gs == 18 && !(a.share_thr & 2) && !(EXP & 8) && __builtin_expect(__ballot(mx > thr_s) != 0, 0))
                        slow_path(prev, lane_r0 + (int64_t)(tl - 1) * S_RT, 0xffffu);
                    frag_wait(fr[gs % S_PRE]);
                    if (gs == 0) {
                        const f32x16 z = {};
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[gs % S_PRE], qh[gs], z, 0, 0, 0);
                    } else {
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[gs % S_PRE], qh[gs], acc, 0, 0, 0);
                    }
                    // (asm: fmaxf canonicalises both inputs first -- three instructions per pair; a NaN score loses v_max3 as it fails a compare)
                    if (gs >= 1 && gs <= 8 && !(EXP & 8)) asm("v_max3_f32 %0, %0, %1, %2" : "+v"(mx) : "v"(prev[2 * gs - 2]), "v"(prev[2 * gs - 1]));
                    if (t + S_PRE < S_CS) read_frag(fr[gs % S_PRE], std::integral_constant<int, cur * S_SLOT + ((t + S_PRE) >> 2) * 128>{}, t + S_PRE);
                    else read_frag(fr[gs % S_PRE], std::integral_constant<int, nxt * S_SLOT + ((t + S_PRE - S_CS) >> 2) * 128>{}, t + S_PRE - S_CS);
                    if (t % 4 == 1 && t / 4 < C::NIW) issue_part(std::integral_constant<int, tgt>{}, cch, t / 4);
                    __builtin_amdgcn_sched_barrier(0);
                };
                if (c == 0) {
                    step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{}); step(std::integral_constant<int, 2>{});
                    step(std::integral_constant<int, 3>{}); step(std::integral_constant<int, 4>{}); step(std::integral_constant<int, 5>{});
                    step(std::integral_constant<int, 6>{}); step(std::integral_constant<int, 7>{}); step(std::integral_constant<int, 8>{});
                    step(std::integral_constant<int, 9>{}); step(std::integral_constant<int, 10>{}); step(std::integral_constant<int, 11>{});
                } else {
                    step(std::integral_constant<int, 12>{}); step(std::integral_constant<int, 13>{}); step(std::integral_constant<int, 14>{});
                    step(std::integral_constant<int, 15>{}); step(std::integral_constant<int, 16>{}); step(std::integral_constant<int, 17>{});
                    step(std::integral_constant<int, 18>{}); step(std::integral_constant<int, 19>{}); step(std::integral_constant<int, 20>{});
                    step(std::integral_constant<int, 21>{}); step(std::integral_constant<int, 22>{}); step(std::integral_constant<int, 23>{});
                }
            };
            half(std::integral_constant<int, 0>{});
            half(std::integral_constant<int, 1>{});
            tp += S_RT * IMGB;
        };
        using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
        for (int tl = 0; tl < ntiles; tl += 6) {           // ring period (3 tiles) x accumulator parity (2): six bodies
            tile_body(I0{}, accA, accB, tl);                 // (tile -1 = the -inf accumulators: nothing passes)
            if (tl + 1 < ntiles) tile_body(I1{}, accB, accA, tl + 1);
            if (tl + 2 < ntiles) tile_body(I2{}, accA, accB, tl + 2);
            if (tl + 3 < ntiles) tile_body(I0{}, accB, accA, tl + 3);
            if (tl + 4 < ntiles) tile_body(I1{}, accA, accB, tl + 4);
            if (tl + 5 < ntiles) tile_body(I2{}, accB, accA, tl + 5);
        }
        if (pace_on && w == PW && lane == 0) __hip_atomic_store(prog_w + qt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int m = 0; m < S_PRE; ++m) asm volatile("" : "+v"(fr[m]));
        {
            const bool last_in_a = ((ntiles - 1) & 1) == 0;
            f32x16 last;
            const int64_t rbl = lane_r0 + (int64_t)(ntiles - 1) * S_RT, row_end = a.row0 + a.n_rows;
            u32 inmask = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                last[r] = last_in_a ? accA[r] : accB[r];
                inmask |= (rbl + (r & 3) + 8 * (r >> 2) < row_end) ? (1u << r) : 0u;
            }
            if (!(a.share_thr & 2)) slow_path(last, rbl, inmask);
        }
    }
    if (DBG) {
        u32 app = d_app;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) app += __shfl_xor(app, o);
        if (lane == 0) {
            atomicAdd((unsigned long long*)a.dbg + 0, (unsigned long long)d_slow);
            atomicAdd((unsigned long long*)a.dbg + 1, (unsigned long long)d_comp);
            atomicAdd((unsigned long long*)a.dbg + 2, (unsigned long long)app);
            atomicAdd((unsigned long long*)a.dbg + 3, (unsigned long long)ntiles);
            atomicAdd((unsigned long long*)a.dbg + 5, d_clk_slow);
            atomicAdd((unsigned long long*)a.dbg + 6, d_clk_bar);
            atomicAdd((unsigned long long*)a.dbg + 8, d_clk_vm);
            atomicAdd((unsigned long long*)a.dbg + 7, (unsigned long long)(clock64() - d_clk_all));
        }
    }
    const int part = s_idx;
    for (int jj = 0; jj < C::QW; ++jj) {
        const int qq = q_base + jj;
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

// ---- lean form, ONE barrier per tile (round 4) ----------------------------------------------------------------------------------------------
// Debug counters of scan_screen_lean_kernel: 434 of a tile's ~2 400 cycles are spent waiting at its two ring barriers (six DMA waves, a pacing
// wave and an idle one arrive at different times, twice per tile).  Half-k ring slots need a barrier per half because six slots hold only
// three tiles; the 80 KiB of LDS candidate slots are what keeps the ring that small.  Here the candidates live in global memory as in the
// K-split kernel (slot counts in registers, appends are fire-and-forget stores), the ring holds FOUR tiles (8 slots, 96 KiB) and is handed
// over once per tile: at the barrier of tile t, tile t + 1 has landed (the fragment prefetch crosses into it), t + 2 is in flight, t + 3 is
// issued during the tile into the slots of t - 1.  Four bodies (ring period 4 tiles, accumulator parity 2).
struct Lean2Cfg {
    static constexpr int NW = 8, QW = 32, NR = 8, NDW = 6, NIW = 4;
    static constexpr int CAP = RMU_KS_CAP;
    static constexpr int RING_BYTES = NR * S_SLOT;
    static constexpr int GT_OFF = RING_BYTES;
    static constexpr int LDS_BYTES = GT_OFF + NW * 256;
};

template <int EXP = 0>
__global__ __launch_bounds__(512) void scan_screen_lean2_kernel(const ScanLaunch a) {
    using C = Lean2Cfg;
    constexpr bool DBG = (EXP & 4) != 0;
    constexpr int NW = 8, S_PRE = 4;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
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
    const char* img = (const char*)a.x + a.row0 * (int64_t)IMGB;
    char* ring = ssm;
    const int q_base = (qt * NW + w) * C::QW;
    const bool q_ok = q_base + j < a.nq;
    float thr_s = q_ok ? -INFINITY : INFINITY;            // 4096 * max(own k-th best, shared threshold): only ever rises
    u32 cnt = 0;                                          // entries in this lane's query slot (equal in lanes j and j + 32)
    u64* const gslot = a.gcand + ((size_t)s_idx * a.nq + (q_ok ? q_base + j : 0)) * C::CAP;
    u32* gthr_w = a.gthr + q_base;
    const u32* gt_lds = (const u32*)(ssm + C::GT_OFF) + w * 64;
    const bool pace_on = a.prog != nullptr;               // sibling pacing: see scan_screen_kernel
    u32* prog_w = a.prog + (size_t)s_idx * 4;
    bool pace_live = pace_on;
    const u32* gsrc = gthr_w + j;
    if (pace_on && lane >= 32 && lane < 36) gsrc = prog_w + (lane - 32);
    auto refresh_gthr = [&]() {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                         (__attribute__((address_space(3))) void*)(ssm + C::GT_OFF + w * 256), 4, 0, 16);
    };
    constexpr int PW = NW - 1;
    auto pace_step = [&](int tile) {
        if (lane == 0) __hip_atomic_store(prog_w + qt, ~(u32)tile, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const u32 m4 = max(max(gt_lds[32], gt_lds[33]), max(gt_lds[34], gt_lds[35]));
        int lead = m4 ? tile - (int)~m4 : -1;
        if (__builtin_expect(__builtin_amdgcn_readfirstlane(lead) > a.pace, 0)) {
            int spins = 0;
            do {
                __builtin_amdgcn_s_sleep(24);
                u32 v = 0;
                if (lane < 4) v = __hip_atomic_load(prog_w + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                v = max(v, (u32)__shfl_xor((int)v, 1));
                v = max(v, (u32)__shfl_xor((int)v, 2));
                const u32 vm = (u32)__builtin_amdgcn_readfirstlane((int)v);
                lead = vm ? tile - (int)~vm : -1;
            } while (lead > a.pace && ++spins < 400);
            if (spins >= 400) pace_live = false;
        }
    };
    f16x8 qh[S_TS];
    {
        const char* qrow = (const char*)a.q + (size_t)(q_ok ? q_base + j : 0) * IMGB + h * 16;
#pragma unroll
        for (int T = 0; T < S_TS; ++T) qh[T] = *(const f16x8*)(qrow + T * 32);
#pragma unroll
        for (int T = 0; T < S_TS; ++T) asm volatile("" : "+v"(qh[T]));     // complete before any LDS-DMA (see scan_screen_kernel)
    }
    // DMA: wave w < 6 carries pieces n * 6 + w (n = 0..3) of a tile's 24 = 12 * half + piece; issued during tile t they belong to tile t + 3
    u32 dma_off[C::NIW];
    int dma_dst[C::NIW];
#pragma unroll
    for (int n = 0; n < C::NIW; ++n) {
        const int id = n * C::NDW + (w < C::NDW ? w : 0);
        const int half = id / 12, pid = id % 12;
        const int f = pid * 64 + lane;
        const int i = f / S_U16, p = f % S_U16;
        dma_off[n] = (u32)(i * IMGB + (p ^ ((i >> 1) & 7)) * 16 + half * S_CKB) + 3u * S_RT * IMGB;
        dma_dst[n] = half * S_SLOT + pid * 1024;
    }
    const char* tp = img + (t0 * S_RT) * (int64_t)IMGB;   // the current tile's rows (uniform)
    auto issue_part = [&](auto TS, const char* base, int n) {   // TS = ring position (0..3) of the tile the piece belongs to
        if (EXP & 1) return;
        if (w >= C::NDW) return;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + dma_off[n]),
                                         (__attribute__((address_space(3))) void*)(ring + decltype(TS)::value * 2 * S_SLOT + dma_dst[n]), 16, 0, 0);
    };
    u32 ab[4], ab_hi[4];                                  // (the offset field is 16 bits: slots 4..7 go through a second base)
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        ab[m] = lds_addr(ring) + (u32)(j * S_CKB + (((2 * m + h) ^ ((j >> 1) & 7)) * 16));
        ab_hi[m] = ab[m] + 4u * S_SLOT;
    }
    f16x8 fr[S_PRE];
#pragma unroll
    for (int m = 0; m < S_PRE; ++m) fr[m] = f16x8{};
    auto read_frag = [&](f16x8& dst, auto OFF, int t) {
        if (EXP & 2) { asm volatile("" : "+v"(dst)); return; }
        constexpr int off = decltype(OFF)::value;
        const u32 ad = off >= 4 * S_SLOT ? ab_hi[t & 3] : ab[t & 3];
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(ad), "n"(off >= 4 * S_SLOT ? off - 4 * S_SLOT : off));
    };
    auto frag_wait = [&](f16x8& f) {
        if (EXP & 2) return;
        asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(f) : "n"(S_PRE - 1));
    };
    u32 d_slow = 0, d_comp = 0, d_app = 0;
    unsigned long long d_clk_slow = 0, d_clk_bar = 0, d_clk_vm = 0, d_clk_all = DBG ? clock64() : 0;
    // keep the best K' of query lane jj's slot (sorted), raise its threshold, publish it (VMEM as inline asm: see scan_screen_ks_kernel)
    auto compact = [&](int jj) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const u32 n = (u32)__builtin_amdgcn_readlane((int)cnt, jj);
        u64* slot = (u64*)(((u64)(u32)__builtin_amdgcn_readlane((int)(u32)((u64)gslot >> 32), jj) << 32) |
                           (u64)(u32)__builtin_amdgcn_readlane((int)(u32)(u64)gslot, jj));
        u64 key[1];
        u32 rank[1];
        key[0] = 0ull;
        if ((u32)lane < n) asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(key[0]) : "v"(slot + lane) : "memory");
        rank_keys<1>(key, n, rank);
        const bool keep = (u32)lane < n && rank[0] < (u32)a.k;
        if (keep) asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(slot + rank[0]), "v"(key[0]) : "memory");
        const u64 kb = __ballot(keep && rank[0] == (u32)(a.k - 1));
        if (kb) {
            const u32 hi = (u32)__builtin_amdgcn_readlane((int)(u32)(key[0] >> 32), __builtin_ctzll(kb));
            if (j == jj) thr_s = fmaxf(thr_s, rmu_ord2f(hi) * 4096.0f);
            if (lane == 0) asm volatile("global_atomic_umax %0, %1, off sc1" ::"v"(gthr_w + jj), "v"(hi) : "memory");
        }
        if (j == jj) cnt = n < (u32)a.k ? n : (u32)a.k;
        if (DBG) ++d_comp;
    };
    auto slow_path = [&](const f32x16& p, int64_t rbase, u32 inmask) {
        unsigned long long c0 = 0;
        u32 todo = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) todo |= (p[r] > thr_s) ? (1u << r) : 0u;
        todo &= inmask;
        if (DBG) { ++d_slow; d_app += __builtin_popcount(todo); c0 = clock64(); }
        u32 uni = 0;
        for (u64 bl = __ballot(todo != 0); bl; bl &= bl - 1) uni |= (u32)__builtin_amdgcn_readlane((int)todo, __builtin_ctzll(bl));
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if ((uni >> r) & 1u) {
                bool has = ((todo >> r) & 1u) && p[r] > thr_s;
                u32 other = (u32)__shfl_xor((int)has, 32);
                const u64 full = __ballot(cnt + (u32)has + other > (u32)C::CAP);
                if (__builtin_expect(full != 0, 0)) {
                    for (u32 fm = (u32)full | (u32)(full >> 32); fm; fm &= fm - 1) compact(__builtin_ctz(fm));
                    has = has && p[r] > thr_s;
                    other = (u32)__shfl_xor((int)has, 32);
                }
                const u32 pos = cnt + (h ? other : 0u);
                if (has) {
                    const u64 key = rmu_make_key(p[r] * (1.0f / 4096.0f) + 0.0f, (u32)(rbase + (r & 3) + 8 * (r >> 2)));
                    asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(gslot + pos), "v"(key) : "memory");
                }
                cnt += (u32)has + other;
            }
        }
        if (DBG) d_clk_slow += clock64() - c0;
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
    if (ntiles > 0) {
        refresh_gthr();
        {
            const char* b0 = tp - 3 * S_RT * IMGB;        // dma_off carries the in-loop look-ahead of three tiles
#pragma unroll
            for (int n = 0; n < C::NIW; ++n) issue_part(I0{}, b0, n);
#pragma unroll
            for (int n = 0; n < C::NIW; ++n) issue_part(I1{}, b0 + S_RT * IMGB, n);
#pragma unroll
            for (int n = 0; n < C::NIW; ++n) issue_part(I2{}, b0 + 2 * S_RT * IMGB, n);
        }
        if (w < C::NDW) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // tiles 0 and 1 (and the thresholds)
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int m = 0; m < S_PRE; ++m) {
            if (!(EXP & 2)) asm volatile("ds_read_b128 %0, %1" : "=v"(fr[m]) : "v"(ab[m]));
        }
        f32x16 accA, accB;
#pragma unroll
        for (int r = 0; r < 16; ++r) { accB[r] = -INFINITY; accA[r] = 0.f; }
        const int64_t lane_r0 = a.row0 + t0 * S_RT + 4 * h;
        // one tile; P = its position in the ring (compile time): its chunks sit in slots 2P and 2P + 1
        auto tile_body = [&](auto PI, f32x16& acc, const f32x16& prev, int tl) {
            constexpr int P = decltype(PI)::value;
            unsigned long long cb = 0;
            if (DBG) cb = clock64();
            // in flight at most: this wave's operations of the previous tile (4 pieces of tile tl + 2 and a threshold refresh | the refresh)
            if (w < C::NDW) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
            if (DBG) { const unsigned long long cv = clock64(); d_clk_vm += cv - cb; cb = cv; }
            __builtin_amdgcn_s_barrier();                  // tile tl + 1 has landed; nobody reads tile tl - 1 any more
            if (DBG) d_clk_bar += clock64() - cb;
            {
                const u32 go = gt_lds[j];
                if (go && (a.share_thr & 1)) thr_s = fmaxf(thr_s, rmu_ord2f(go - 1u) * 4096.0f);
                if (w == PW && pace_live) pace_step(tl);
            }
            float mx = -INFINITY;
            auto step = [&](auto TI) {
                constexpr int gs = decltype(TI)::value, t = gs % S_CS, cch = gs / S_CS;
                constexpr int cur = 2 * P + cch, nxt = (cur + 1) % C::NR;
                if (gs == 18 && !(a.share_thr & 2) && !(EXP & 8) && __builtin_expect(__ballot(mx > thr_s) != 0, 0))
                    slow_path(prev, lane_r0 + (int64_t)(tl - 1) * S_RT, 0xffffu);
                frag_wait(fr[gs % S_PRE]);
                if (gs == 0) {
                    const f32x16 z = {};
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[gs % S_PRE], qh[gs], z, 0, 0, 0);
                } else {
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[gs % S_PRE], qh[gs], acc, 0, 0, 0);
                }
                if (gs >= 1 && gs <= 8 && !(EXP & 8)) asm("v_max3_f32 %0, %0, %1, %2" : "+v"(mx) : "v"(prev[2 * gs - 2]), "v"(prev[2 * gs - 1]));
                if (t + S_PRE < S_CS) read_frag(fr[gs % S_PRE], std::integral_constant<int, cur * S_SLOT + ((t + S_PRE) >> 2) * 128>{}, t + S_PRE);
                else read_frag(fr[gs % S_PRE], std::integral_constant<int, nxt * S_SLOT + ((t + S_PRE - S_CS) >> 2) * 128>{}, t + S_PRE - S_CS);
                if (gs % 6 == 1) issue_part(std::integral_constant<int, (P + 3) & 3>{}, tp, gs / 6);      // steps 1, 7, 13, 19
                if (gs == 4) refresh_gthr();
                __builtin_amdgcn_sched_barrier(0);
            };
            step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{}); step(std::integral_constant<int, 2>{});
            step(std::integral_constant<int, 3>{}); step(std::integral_constant<int, 4>{}); step(std::integral_constant<int, 5>{});
            step(std::integral_constant<int, 6>{}); step(std::integral_constant<int, 7>{}); step(std::integral_constant<int, 8>{});
            step(std::integral_constant<int, 9>{}); step(std::integral_constant<int, 10>{}); step(std::integral_constant<int, 11>{});
            step(std::integral_constant<int, 12>{}); step(std::integral_constant<int, 13>{}); step(std::integral_constant<int, 14>{});
            step(std::integral_constant<int, 15>{}); step(std::integral_constant<int, 16>{}); step(std::integral_constant<int, 17>{});
            step(std::integral_constant<int, 18>{}); step(std::integral_constant<int, 19>{}); step(std::integral_constant<int, 20>{});
            step(std::integral_constant<int, 21>{}); step(std::integral_constant<int, 22>{}); step(std::integral_constant<int, 23>{});
            tp += S_RT * IMGB;
        };
        for (int tl = 0; tl < ntiles; tl += 4) {           // ring period: four bodies (the accumulator parity alternates with it)
            tile_body(I0{}, accA, accB, tl);                 // (tile -1 = the -inf accumulators: nothing passes)
            if (tl + 1 < ntiles) tile_body(I1{}, accB, accA, tl + 1);
            if (tl + 2 < ntiles) tile_body(I2{}, accA, accB, tl + 2);
            if (tl + 3 < ntiles) tile_body(I3{}, accB, accA, tl + 3);
        }
        if (pace_on && w == PW && lane == 0) __hip_atomic_store(prog_w + qt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int m = 0; m < S_PRE; ++m) asm volatile("" : "+v"(fr[m]));
        {
            const bool last_in_a = ((ntiles - 1) & 1) == 0;
            f32x16 last;
            const int64_t rbl = lane_r0 + (int64_t)(ntiles - 1) * S_RT, row_end = a.row0 + a.n_rows;
            u32 inmask = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                last[r] = last_in_a ? accA[r] : accB[r];
                inmask |= (rbl + (r & 3) + 8 * (r >> 2) < row_end) ? (1u << r) : 0u;
            }
            if (!(a.share_thr & 2)) slow_path(last, rbl, inmask);
        }
    }
    if (DBG) {
        u32 app = d_app;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) app += __shfl_xor(app, o);
        if (lane == 0) {
            atomicAdd((unsigned long long*)a.dbg + 0, (unsigned long long)d_slow);
            atomicAdd((unsigned long long*)a.dbg + 1, (unsigned long long)d_comp);
            atomicAdd((unsigned long long*)a.dbg + 2, (unsigned long long)app);
            atomicAdd((unsigned long long*)a.dbg + 3, (unsigned long long)ntiles);
            atomicAdd((unsigned long long*)a.dbg + 5, d_clk_slow);
            atomicAdd((unsigned long long*)a.dbg + 6, d_clk_bar);
            atomicAdd((unsigned long long*)a.dbg + 8, d_clk_vm);
            atomicAdd((unsigned long long*)a.dbg + 7, (unsigned long long)(clock64() - d_clk_all));
        }
    }
    // ---- emit: best K' approximate candidates of this (chunk, query), sorted.  Eight slots are read back per round trip.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int part = s_idx;
    for (int j0 = 0; j0 < 32; j0 += 8) {
        if (q_base + j0 >= a.nq) break;
        u64 key[8][1];
        u32 nn[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int jj = j0 + e;
            nn[e] = (u32)__builtin_amdgcn_readlane((int)cnt, jj);
            const u64* slot = (const u64*)(((u64)(u32)__builtin_amdgcn_readlane((int)(u32)((u64)gslot >> 32), jj) << 32) |
                                           (u64)(u32)__builtin_amdgcn_readlane((int)(u32)(u64)gslot, jj));
            key[e][0] = (u32)lane < nn[e] ? __hip_atomic_load(slot + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int qq = q_base + j0 + e;
            if (qq < a.nq) {
                u32 rank[1];
                rank_keys<1>(key[e], nn[e], rank);
                u64* dst = a.partial + ((size_t)part * a.nq + qq) * a.k;
                if ((u32)lane < nn[e]) {
                    if (rank[0] < (u32)a.k) dst[rank[0]] = key[e][0];
                } else if (lane < a.k) {
                    dst[lane] = 0ull;
                }
            }
        }
    }
}

// ---- lean form, one barrier per TWO tiles (round 4) ----------------------------------------------------------------------------------------------
// scan_screen_lean2_kernel still waits 354 cycles per tile at its one barrier: eight waves with per-tile jitter, six of them carrying the DMA.
// Here the ring holds SIX tiles (12 slots, 144 KiB -- the LDS has nothing else to hold), is handed over once per PAIR of tiles, and every wave
// carries three of a tile's 24 DMA pieces (the pacing store of the last wave is a plain store issued a whole pair before the next counted
// wait: it cannot hold the ring up).  At the barrier of pair p the tiles up to 2p + 2 have landed (the fragment prefetch crosses into the
// next pair's first tile), 2p + 3 may be in flight, and pair p + 2 is issued during pair p into the slots of pair p - 1.  Six bodies.
// NWV = 8: full query tiles (256 queries per workgroup, two waves per SIMD).  NWV = 4: ONE query tile (batches <= 128 queries: every image byte
// is read by exactly one workgroup -- NT streams it past the L2 with non-temporal loads; the HBM-bound regime, where the round-3 form's ~300
// instructions per tile and wave cost bandwidth: one wave per SIMD cannot issue them and keep 24 KiB per microsecond in flight).
template <int NWV, bool DEEP = false>
struct Lean3Cfg {
    static constexpr int NW = NWV, QW = 32, NR = 12, NDW = NWV, NIW = 24 / NWV;
    // DEEP (round 6): 32 < k <= 104 -- K' <= 120 candidates per (chunk, query) slot of 128 keys, two keys per lane wherever the whole wave works
    // on one slot (compaction, emit); everything else -- ring, MFMA chain, filter, appends -- is the K' <= 40 kernel unchanged
    static constexpr int CAP = DEEP ? RMU_KS_CAP_DEEP : RMU_KS_CAP, NPL = DEEP ? 2 : 1;
    static constexpr int RING_BYTES = NR * S_SLOT;
    static constexpr int GT_OFF = RING_BYTES;
    static constexpr int NRM_OFF = GT_OFF + NW * 256;      // L2 form: -2048 |x|^2 of the rows of the six ring tiles (32 floats per tile)
    static constexpr int LDS_BYTES = NRM_OFF + 6 * S_RT * 4;
};

// L2N = 1 (round 5): the index ranks by 2 q.x - |x|^2 (RMU_METRIC_L2SQ).  The image has no k-slot left for the norm (384 fp16 = the 24 MFMA
// steps exactly), so it enters as the chain's C operand: a.nrm[row] = -2048 |x|^2 (fp32, built at add time from the exact scan's own
// -|x|^2 column) initialises the accumulator of the tile's first MFMA -- acc = 4096 (q~.x~ - |x|^2 / 2), the approximate HALF score; filter,
// thresholds, candidate keys and merges never know.  A lane's 16 accumulator rows (4h + 8i + c) are four 16-byte LDS reads, issued half a
// tile ahead between two fragment reads (the lgkmcnt of the four steps behind them counts them in); the norms of a PAIR of tiles are one
// 256-byte LDS-DMA by wave 0, issued two pairs ahead next to the threshold refresh (older than the pieces the pair barrier's vmcnt leaves
// in flight, like the refresh).  +4 KiB-reads per 24 on the LDS return path; the fp16 image bytes are unchanged.
template <int EXP = 0, int NWV = 8, int NT = 0, int L2N = 0, bool DEEP = false>
__global__ __launch_bounds__(64 * NWV) void scan_screen_lean3_kernel(const ScanLaunch a) {
    using C = Lean3Cfg<NWV, DEEP>;
    constexpr bool DBG = (EXP & 4) != 0;
    constexpr int NW = NWV, S_PRE = 4;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
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
    const char* img = (const char*)a.x + a.row0 * (int64_t)IMGB;
    char* ring = ssm;
    const int q_base = (qt * NW + w) * C::QW;
    const bool q_ok = q_base + j < a.nq;
    const bool wave_live = q_base < a.nq;                 // (uniform)
    float thr_s = q_ok ? -INFINITY : INFINITY;            // 4096 * max(own k-th best, shared threshold): only ever rises
    u32 cnt = 0;                                          // entries in this lane's query slot (equal in lanes j and j + 32)
    u64* const gslot = a.gcand + ((size_t)s_idx * a.nq + (q_ok ? q_base + j : 0)) * C::CAP;
    u32* gthr_w = a.gthr + q_base;
    const u32* gt_lds = (const u32*)(ssm + C::GT_OFF) + w * 64;
    const bool pace_on = a.prog != nullptr;               // sibling pacing: see scan_screen_kernel
    u32* prog_w = a.prog + (size_t)s_idx * 4;
    bool pace_live = pace_on;
    const u32* gsrc = gthr_w + j;
    if (pace_on && lane >= 32 && lane < 36) gsrc = prog_w + (lane - 32);
    auto refresh_gthr = [&]() {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                         (__attribute__((address_space(3))) void*)(ssm + C::GT_OFF + w * 256), 4, 0, 16);
    };
    constexpr int PW = NW - 1;
    auto pace_step = [&](int tile) {
        if (lane == 0) __hip_atomic_store(prog_w + qt, ~(u32)tile, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const u32 m4 = max(max(gt_lds[32], gt_lds[33]), max(gt_lds[34], gt_lds[35]));
        int lead = m4 ? tile - (int)~m4 : -1;
        if (__builtin_expect(__builtin_amdgcn_readfirstlane(lead) > a.pace, 0)) {
            int spins = 0;
            do {
                __builtin_amdgcn_s_sleep(24);
                u32 v = 0;
                if (lane < 4) v = __hip_atomic_load(prog_w + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                v = max(v, (u32)__shfl_xor((int)v, 1));
                v = max(v, (u32)__shfl_xor((int)v, 2));
                const u32 vm = (u32)__builtin_amdgcn_readfirstlane((int)v);
                lead = vm ? tile - (int)~vm : -1;
            } while (lead > a.pace && ++spins < 400);
            if (spins >= 400) pace_live = false;
        }
    };
    f16x8 qh[S_TS];
    {
        const char* qrow = (const char*)a.q + (size_t)(q_ok ? q_base + j : 0) * IMGB + h * 16;
#pragma unroll
        for (int T = 0; T < S_TS; ++T) qh[T] = *(const f16x8*)(qrow + T * 32);
#pragma unroll
        for (int T = 0; T < S_TS; ++T) asm volatile("" : "+v"(qh[T]));     // complete before any LDS-DMA (see scan_screen_kernel)
    }
    // DMA: wave w carries pieces n * NW + w (n = 0..NIW-1) of a tile's 24 = 12 * half + piece; issued during tile t they belong to tile t + 4
    u32 dma_off[C::NIW];
    int dma_dst[C::NIW];
#pragma unroll
    for (int n = 0; n < C::NIW; ++n) {
        const int id = n * C::NDW + w;
        const int half = id / 12, pid = id % 12;
        const int f = pid * 64 + lane;
        const int i = f / S_U16, p = f % S_U16;
        dma_off[n] = (u32)(i * IMGB + (p ^ ((i >> 1) & 7)) * 16 + half * S_CKB) + 4u * S_RT * IMGB;
        dma_dst[n] = half * S_SLOT + pid * 1024;
    }
    const char* tp = img + (t0 * S_RT) * (int64_t)IMGB;   // the current tile's rows (uniform)
    auto issue_part = [&](auto TS, const char* base, int n) {   // TS = ring position (0..5) of the tile the piece belongs to
        if (EXP & 1) return;
        if (NT)        // literal aux operands only
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + dma_off[n]),
                                             (__attribute__((address_space(3))) void*)(ring + decltype(TS)::value * 2 * S_SLOT + dma_dst[n]), 16, 0, 2);
        else
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + dma_off[n]),
                                             (__attribute__((address_space(3))) void*)(ring + decltype(TS)::value * 2 * S_SLOT + dma_dst[n]), 16, 0, 0);
    };
    u32 ab[4], ab_hi[4], ab_h2[4];                        // (the offset field is 16 bits: slots 4..7 and 8..11 go through their own bases)
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        ab[m] = lds_addr(ring) + (u32)(j * S_CKB + (((2 * m + h) ^ ((j >> 1) & 7)) * 16));
        ab_hi[m] = ab[m] + 4u * S_SLOT;
        ab_h2[m] = ab[m] + 8u * S_SLOT;
    }
    f16x8 fr[S_PRE];
#pragma unroll
    for (int m = 0; m < S_PRE; ++m) fr[m] = f16x8{};
    auto read_frag = [&](f16x8& dst, auto OFF, int t) {
        if (EXP & 2) { asm volatile("" : "+v"(dst)); return; }
        constexpr int off = decltype(OFF)::value;
        const u32 ad = off >= 8 * S_SLOT ? ab_h2[t & 3] : off >= 4 * S_SLOT ? ab_hi[t & 3] : ab[t & 3];
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(ad), "n"(off % (4 * S_SLOT)));
    };
    auto frag_wait = [&](f16x8& f) {
        if (EXP & 2) return;
        asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(f) : "n"(S_PRE - 1));
    };
    auto frag_wait_nrm = [&](f16x8& f) {                   // the four norm reads sit between this fragment's read and the newest ones
        if (EXP & 2) return;
        asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(f) : "n"(S_PRE - 1 + 4));
    };
    // L2 form: row norms of the ring tiles
    const float* np = L2N ? a.nrm + a.row0 + t0 * S_RT + lane : nullptr;     // one float per lane = the 64 rows of a pair of tiles
    const u32 nrm_ad = lds_addr(ssm + C::NRM_OFF) + (u32)h * 16u;
    f32x4 zq[4] = {};
    auto issue_nrm = [&](auto PI, const float* src) {      // PI = ring position of the pair's first tile
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(ssm + C::NRM_OFF + decltype(PI)::value * S_RT * 4), 4, 0, 0);
    };
    auto read_nrm = [&](auto PI) {                         // this lane's 16 accumulator rows of the tile at ring position PI: rows 4h + 8i + (0..3)
        constexpr int o = decltype(PI)::value * S_RT * 4;
        f32x4 &z0 = zq[0], &z1 = zq[1], &z2 = zq[2], &z3 = zq[3];      // (asm operands alone do not capture in a generic lambda)
        const u32 ad = nrm_ad;
        asm volatile("ds_read_b128 %0, %4 offset:%5\n\tds_read_b128 %1, %4 offset:%6\n\tds_read_b128 %2, %4 offset:%7\n\tds_read_b128 %3, %4 offset:%8"
                     : "=v"(z0), "=v"(z1), "=v"(z2), "=v"(z3)
                     : "v"(ad), "n"(o), "n"(o + 32), "n"(o + 64), "n"(o + 96));
    };
    u32 d_slow = 0, d_comp = 0, d_app = 0;
    unsigned long long d_clk_slow = 0, d_clk_bar = 0, d_clk_vm = 0, d_clk_all = DBG ? clock64() : 0;
    // keep the best K' of query lane jj's slot (sorted), raise its threshold, publish it (VMEM as inline asm: see scan_screen_ks_kernel)
    auto compact = [&](int jj) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const u32 n = (u32)__builtin_amdgcn_readlane((int)cnt, jj);
        u64* slot = (u64*)(((u64)(u32)__builtin_amdgcn_readlane((int)(u32)((u64)gslot >> 32), jj) << 32) |
                           (u64)(u32)__builtin_amdgcn_readlane((int)(u32)(u64)gslot, jj));
        u64 key[C::NPL];
        u32 rank[C::NPL];
#pragma unroll
        for (int pp = 0; pp < C::NPL; ++pp) {
            key[pp] = 0ull;
            if ((u32)(lane + 64 * pp) < n) asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=v"(key[pp]) : "v"(slot + lane + 64 * pp) : "memory");
        }
#pragma unroll
        for (int pp = 0; pp < C::NPL; ++pp) asm volatile("s_waitcnt vmcnt(0)" : "+v"(key[pp])::"memory");
        rank_keys<C::NPL>(key, n, rank);
#pragma unroll
        for (int pp = 0; pp < C::NPL; ++pp) {
            const bool keep = (u32)(lane + 64 * pp) < n && rank[pp] < (u32)a.k;
            if (keep) asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(slot + rank[pp]), "v"(key[pp]) : "memory");
            const u64 kb = __ballot(keep && rank[pp] == (u32)(a.k - 1));
            if (kb) {
                const u32 hi = (u32)__builtin_amdgcn_readlane((int)(u32)(key[pp] >> 32), __builtin_ctzll(kb));
                if (j == jj) thr_s = fmaxf(thr_s, rmu_ord2f(hi) * 4096.0f);
                if (lane == 0) asm volatile("global_atomic_umax %0, %1, off sc1" ::"v"(gthr_w + jj), "v"(hi) : "memory");
            }
        }
        if (j == jj) cnt = n < (u32)a.k ? n : (u32)a.k;
        if (DBG) ++d_comp;
    };
    auto slow_path = [&](const f32x16& p, int64_t rbase, u32 inmask) {
        unsigned long long c0 = 0;
        u32 todo = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) todo |= (p[r] > thr_s) ? (1u << r) : 0u;
        todo &= inmask;
        if (DBG) { ++d_slow; d_app += __builtin_popcount(todo); c0 = clock64(); }
        u32 uni = 0;
        for (u64 bl = __ballot(todo != 0); bl; bl &= bl - 1) uni |= (u32)__builtin_amdgcn_readlane((int)todo, __builtin_ctzll(bl));
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if ((uni >> r) & 1u) {
                bool has = ((todo >> r) & 1u) && p[r] > thr_s;
                u32 other = (u32)__shfl_xor((int)has, 32);
                const u64 full = __ballot(cnt + (u32)has + other > (u32)C::CAP);
                if (__builtin_expect(full != 0, 0)) {
                    for (u32 fm = (u32)full | (u32)(full >> 32); fm; fm &= fm - 1) compact(__builtin_ctz(fm));
                    has = has && p[r] > thr_s;
                    other = (u32)__shfl_xor((int)has, 32);
                }
                const u32 pos = cnt + (h ? other : 0u);
                if (has) {
                    const u64 key = rmu_make_key(p[r] * (1.0f / 4096.0f) + 0.0f, (u32)(rbase + (r & 3) + 8 * (r >> 2)));
                    asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(gslot + pos), "v"(key) : "memory");
                }
                cnt += (u32)has + other;
            }
        }
        if (DBG) d_clk_slow += clock64() - c0;
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
    using I4 = std::integral_constant<int, 4>; using I5 = std::integral_constant<int, 5>;
    bool direct = false;      // (uniform per wave) this wave's lists are already in a.partial: see the cold tile below
    if (ntiles > 0) {
        refresh_gthr();
        if (L2N) {                                        // (older than every piece: complete at the first counted wait)
            if (w == 0) {
                issue_nrm(I0{}, np);
                issue_nrm(I2{}, np + 2 * S_RT);
            }
            np += 4 * S_RT;
        }
        {
            const char* b0 = tp - 4 * S_RT * IMGB;        // dma_off carries the in-loop look-ahead of four tiles
#pragma unroll
            for (int n = 0; n < C::NIW; ++n) issue_part(I0{}, b0, n);
#pragma unroll
            for (int n = 0; n < C::NIW; ++n) issue_part(I1{}, b0 + S_RT * IMGB, n);
#pragma unroll
            for (int n = 0; n < C::NIW; ++n) issue_part(I2{}, b0 + 2 * S_RT * IMGB, n);
#pragma unroll
            for (int n = 0; n < C::NIW; ++n) issue_part(I3{}, b0 + 3 * S_RT * IMGB, n);
        }
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::NIW) : "memory");   // tiles 0, 1, 2 (and the thresholds)
        __builtin_amdgcn_s_barrier();
        if (L2N && (NWV == 8 || wave_live)) read_nrm(I0{});   // (in front of the fragment prefetch: step 0's counted wait covers it)
#pragma unroll
        for (int m = 0; m < S_PRE; ++m) {
            if (!(EXP & 2)) asm volatile("ds_read_b128 %0, %1" : "=v"(fr[m]) : "v"(ab[m]));
        }
        f32x16 accA, accB;
#pragma unroll
        for (int r = 0; r < 16; ++r) { accB[r] = -INFINITY; accA[r] = 0.f; }
        const int64_t lane_r0 = a.row0 + t0 * S_RT + 4 * h;
        // one tile; P = its position in the six-tile ring (compile time): its chunks sit in slots 2P and 2P + 1
        auto tile_body = [&](auto PI, f32x16& acc, const f32x16& prev, int tl) {
            constexpr int P = decltype(PI)::value;
            if (P % 2 == 0) {                              // a pair of tiles starts
                unsigned long long cb = 0;
                if (DBG) cb = clock64();
                // in flight at most: the NIW pieces (tile tl + 3) this wave issued during the previous tile; the refresh is older
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::NIW) : "memory");
                if (DBG) { const unsigned long long cv = clock64(); d_clk_vm += cv - cb; cb = cv; }
                __builtin_amdgcn_s_barrier();              // tiles up to tl + 2 have landed; nobody reads the previous pair any more
                if (DBG) d_clk_bar += clock64() - cb;
                const u32 go = gt_lds[j];
                if (go && (a.share_thr & 1)) thr_s = fmaxf(thr_s, rmu_ord2f(go - 1u) * 4096.0f);
                if (w == PW && pace_live) pace_step(tl);
            }
            // (Measured per wave: waves 0-3 wait ~600 cycles per tile at the pair barrier, waves 4-7 ~75 -- issue arbitration between the two waves
            // of a SIMD is by age.  Giving the younger half s_setprio 1 for the first tile of every pair halves the total wait and changes
            // the kernel's time by nothing: the SIMD's throughput, not the rendezvous, sets it.)
            float mx = -INFINITY;
            auto step = [&](auto TI) {
                constexpr int gs = decltype(TI)::value, t = gs % S_CS, cch = gs / S_CS;
                constexpr int cur = 2 * P + cch, nxt = (cur + 1) % C::NR;
                // (a wave none of whose 32 queries exist -- batches below 97 queries in the one-tile form -- only carries its DMA pieces)
                if (NWV == 8 || wave_live) {
                if (gs == 18 && !(a.share_thr & 2) && !(EXP & 8) && __builtin_expect(__ballot(mx > thr_s) != 0, 0))
                    slow_path(prev, lane_r0 + (int64_t)(tl - 1) * S_RT, 0xffffu);
                if (L2N && gs >= 13 && gs <= 16) frag_wait_nrm(fr[gs % S_PRE]);
                else frag_wait(fr[gs % S_PRE]);
                if (gs == 0 && L2N) {
                    const f32x16 z = {zq[0][0], zq[0][1], zq[0][2], zq[0][3], zq[1][0], zq[1][1], zq[1][2], zq[1][3],
                                      zq[2][0], zq[2][1], zq[2][2], zq[2][3], zq[3][0], zq[3][1], zq[3][2], zq[3][3]};
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[gs % S_PRE], qh[gs], z, 0, 0, 0);
                } else if (gs == 0) {
                    const f32x16 z = {};
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[gs % S_PRE], qh[gs], z, 0, 0, 0);
                } else {
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[gs % S_PRE], qh[gs], acc, 0, 0, 0);
                }
                // (the MFMAs as inline asm with the wait in the same statement -- no compiler s_nop padding -- measured no faster: 6.64-6.67 vs 6.54)
                if (gs >= 1 && gs <= 8 && !(EXP & 8)) asm("v_max3_f32 %0, %0, %1, %2" : "+v"(mx) : "v"(prev[2 * gs - 2]), "v"(prev[2 * gs - 1]));
                if (t + S_PRE < S_CS) read_frag(fr[gs % S_PRE], std::integral_constant<int, cur * S_SLOT + ((t + S_PRE) >> 2) * 128>{}, t + S_PRE);
                else read_frag(fr[gs % S_PRE], std::integral_constant<int, nxt * S_SLOT + ((t + S_PRE - S_CS) >> 2) * 128>{}, t + S_PRE - S_CS);
                if (L2N && gs == 12) read_nrm(std::integral_constant<int, (P + 1) % 6>{});        // the NEXT tile's norms (landed: see the kernel's header)
                }
                if (gs % (S_TS / C::NIW) == 1) issue_part(std::integral_constant<int, (P + 4) % 6>{}, tp, gs / (S_TS / C::NIW));   // steps 1, 9, 17 | 1, 5, .., 21
                if (gs == 4 && P % 2 == 0) refresh_gthr();
                if (L2N && gs == 4 && P % 2 == 0) {       // norms of the pair two pairs ahead, into the slots of the pair that has just been left
                    if (w == 0) issue_nrm(std::integral_constant<int, (P + 4) % 6>{}, np);
                    np += 2 * S_RT;
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{}); step(std::integral_constant<int, 2>{});
            step(std::integral_constant<int, 3>{}); step(std::integral_constant<int, 4>{}); step(std::integral_constant<int, 5>{});
            step(std::integral_constant<int, 6>{}); step(std::integral_constant<int, 7>{}); step(std::integral_constant<int, 8>{});
            step(std::integral_constant<int, 9>{}); step(std::integral_constant<int, 10>{}); step(std::integral_constant<int, 11>{});
            step(std::integral_constant<int, 12>{}); step(std::integral_constant<int, 13>{}); step(std::integral_constant<int, 14>{});
            step(std::integral_constant<int, 15>{}); step(std::integral_constant<int, 16>{}); step(std::integral_constant<int, 17>{});
            step(std::integral_constant<int, 18>{}); step(std::integral_constant<int, 19>{}); step(std::integral_constant<int, 20>{});
            step(std::integral_constant<int, 21>{}); step(std::integral_constant<int, 22>{}); step(std::integral_constant<int, 23>{});
            tp += S_RT * IMGB;
        };
        for (int tl = 0; tl < ntiles; tl += 6) {           // ring period: six bodies (the accumulator parity alternates with it)
            tile_body(I0{}, accA, accB, tl);                 // (tile -1 = the -inf accumulators: nothing passes)
            if (tl + 1 < ntiles) tile_body(I1{}, accB, accA, tl + 1);
            if (tl + 2 < ntiles) tile_body(I2{}, accA, accB, tl + 2);
            if (tl + 3 < ntiles) tile_body(I3{}, accB, accA, tl + 3);
            if (tl + 4 < ntiles) tile_body(I4{}, accA, accB, tl + 4);
            if (tl + 5 < ntiles) tile_body(I5{}, accB, accA, tl + 5);
        }
        if (pace_on && w == PW && lane == 0) __hip_atomic_store(prog_w + qt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int m = 0; m < S_PRE; ++m) asm volatile("" : "+v"(fr[m]));
        {
            const bool last_in_a = ((ntiles - 1) & 1) == 0;
            f32x16 last;
            const int64_t rbl = lane_r0 + (int64_t)(ntiles - 1) * S_RT, row_end = a.row0 + a.n_rows;
            u32 inmask = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                last[r] = last_in_a ? accA[r] : accB[r];
                inmask |= (rbl + (r & 3) + 8 * (r >> 2) < row_end) ? (1u << r) : 0u;
            }
            // (round 6, second session) COLD tile of the ladder's first launch -- one tile per chunk, empty thresholds: every row of the tile is a
            // candidate of every query.  Through slow_path that is 16 store instructions of 64 scattered 8-byte keys per wave (the slots are
            // query-major, a query's lanes 384 B apart: every lane its own line request -- 112 k of a wave's 174 k cycles in the debug build), a
            // read-back and a copy.  Here the wave's 32 x 32 keys go through 8 KiB of the (now idle) ring and straight into a.partial, a store
            // instruction = two queries' lists = 512 contiguous bytes; the slots are never touched and the emit skips this wave.  The lists leave
            // unsorted (share_thr bit 2: the launch's merge knows).  The barrier is workgroup-uniform (share_thr, ntiles); `cold` is per wave.
            bool cold = false;
            if ((a.share_thr & 4) && !(a.share_thr & 2) && ntiles == 1 && a.k >= 32) {
                __builtin_amdgcn_s_barrier();            // every wave's DMA has landed (the drain above) and nobody reads the ring any more
                u32 todo = 0;
#pragma unroll
                for (int r = 0; r < 16; ++r) todo |= (last[r] > thr_s) ? (1u << r) : 0u;
                todo &= inmask;
                cold = __ballot(q_ok && (cnt != 0u || todo != 0xffffu)) == 0ull;
            }
            if (cold) {
                u64* tl = (u64*)(ring + w * 8192);       // [query 0..31][position 0..31]: lane (j, h) owns positions [16 h, +16) of query j
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    tl[j * 32 + 16 * h + r] = rmu_make_key(last[r] * (1.0f / 4096.0f) + 0.0f, (u32)(rbl + (r & 3) + 8 * (r >> 2)));
                __builtin_amdgcn_wave_barrier();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                u64* const pbase = a.partial + ((size_t)s_idx * a.nq + q_base) * a.k;
#pragma unroll
                for (int it = 0; it < 16; ++it) {
                    const int lin = it * 64 + lane, qq = lin >> 5, pos = lin & 31;
                    const u64 key = tl[lin];
                    if (q_base + qq < a.nq) pbase[(size_t)qq * a.k + pos] = key;
                }
                const int extra = a.k - 32;                // positions 32 .. K' - 1 of every list: zeros
                for (int e = lane; e < 32 * extra; e += 64) {
                    const int qq = e / extra, pos = 32 + e % extra;
                    if (q_base + qq < a.nq) pbase[(size_t)qq * a.k + pos] = 0ull;
                }
                direct = true;
            } else if (!(a.share_thr & 2)) slow_path(last, rbl, inmask);
        }
    }
    if (DBG) {
        u32 app = d_app;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) app += __shfl_xor(app, o);
        if (lane == 0) {
            atomicAdd((unsigned long long*)a.dbg + 0, (unsigned long long)d_slow);
            atomicAdd((unsigned long long*)a.dbg + 1, (unsigned long long)d_comp);
            atomicAdd((unsigned long long*)a.dbg + 2, (unsigned long long)app);
            atomicAdd((unsigned long long*)a.dbg + 3, (unsigned long long)ntiles);
            atomicAdd((unsigned long long*)a.dbg + 5, d_clk_slow);
            atomicAdd((unsigned long long*)a.dbg + 6, d_clk_bar);
            atomicAdd((unsigned long long*)a.dbg + 8, d_clk_vm);
            if (w < 7) atomicAdd((unsigned long long*)a.dbg + 9 + w, d_clk_bar);     // barrier wait of waves 0..6 ("seg" + following words in the dump)
            atomicAdd((unsigned long long*)a.dbg + 7, (unsigned long long)(clock64() - d_clk_all));
        }
    }
    // ---- emit: best K' approximate candidates of this (chunk, query), sorted.  All 32 slots (DEEP: 8 at a time) are read back in ONE round trip (the query fragments are dead: registers are free).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int part = s_idx;
    constexpr int GE = DEEP ? 8 : 32;
    // share_thr bit 2 (the ladder's FIRST launch: rmu_api.hip screen_enqueue): slots of at most K' entries are written as they are -- compact,
    // zeros behind them, NOT sorted -- and that launch's merge runs in its unsorted mode (topk_merge.hip).  A cold range is one tile per chunk,
    // every row a candidate: sorting 32 keys for each of a workgroup's 256 queries was ~25 us of VALU time in a launch that scans 2 000 rows.
    const bool emit_raw = (a.share_thr & 4) != 0;
    // (round 6, second session) ONE query tile (NWV = 4, batches <= 128 queries): the queries live in the first ceil(nq / 32) waves, and a slot
    // is sorted by a whole wave, one query after the other (rank_keys: n x ~8 instructions per query) -- a COLD range (the ladder's first: every
    // row of its tiles is a candidate, n = 32..40) kept ONE wave busy for ~1.6 us per query while three idled: 34 us of the batch-16 search and
    // 59 us of the batch-32 one (profiles/r06_search_timeline.txt).  The slots are in global memory, so any wave can sort any query: the
    // counts go through the waves' (now idle) threshold words in LDS and query q is emitted by wave q % 4.  Same keys, same ranks, same bytes.
    if constexpr (NWV == 4) {
        if (lane < 32) lds_store_b32(lds_addr(ssm + C::GT_OFF + w * 256) + (u32)lane * 4u, direct ? 0xffffffffu : cnt);   // (marker: the list is already written)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const u32* call = (const u32*)(ssm + C::GT_OFF);
        for (int e0 = 0; e0 < 32; e0 += GE) {
            if (w + 4 * e0 >= a.nq) break;
            u64 key[GE][C::NPL];
            u32 nn[GE];
            bool done[GE];
#pragma unroll
            for (int e = 0; e < GE; ++e) {
                const int qq = w + 4 * (e0 + e);
                const bool ok = qq < a.nq;
                nn[e] = ok ? (u32)__builtin_amdgcn_readfirstlane((int)call[(qq >> 5) * 64 + (qq & 31)]) : 0u;
                done[e] = nn[e] == 0xffffffffu;
                if (done[e]) nn[e] = 0u;
                const u64* slot = a.gcand + ((size_t)s_idx * a.nq + (ok ? qq : 0)) * C::CAP;
#pragma unroll
                for (int pp = 0; pp < C::NPL; ++pp)
                    key[e][pp] = (u32)(lane + 64 * pp) < nn[e] ? __hip_atomic_load(slot + lane + 64 * pp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
            }
#pragma unroll
            for (int e = 0; e < GE; ++e) {
                const int qq = w + 4 * (e0 + e);
                if (qq < a.nq && !done[e]) {
                    u64* dst = a.partial + ((size_t)part * a.nq + qq) * a.k;
                    if (emit_raw && nn[e] <= (u32)a.k) {       // (uniform) the slot as it is, zeros behind it: the merge of this launch does not need it sorted
#pragma unroll
                        for (int pp = 0; pp < C::NPL; ++pp) {
                            const int le = lane + 64 * pp;
                            if (le < a.k) dst[le] = (u32)le < nn[e] ? key[e][pp] : 0ull;
                        }
                        continue;
                    }
                    u32 rank[C::NPL];
                    rank_keys<C::NPL>(key[e], nn[e], rank);
#pragma unroll
                    for (int pp = 0; pp < C::NPL; ++pp) {
                        const int le = lane + 64 * pp;
                        if ((u32)le < nn[e]) {
                            if (rank[pp] < (u32)a.k) dst[rank[pp]] = key[e][pp];
                        } else if (le < a.k) {
                            dst[le] = 0ull;
                        }
                    }
                }
            }
        }
    } else {
    for (int j0 = 0; j0 < 32; j0 += GE) {
        if (q_base + j0 >= a.nq || direct) break;
        u64 key[GE][C::NPL];
        u32 nn[GE];
#pragma unroll
        for (int e = 0; e < GE; ++e) {
            const int jj = j0 + e;
            nn[e] = (u32)__builtin_amdgcn_readlane((int)cnt, jj);
            const u64* slot = (const u64*)(((u64)(u32)__builtin_amdgcn_readlane((int)(u32)((u64)gslot >> 32), jj) << 32) |
                                           (u64)(u32)__builtin_amdgcn_readlane((int)(u32)(u64)gslot, jj));
#pragma unroll
            for (int pp = 0; pp < C::NPL; ++pp)
                key[e][pp] = (u32)(lane + 64 * pp) < nn[e] ? __hip_atomic_load(slot + lane + 64 * pp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
        }
#pragma unroll
        for (int e = 0; e < GE; ++e) {
            const int qq = q_base + j0 + e;
            if (qq < a.nq) {
                u64* dst = a.partial + ((size_t)part * a.nq + qq) * a.k;
                if (emit_raw && nn[e] <= (u32)a.k) {
#pragma unroll
                    for (int pp = 0; pp < C::NPL; ++pp) {
                        const int le = lane + 64 * pp;
                        if (le < a.k) dst[le] = (u32)le < nn[e] ? key[e][pp] : 0ull;
                    }
                    continue;
                }
                u32 rank[C::NPL];
                rank_keys<C::NPL>(key[e], nn[e], rank);
#pragma unroll
                for (int pp = 0; pp < C::NPL; ++pp) {
                    const int le = lane + 64 * pp;
                    if ((u32)le < nn[e]) {
                        if (rank[pp] < (u32)a.k) dst[rank[pp]] = key[e][pp];
                    } else if (le < a.k) {
                        dst[le] = 0ull;
                    }
                }
            }
        }
    }
    }
}

// ---- K-SPLIT form of the screening scan (round 4; full 256-query tiles) --------------------------------------------------------------
// What bounds the 8-wave kernel above is the LDS return path, not the matrix pipe: 128 B/clk per CU = 32 B/clk per SIMD, and ONE 1-KiB A
// fragment per 32-cycle MFMA is exactly that rate (measured with the ping-pong form: 8 cycles per KiB and CU): 0.59 MFMA busy.  Feeding
// two MFMAs from every fragment needs 64 queries per wave -- 192 registers of query fragments, which only fits one wave per SIMD (the
// 4-wave G = 2 form: issue-bound, 0.35).  Here the two waves of a SIMD pair (w, w + 4) hold the SAME 64 queries and split K instead: wave
// half sh reads only the half-k chunk sh of every row tile (12 fragments, 12 KiB) and multiplies it with its k half of both 32-query groups
// -- 96 registers of query fragments, 24 MFMAs per 12 fragment reads, two waves per SIMD.  The price is an exchange: each wave keeps the
// partial sums of the group it OWNS (group sh of the pair) and hands the other group's 32 x 32 partials to its partner through 4 KiB of
// LDS (16 KiB of LDS traffic per tile and wave in all instead of 24), one tile behind the MFMAs:
//   tile t, behind ring barrier A : ds_write the send-partials of tile t - 1                     (the partner has finished reading t - 2's)
//   step 5, behind barrier B      : ds_read the partner's partials of tile t - 1                 (waited for by the counted wait of step 9)
//   steps 9-11                    : full score = own partial + partner's; 16 compares against the lane's threshold; ONE branch
// fp32 addition of two 192-term fp32 sums instead of one 384-term chain: the same 408 * 2^-24 bound as in the header.
// The 80 KiB of LDS candidate slots do not fit beside a ring deep enough for the faster tiles (4 tiles = 96 KiB) and the exchange area, so
// candidates live in GLOBAL memory (a.gcand: [chunk][query][CAP] keys, L2 resident): both lanes (j, j + 32) of a query belong to the one
// wave that filters it, so the slot count is a REGISTER (kept equal in both lanes through one shuffle per append) and an append is a
// fire-and-forget global store -- no LDS atomic, no wait; compaction (a slot of 40 full: never in a seeded launch) and the final emit read
// the slot back past the vector L1.
struct KsCfg {
    static constexpr int NW = 8, NR = 8, NDW = 6, NIW = 4;   // ring: 4 tiles x 2 half-k chunks; six waves carry the 24 DMA pieces of a tile
    static constexpr int CAP = RMU_KS_CAP;
    static constexpr int RING_BYTES = NR * S_SLOT;
    static constexpr int XCH_OFF = RING_BYTES;               // 8 x 4 KiB: a wave's partial sums for its partner's queries
    static constexpr int GT_OFF = XCH_OFF + NW * 4096;
    static constexpr int LDS_BYTES = GT_OFF + NW * 256;
};
static_assert(KsCfg::LDS_BYTES <= 160 * 1024, "LDS");

template <int EXP = 0>
__global__ __launch_bounds__(512) void scan_screen_ks_kernel(const ScanLaunch a) {
    using C = KsCfg;
    constexpr bool DBG = (EXP & 4) != 0;
    constexpr int PRE = 4;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, j = lane & 31;
    const int pr = w & 3, sh = w >> 2;                    // query pair of the workgroup's four; k half
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
    const char* img = (const char*)a.x + a.row0 * (int64_t)IMGB;
    char* ring = ssm;
    const int q_own = qt * 256 + 64 * pr + 32 * sh;       // lanes j and j + 32 filter query q_own + j (group 0 of this wave's MFMAs) ...
    const int q_oth = qt * 256 + 64 * pr + 32 * (1 - sh); // ... and group 1 are the partner's queries
    const bool q_ok = q_own + j < a.nq;
    float thr_loc = q_ok ? -INFINITY : INFINITY, thr_g = -INFINITY, thr_s = thr_loc;
    u32* gthr_w = a.gthr + q_own;
    const u32* gt_lds = (const u32*)(ssm + C::GT_OFF) + w * 64;
    u32 cnt = 0;                                          // entries in this lane's query slot (equal in lanes j and j + 32)
    u64* const gslot = a.gcand + ((size_t)s_idx * a.nq + (q_ok ? q_own + j : 0)) * C::CAP;
    // sibling pacing: as in scan_screen_kernel (the last wave carries no corpus DMA)
    const bool pace_on = a.prog != nullptr;
    u32* prog_w = a.prog + (size_t)s_idx * 4;
    bool pace_live = pace_on;
    const u32* gsrc = gthr_w + j;
    if (pace_on && lane >= 32 && lane < 36) gsrc = prog_w + (lane - 32);
    auto refresh_gthr = [&]() {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                         (__attribute__((address_space(3))) void*)(ssm + C::GT_OFF + w * 256), 4, 0, 16);
    };
    constexpr int PW = C::NW - 1;
    auto pace_step = [&](int tile) {
        if (lane == 0) __hip_atomic_store(prog_w + qt, ~(u32)tile, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const u32 m4 = max(max(gt_lds[32], gt_lds[33]), max(gt_lds[34], gt_lds[35]));
        int lead = m4 ? tile - (int)~m4 : -1;
        if (__builtin_expect(__builtin_amdgcn_readfirstlane(lead) > a.pace, 0)) {
            int spins = 0;
            do {
                __builtin_amdgcn_s_sleep(24);
                u32 v = 0;
                if (lane < 4) v = __hip_atomic_load(prog_w + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                v = max(v, (u32)__shfl_xor((int)v, 1));
                v = max(v, (u32)__shfl_xor((int)v, 2));
                const u32 vm = (u32)__builtin_amdgcn_readfirstlane((int)v);
                lead = vm ? tile - (int)~vm : -1;
            } while (lead > a.pace && ++spins < 400);
            if (spins >= 400) pace_live = false;
        }
    };

    // ---- query fragments: this wave's k half (192 of 384) of both groups; step T covers k [192 sh + 16 T, + 16), lane half h owns 8 of them
    f16x8 qh[2][S_CS];
    {
        const int qa = q_ok ? q_own + j : 0, qb = q_oth + j < a.nq ? q_oth + j : 0;
        const char* ra = (const char*)a.q + (size_t)qa * IMGB + sh * S_CKB + h * 16;
        const char* rb_ = (const char*)a.q + (size_t)qb * IMGB + sh * S_CKB + h * 16;
#pragma unroll
        for (int T = 0; T < S_CS; ++T) {
            qh[0][T] = *(const f16x8*)(ra + T * 32);
            qh[1][T] = *(const f16x8*)(rb_ + T * 32);
        }
#pragma unroll
        for (int T = 0; T < S_CS; ++T) asm volatile("" : "+v"(qh[0][T]), "+v"(qh[1][T]));   // complete before any LDS-DMA (see scan_screen_kernel)
    }
    // ---- DMA map (six waves, four 1-KiB pieces of a tile each): piece id = n * 6 + w in 0..23 = 12 * chunk + piece of the chunk; the slot
    // layout and its swizzle are the ones of scan_screen_kernel
    u32 dma_off[C::NIW];
    int dma_dst[C::NIW];
#pragma unroll
    for (int n = 0; n < C::NIW; ++n) {
        const int id = n * C::NDW + (w < C::NDW ? w : 0);
        const int half = id / 12, pid = id % 12;
        const int f = pid * 64 + lane;
        const int i = f / S_U16, p = f % S_U16;
        dma_off[n] = (u32)(i * IMGB + (p ^ ((i >> 1) & 7)) * 16 + half * S_CKB);
        dma_dst[n] = half * S_SLOT + pid * 1024;
    }
    auto issue_piece = [&](int tl, int n) {               // piece n of this wave, tile tl of the workgroup's chunk
        if (EXP & 1) return;
        if (w >= C::NDW) return;                          // (uniform)
        const int te = tl < ntiles ? tl : ntiles - 1;     // past the end: a harmless reload keeps the vmcnt arithmetic uniform
        const char* sbase = img + ((t0 + te) * S_RT) * (int64_t)IMGB;
        char* dst = ring + ((2 * tl) & (C::NR - 1)) * S_SLOT + dma_dst[n];
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(sbase + dma_off[n]),
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    };
    int abase[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) abase[m] = j * S_CKB + (((2 * m + h) ^ ((j >> 1) & 7)) * 16);
    f16x8 fr[PRE];
#pragma unroll
    for (int m = 0; m < PRE; ++m) fr[m] = f16x8{};
    const u32 ring_addr = lds_addr(ring);
    auto read_frag = [&](f16x8& dst, int slot_off, int t) {
        if (EXP & 2) { asm volatile("" : "+v"(dst)); return; }
        const u32 addr = ring_addr + (u32)(abase[t & 3] + slot_off);
        if ((t >> 2) == 0) asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr));
        else if ((t >> 2) == 1) asm volatile("ds_read_b128 %0, %1 offset:128" : "=v"(dst) : "v"(addr));
        else asm volatile("ds_read_b128 %0, %1 offset:256" : "=v"(dst) : "v"(addr));
    };
    const u32 xw_addr = lds_addr(ssm + C::XCH_OFF + w * 4096) + lane * 16u;            // mine to write
    const u32 xr_addr = lds_addr(ssm + C::XCH_OFF + (w ^ 4) * 4096) + lane * 16u;      // my partner's to read
    auto xch_write = [&](const f32x16& v) {
        const f32x4 p0 = {v[0], v[1], v[2], v[3]}, p1 = {v[4], v[5], v[6], v[7]}, p2 = {v[8], v[9], v[10], v[11]}, p3 = {v[12], v[13], v[14], v[15]};
        asm volatile("s_nop 7\n\ts_nop 7\n\tds_write_b128 %0, %1\n\tds_write_b128 %0, %2 offset:1024\n\tds_write_b128 %0, %3 offset:2048\n\t"
                     "ds_write_b128 %0, %4 offset:3072\n\ts_nop 1" ::"v"(xw_addr), "v"(p0), "v"(p1), "v"(p2), "v"(p3));
    };
    f32x4 xr[4];
    auto xch_read = [&]() {
        asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %4 offset:2048\n\tds_read_b128 %3, %4 offset:3072"
                     : "=v"(xr[0]), "=v"(xr[1]), "=v"(xr[2]), "=v"(xr[3]) : "v"(xr_addr));
    };
    auto set_thr = [&]() { thr_s = fmaxf(thr_loc, thr_g) * 4096.0f; };

    u32 d_slow = 0, d_comp = 0, d_app = 0;
    unsigned long long d_clk_slow = 0, d_clk_bar = 0, d_clk_b2 = 0, d_clk_vm = 0, d_clk_all = DBG ? clock64() : 0;
    // keep the best K' of query lane jj's slot (sorted), raise its threshold, publish it: the whole wave works on one slot
    auto compact = [&](int jj) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                       // this wave's appends are in L2
        const u32 n = (u32)__builtin_amdgcn_readlane((int)cnt, jj);
        u64* slot = (u64*)(((u64)(u32)__builtin_amdgcn_readlane((int)(u32)((u64)gslot >> 32), jj) << 32) |
                           (u64)(u32)__builtin_amdgcn_readlane((int)(u32)(u64)gslot, jj));
        u64 key[1];
        u32 rank[1];
        // (every VMEM operation of the slow path is inline asm: one the compiler can see puts an s_waitcnt vmcnt(0) in front of the tile loop's
        // first MFMA -- the join of this path -- and drains the DMA ring once per tile)
        key[0] = 0ull;
        if ((u32)lane < n) asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(key[0]) : "v"(slot + lane) : "memory");
        rank_keys<1>(key, n, rank);
        const bool keep = (u32)lane < n && rank[0] < (u32)a.k;
        if (keep) asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(slot + rank[0]), "v"(key[0]) : "memory");
        const u64 kb = __ballot(keep && rank[0] == (u32)(a.k - 1));
        if (kb) {
            const int src = __builtin_ctzll(kb);
            const u32 hi = (u32)__builtin_amdgcn_readlane((int)(u32)(key[0] >> 32), src);
            if (j == jj) thr_loc = rmu_ord2f(hi);
            if (lane == 0) asm volatile("global_atomic_umax %0, %1, off sc1" ::"v"(gthr_w + jj), "v"(hi) : "memory");
        }
        if (j == jj) cnt = n < (u32)a.k ? n : (u32)a.k;
        set_thr();
        if (DBG) ++d_comp;
    };
    // append the passing scores of one tile: pf = 4096 * s~ of rows rbase + (r & 3) + 8 (r >> 2), `inmask` = slots inside the range
    auto slow_path = [&](const float (&pf)[16], int64_t rbase, u32 inmask) {
        unsigned long long c0 = 0;
        u32 todo = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) todo |= (pf[r] > thr_s) ? (1u << r) : 0u;
        todo &= inmask;
        if (DBG) { ++d_slow; d_app += __builtin_popcount(todo); c0 = clock64(); }
        u32 uni = 0;
        for (u64 bl = __ballot(todo != 0); bl; bl &= bl - 1) uni |= (u32)__builtin_amdgcn_readlane((int)todo, __builtin_ctzll(bl));
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if ((uni >> r) & 1u) {                                              // scalar test: most slots are skipped
                bool has = ((todo >> r) & 1u) && pf[r] > thr_s;                 // (a compaction in this call may have raised the threshold)
                u32 other = (u32)__shfl_xor((int)has, 32);
                u64 full = __ballot(cnt + (u32)has + other > (u32)C::CAP);
                if (__builtin_expect(full != 0, 0)) {
                    for (u32 fm = (u32)full | (u32)(full >> 32); fm; fm &= fm - 1) compact(__builtin_ctz(fm));
                    has = has && pf[r] > thr_s;
                    other = (u32)__shfl_xor((int)has, 32);
                }
                const u32 pos = cnt + (h ? other : 0u);
                if (has) {
                    // (inline asm: a store the compiler can see gets an s_waitcnt vmcnt(0) in front of the next write of its data registers --
                    // the first MFMA of the tile loop, where it drains the DMA ring once per trip; a 64-bit store has read its data when it issues)
                    const u64 key = rmu_make_key(pf[r] * (1.0f / 4096.0f) + 0.0f, (u32)(rbase + (r & 3) + 8 * (r >> 2)));
                    asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(gslot + pos), "v"(key) : "memory");
                }
                cnt += (u32)has + other;
            }
        }
        if (DBG) d_clk_slow += clock64() - c0;
    };

    constexpr bool PP = (EXP & 16) != 0;
    if constexpr (PP) {
        // ---- PING-PONG over the K split: the two waves of a SIMD are also the two k halves, and they never do the same thing at the same time.
        // X = waves 0-3 (k half 0), Y = waves 4-7 (k half 1).  A wave alternates a LOAD segment -- the 12 fragments of its half-k chunk into 48
        // registers, the exchange (write its partner's partial sums of the tile it has just computed, read what the partner left), 16 sums +
        // compares, thresholds -- with a COMPUTE segment of 24 back-to-back MFMAs with its DMA pieces in their shadow; X loads tile t in phase 2t
        // and computes it in phase 2t + 1, Y one phase later; one s_barrier per phase.  The load segment moves 16 KiB through a SIMD's 32 B/clk
        // LDS return path (~600 cycles) under the partner's 768 cycles of MFMA: the matrix pipe is the longer leg of every phase.
        //   ring: chunk (t, half) is read by ONE group in ONE phase; its slot is refilled (tile t + 4) by the same waves in the compute segment
        //   that follows -- no other wave ever touches it.
        //   exchange: X writes its partial of tile t - 1 for Y's queries in load(t) (phase 2t), Y reads it in its load(t) (phase 2t + 1) and
        //   filters tile t - 1 (own partial: the tile it computed last); Y writes in phase 2t + 1, X reads in load(t + 1) and filters tile t - 1
        //   as well -- by then it has computed tile t on top, so its own partials alternate between two accumulators.
        //   The loop runs two trips past the last tile (reads of a ring that is not refilled, results masked) so that every tile is filtered.
        if (ntiles > 0) {
            const int wi = w & 3;
            u32 poff[3];
            int pdst[3];
#pragma unroll
            for (int n = 0; n < 3; ++n) {
                const int pid = n * 4 + wi;
                const int f = pid * 64 + lane;
                const int i = f / S_U16, p = f % S_U16;
                poff[n] = (u32)(i * IMGB + (p ^ ((i >> 1) & 7)) * 16 + sh * S_CKB);
                pdst[n] = sh * S_SLOT + pid * 1024;
            }
            auto issue_pp = [&](int tl, int n) {
                const int te = tl < ntiles ? tl : ntiles - 1;
                const char* sbase = img + ((t0 + te) * S_RT) * (int64_t)IMGB;
                char* dst = ring + ((2 * tl) & (C::NR - 1)) * S_SLOT + pdst[n];
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(sbase + poff[n]),
                                                 (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
            };
            // VMEM order of a wave: [PPP R] per tile (prologue: tiles 0-3, then tile it + 4 in compute(it)).  Behind compute(it) the pieces of tile
            // it + 1 must have landed: 13 younger operations may still be in flight (R, PPP R, PPP R, PPP R); the same count holds for tile 0 here
#pragma unroll
            for (int tl = 0; tl < 4; ++tl) {
#pragma unroll
                for (int n = 0; n < 3; ++n) issue_pp(tl, n);
                refresh_gthr();
            }
            asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (sh) __builtin_amdgcn_s_barrier();          // Y runs one phase behind X
            f16x8 fq[S_CS];
            f32x16 oA, oB, c1;
#pragma unroll
            for (int r = 0; r < 16; ++r) { oA[r] = -INFINITY; oB[r] = -INFINITY; c1[r] = 0.f; }
            const int64_t lane_r0 = a.row0 + t0 * S_RT + 4 * h;
            const int64_t row_end = a.row0 + a.n_rows;
            u32 lastmask = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) lastmask |= (lane_r0 + (int64_t)(ntiles - 1) * S_RT + (r & 3) + 8 * (r >> 2) < row_end) ? (1u << r) : 0u;
            const int lag = sh ? 1 : 2;
            // one trip: LOAD(it) | barrier | COMPUTE(it) | barrier.  cw = the own-partial accumulator this trip's MFMAs overwrite, co = the other
            auto trip = [&](f32x16& cw, const f32x16& co, int it) {
                unsigned long long ck0 = 0, ck1 = 0, ck2 = 0, ck3 = 0;
                if (DBG) ck0 = clock64();
                // ---- LOAD: exchange first (the partner's partials are what the filter waits for), then the 12 fragments with the 16 sums and ONE
                // compare between them -- the LDS queue, not the issue port, paces this segment, so the VALU work rides in its stalls
                xch_read();
                xch_write(c1);
                const int soff = (((2 * it) & (C::NR - 1)) + sh) * S_SLOT;
#pragma unroll
                for (int t = 0; t < 4; ++t) read_frag(fq[t], soff, t);
                asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(xr[0]), "+v"(xr[1]), "+v"(xr[2]), "+v"(xr[3]));
                float pf[16];
                if (sh) {                                  // Y: its own partials of tile it - 1 are the ones it computed last
#pragma unroll
                    for (int t = 4; t < S_CS; ++t) {
                        read_frag(fq[t], soff, t);
                        pf[2 * (t - 4)] = co[2 * (t - 4)] + xr[(t - 4) >> 1][(2 * (t - 4)) & 3];
                        pf[2 * (t - 4) + 1] = co[2 * (t - 4) + 1] + xr[(t - 4) >> 1][(2 * (t - 4) + 1) & 3];
                        __builtin_amdgcn_sched_barrier(0);
                    }
                } else {                                   // X: of tile it - 2, in the accumulator it is about to overwrite
#pragma unroll
                    for (int t = 4; t < S_CS; ++t) {
                        read_frag(fq[t], soff, t);
                        pf[2 * (t - 4)] = cw[2 * (t - 4)] + xr[(t - 4) >> 1][(2 * (t - 4)) & 3];
                        pf[2 * (t - 4) + 1] = cw[2 * (t - 4) + 1] + xr[(t - 4) >> 1][(2 * (t - 4) + 1) & 3];
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                float mx = pf[0];                          // (v_max3: a NaN sum never wins, exactly as it never passes a compare)
#pragma unroll
                for (int r = 1; r < 15; r += 2) mx = fmaxf(mx, fmaxf(pf[r], pf[r + 1]));
                mx = fmaxf(mx, pf[15]);
                const int ft = it - lag;                   // the tile whose scores are complete now
                if (!(a.share_thr & 2) && !(EXP & 8) && __builtin_expect(__ballot(mx > thr_s) != 0, 0)) {
                    const u32 inmask = (ft < 0 || ft >= ntiles) ? 0u : (ft == ntiles - 1 ? lastmask : 0xffffu);
                    if (inmask) slow_path(pf, lane_r0 + (int64_t)ft * S_RT, inmask);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int t = 0; t < S_CS; ++t) asm volatile("" : "+v"(fq[t]));
                if (DBG) ck1 = clock64();
                __builtin_amdgcn_sched_barrier(0);         // (no MFMA of the compute segment above the barrier)
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                if (DBG) ck2 = clock64();
                // ---- COMPUTE: 24 MFMAs, this wave's three DMA pieces of tile it + 4 and the threshold refresh in their shadow
#pragma unroll
                for (int t = 0; t < S_CS; ++t) {
                    if (t == 0) {
                        const f32x16 z = {};
                        cw = __builtin_amdgcn_mfma_f32_32x32x16_f16(fq[t], qh[0][t], z, 0, 0, 0);
                        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fq[t], qh[1][t], z, 0, 0, 0);
                    } else {
                        cw = __builtin_amdgcn_mfma_f32_32x32x16_f16(fq[t], qh[0][t], cw, 0, 0, 0);
                        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fq[t], qh[1][t], c1, 0, 0, 0);
                    }
                    if (t == 1 || t == 4 || t == 7) issue_pp(it + 4, (t - 1) / 3);
                    if (t == 9) refresh_gthr();
                    __builtin_amdgcn_sched_barrier(0);
                }
                {   // thresholds for the next load segment (the refresh of a trip ago has landed: counted wait below covers the NEXT one)
                    const u32 go = gt_lds[j];
                    thr_g = (go && (a.share_thr & 1)) ? rmu_ord2f(go - 1u) : -INFINITY;
                    set_thr();
                    if (w == PW && pace_live && it + 1 < ntiles) pace_step(it + 1);
                }
                asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
                if (DBG) { asm volatile("" : "+v"(cw), "+v"(c1)); ck3 = clock64(); }
                __builtin_amdgcn_s_barrier();
                if (DBG) { d_clk_vm += ck1 - ck0; d_clk_bar += ck2 - ck1; d_clk_b2 += ck3 - ck2; d_clk_slow += clock64() - ck3; }
            };
            const int nit = ntiles + 2;
            for (int it = 0; it < nit; it += 2) {
                trip(oA, oB, it);
                if (it + 1 < nit) trip(oB, oA, it + 1);
            }
            if (!sh) __builtin_amdgcn_s_barrier();         // (X started one phase early)
            if (pace_on && w == PW && lane == 0) __hip_atomic_store(prog_w + qt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        }
    } else
    if (ntiles > 0) {
        refresh_gthr();                                    // oldest VMEM op: seeded / already published thresholds
#pragma unroll
        for (int tl = 0; tl < 3; ++tl)
#pragma unroll
            for (int n = 0; n < C::NIW; ++n) issue_piece(tl, n);
        if (w < C::NDW) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");     // tiles 0 and 1
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int m = 0; m < PRE; ++m) read_frag(fr[m], sh * S_SLOT, m);
        f32x16 accA0, accA1, accB0, accB1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { accB0[r] = -INFINITY; accB1[r] = 0.f; accA0[r] = 0.f; accA1[r] = 0.f; }
        const int64_t lane_r0 = a.row0 + t0 * S_RT + 4 * h;
        auto rb = [&](int t) { return lane_r0 + (int64_t)t * S_RT; };
        // one tile: 12 steps of two MFMAs (own group into c0, the partner's into c1); p0 / p1 = the previous tile's own / send partials
        auto tile_body = [&](f32x16& c0, f32x16& c1, const f32x16& p0, const f32x16& p1, int tl) {
            unsigned long long cb = 0;
            if (DBG) cb = clock64();
            // in flight at most: this wave's ops of the previous tile (4 pieces of tile tl + 2 and a threshold refresh | the refresh)
            if (w < C::NDW) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
            if (DBG) { const unsigned long long cv = clock64(); d_clk_vm += cv - cb; cb = cv; }
            __builtin_amdgcn_s_barrier();                  // A: tile tl + 1 has landed; nobody reads tile tl - 1 or the exchange of tl - 2 any more
            if (DBG) d_clk_bar += clock64() - cb;
            {
                const u32 go = gt_lds[j];
                thr_g = (go && (a.share_thr & 1)) ? rmu_ord2f(go - 1u) : -INFINITY;
                set_thr();
                if (w == PW && pace_live) pace_step(tl);
            }
            xch_write(p1);
            const int cur_off = (((2 * tl) & (C::NR - 1)) + sh) * S_SLOT, nxt_off = (((2 * tl + 2) & (C::NR - 1)) + sh) * S_SLOT;
            float pf[16];
            u64 any_pass = 0;
#pragma unroll
            for (int t = 0; t < S_CS; ++t) {
                if (t == 5) {
                    unsigned long long cb2 = 0;
                    if (DBG) cb2 = clock64();
                    __builtin_amdgcn_s_barrier();          // B: the exchange writes of this tile (waited for at step 4) are visible
                    if (DBG) d_clk_b2 += clock64() - cb2;
                    xch_read();
                }
                if (!(EXP & 2)) {
                    if (t == 9) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(fr[t % PRE]), "+v"(xr[0]), "+v"(xr[1]), "+v"(xr[2]), "+v"(xr[3]));
                    else if (t == 4 || t > 9) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(fr[t % PRE]));
                    else asm volatile("s_waitcnt lgkmcnt(7)" : "+v"(fr[t % PRE]));
                }
                if (t == 0) {
                    const f32x16 z = {};
                    c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[t % PRE], qh[0][t], z, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[t % PRE], qh[1][t], z, 0, 0, 0);
                } else {
                    c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[t % PRE], qh[0][t], c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[t % PRE], qh[1][t], c1, 0, 0, 0);
                }
                if (t + PRE < S_CS) read_frag(fr[t % PRE], cur_off, t + PRE);
                else read_frag(fr[t % PRE], nxt_off, t + PRE - S_CS);
                if (t % 3 == 1) issue_piece(tl + 3, t / 3);                      // steps 1, 4, 7, 10
                if (t == 3) refresh_gthr();
                __builtin_amdgcn_sched_barrier(0);         // the scheduler otherwise bunches the counted waits of two or three steps in front of their MFMAs
                if (t >= 9 && !(EXP & 8)) {                                     // 16 sums + compares over three steps
                    constexpr int NPS = 6;
#pragma unroll
                    for (int r = (t - 9) * NPS; r < (t - 9) * NPS + NPS && r < 16; ++r) {
                        pf[r] = p0[r] + xr[r >> 2][r & 3];
                        any_pass |= __ballot(pf[r] > thr_s);
                    }
                }
            }
            if (!(a.share_thr & 2) && __builtin_expect(any_pass != 0, 0)) slow_path(pf, rb(tl - 1), 0xffffu);
        };
        for (int tl = 0; tl < ntiles; tl += 2) {           // two copies of the body: accumulator parity
            tile_body(accA0, accA1, accB0, accB1, tl);        // (tile -1 = the -inf accumulators: nothing passes, whatever the exchange area holds)
            if (tl + 1 < ntiles) tile_body(accB0, accB1, accA0, accA1, tl + 1);
        }
        if (pace_on && w == PW && lane == 0) __hip_atomic_store(prog_w + qt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int m = 0; m < PRE; ++m) asm volatile("" : "+v"(fr[m]));
        {   // the last tile: exchange, sum, mask the rows past the range, filter
            const bool last_in_a = ((ntiles - 1) & 1) == 0;
            f32x16 own, snd;
#pragma unroll
            for (int r = 0; r < 16; ++r) { own[r] = last_in_a ? accA0[r] : accB0[r]; snd[r] = last_in_a ? accA1[r] : accB1[r]; }
            __builtin_amdgcn_s_barrier();                  // everybody has read the exchange of the tile before
            xch_write(snd);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            xch_read();
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xr[0]), "+v"(xr[1]), "+v"(xr[2]), "+v"(xr[3]));
            const int64_t rbl = rb(ntiles - 1), row_end = a.row0 + a.n_rows;
            float pf[16];
            u32 inmask = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                pf[r] = own[r] + xr[r >> 2][r & 3];
                inmask |= (rbl + (r & 3) + 8 * (r >> 2) < row_end) ? (1u << r) : 0u;
            }
            if (!(a.share_thr & 2)) slow_path(pf, rbl, inmask);
        }
    }
    if (DBG) {
        u32 app = d_app;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) app += __shfl_xor(app, o);
        if (lane == 0) {
            atomicAdd((unsigned long long*)a.dbg + 0, (unsigned long long)d_slow);
            atomicAdd((unsigned long long*)a.dbg + 1, (unsigned long long)d_comp);
            atomicAdd((unsigned long long*)a.dbg + 2, (unsigned long long)app);
            atomicAdd((unsigned long long*)a.dbg + 3, (unsigned long long)ntiles);
            atomicAdd((unsigned long long*)a.dbg + 4, d_clk_b2);      // ("rounds" in the dump: cycles in barrier B)
            atomicAdd((unsigned long long*)a.dbg + 5, d_clk_slow);
            atomicAdd((unsigned long long*)a.dbg + 6, d_clk_bar);
            atomicAdd((unsigned long long*)a.dbg + 8, d_clk_vm);
            atomicAdd((unsigned long long*)a.dbg + 7, (unsigned long long)(clock64() - d_clk_all));
        }
    }
    // ---- emit: best K' approximate candidates of this (chunk, query), sorted.  Eight slots are read back per round trip.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int part = s_idx;
    for (int j0 = 0; j0 < 32; j0 += 8) {
        if (q_own + j0 >= a.nq) break;
        u64 key[8][1];
        u32 nn[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int jj = j0 + e;
            nn[e] = (u32)__builtin_amdgcn_readlane((int)cnt, jj);
            const u64* slot = (const u64*)(((u64)(u32)__builtin_amdgcn_readlane((int)(u32)((u64)gslot >> 32), jj) << 32) |
                                           (u64)(u32)__builtin_amdgcn_readlane((int)(u32)(u64)gslot, jj));
            key[e][0] = (u32)lane < nn[e] ? __hip_atomic_load(slot + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int qq = q_own + j0 + e;
            if (qq < a.nq) {
                u32 rank[1];
                rank_keys<1>(key[e], nn[e], rank);
                u64* dst = a.partial + ((size_t)part * a.nq + qq) * a.k;
                if ((u32)lane < nn[e]) {
                    if (rank[0] < (u32)a.k) dst[rank[0]] = key[e][0];
                } else if (lane < a.k) {
                    dst[lane] = 0ull;
                }
            }
        }
    }
}

// ---- 128 QUERIES PER WAVE (round 4; batches that fill 512-query tiles) -------------------------------------------------------------------
// Both forms above are bound by what a 1-KiB A fragment costs to fetch from the LDS (~100 B/clk per CU, DMA writes included) against the one
// or two 32-cycle MFMAs it feeds; the K-split pays for its second MFMA with an exchange and ends where it started.  The lever that is left
// is register blocking: ONE wave per SIMD with the whole 512-register file (VGPRs + AGPRs), query fragments of FOUR 32-query groups in 384
// of them, every fragment feeding four MFMAs (128 cycles of matrix work per ds_read_b128; 120 KiB of LDS traffic per 32 rows x 512 queries
// instead of 2 x 216).  What made the 4-wave G = 2 form issue-bound -- ~10 other instructions per MFMA pair -- is ~5 per FOUR MFMAs here:
//   - no second accumulator set: group g of the previous tile is filtered (v_max3 tree over its 16 scores, ONE compare) right before the
//     first MFMA of the new tile overwrites it (zero C operand), in the shadow of group g - 1's MFMA;
//   - candidates in global memory with the slot counts in registers, as in the K-split kernel (512 queries x 40 x 8 B do not fit the LDS).
// 512 queries per workgroup: 1024 queries are 2 query tiles, so each image byte also crosses the L2 -> LDS path half as often.
struct G4Cfg {
    static constexpr int NW = 4, G = 4, QW = 128, NR = 5, NIW = 3;   // five 12-KiB half-k ring slots; three DMA pieces per wave and chunk
    static constexpr int GR = 3;                             // groups whose query fragments live in registers (288); the fourth group's 24 KiB
                                                             // per wave sit in the LDS and are streamed one step ahead like the row fragments:
                                                             // 384 + 64 accumulator registers leave the allocator no room (72 spills into the loop)
    static constexpr int CAP = RMU_KS_CAP;
    static constexpr int RING_BYTES = NR * S_SLOT;
    static constexpr int QL_OFF = RING_BYTES;                // [wave][step][lane] 16 B
    static constexpr int QL_WAVE = S_TS * 1024;
    static constexpr int GT_OFF = QL_OFF + NW * QL_WAVE;
    static constexpr int LDS_BYTES = GT_OFF + NW * 512;
};
static_assert(G4Cfg::LDS_BYTES <= 160 * 1024, "LDS");

struct G4Slow { u32 cnt; float thr_s; };
// The append path of scan_screen_g4_kernel as a REAL call: inlined eight times into a kernel that keeps ~490 registers live it pushed the
// allocator into spilling inside the tile loop; behind a call the saves and restores sit at the (rare) call site.
// sc = 4096 * s~ of rows rbase + (r & 3) + 8 (r >> 2) for this lane's query; inmask = accumulator slots inside the row range; cn = entries in
// the query's slot gs (equal in lanes j and j + 32), ts = 4096 * threshold.  Returns the new count and threshold.
__device__ __attribute__((noinline)) G4Slow g4_slow(f32x16 sc, int64_t rbase, u32 inmask, u32 cn, float ts, u64* gs, u32* gthr_g, int k) {
    const int lane = threadIdx.x & 63, h = lane >> 5, j = lane & 31;
    u32 todo = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) todo |= (sc[r] > ts) ? (1u << r) : 0u;
    todo &= inmask;
    u32 uni = 0;
    for (u64 bl = __ballot(todo != 0); bl; bl &= bl - 1) uni |= (u32)__builtin_amdgcn_readlane((int)todo, __builtin_ctzll(bl));
    for (int r = 0; r < 16; ++r) {
        if (!((uni >> r) & 1u)) continue;                                   // (uniform)
        float v = sc[0];
#pragma unroll
        for (int e = 1; e < 16; ++e) v = (r == e) ? sc[e] : v;              // (uniform select: no dynamic register index)
        bool has = ((todo >> r) & 1u) && v > ts;
        u32 other = (u32)__shfl_xor((int)has, 32);
        const u64 full = __ballot(cn + (u32)has + other > (u32)G4Cfg::CAP);
        if (__builtin_expect(full != 0, 0)) {
            // keep the best K' of a full slot (sorted), raise its threshold, publish it: the whole wave works on one slot.  VMEM as inline asm
            // with explicit waits (see scan_screen_ks_kernel)
            for (u32 fm = (u32)full | (u32)(full >> 32); fm; fm &= fm - 1) {
                const int jj = __builtin_ctz(fm);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const u32 n = (u32)__builtin_amdgcn_readlane((int)cn, jj);
                u64* slot = (u64*)(((u64)(u32)__builtin_amdgcn_readlane((int)(u32)((u64)gs >> 32), jj) << 32) |
                                   (u64)(u32)__builtin_amdgcn_readlane((int)(u32)(u64)gs, jj));
                u64 key[1];
                u32 rank[1];
                key[0] = 0ull;
                if ((u32)lane < n) asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(key[0]) : "v"(slot + lane) : "memory");
                rank_keys<1>(key, n, rank);
                const bool keep = (u32)lane < n && rank[0] < (u32)k;
                if (keep) asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(slot + rank[0]), "v"(key[0]) : "memory");
                const u64 kb = __ballot(keep && rank[0] == (u32)(k - 1));
                if (kb) {
                    const u32 hi = (u32)__builtin_amdgcn_readlane((int)(u32)(key[0] >> 32), __builtin_ctzll(kb));
                    if (j == jj) ts = fmaxf(ts, rmu_ord2f(hi) * 4096.0f);
                    if (lane == 0) asm volatile("global_atomic_umax %0, %1, off sc1" ::"v"(gthr_g + jj), "v"(hi) : "memory");
                }
                if (j == jj) cn = n < (u32)k ? n : (u32)k;
            }
            has = has && v > ts;
            other = (u32)__shfl_xor((int)has, 32);
        }
        const u32 pos = cn + (h ? other : 0u);
        if (has) {
            const u64 key = rmu_make_key(v * (1.0f / 4096.0f) + 0.0f, (u32)(rbase + (r & 3) + 8 * (r >> 2)));
            asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(gs + pos), "v"(key) : "memory");
        }
        cn += (u32)has + other;
    }
    return G4Slow{cn, ts};
}

template <int EXP = 0>
__global__ __launch_bounds__(256) void scan_screen_g4_kernel(const ScanLaunch a) {
    using C = G4Cfg;
    constexpr bool DBG = (EXP & 4) != 0;
    constexpr int G = C::G, PRE = 3;                     // fragment buffers: this step's, the next one's (landed), the one after (in flight)
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
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
    const char* img = (const char*)a.x + a.row0 * (int64_t)IMGB;
    char* ring = ssm;
    const int q_base = (qt * C::NW + w) * C::QW;          // group g, lanes j and j + 32: query q_base + 32 g + j
    float thr_s[G];                                       // 4096 * max(own k-th best, shared threshold): only ever rises
    u32 cnt[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        thr_s[g] = (q_base + 32 * g + j < a.nq) ? -INFINITY : INFINITY;
        cnt[g] = 0;
    }
    u32* gthr_w = a.gthr + q_base;
    const u32* gt_lds = (const u32*)(ssm + C::GT_OFF) + w * 128;
    // (queries past nq never append: their slot address is never used)
    u64* const gslot0 = a.gcand + ((size_t)s_idx * a.nq + q_base + j) * C::CAP;
    constexpr size_t GSTRIDE = (size_t)32 * C::CAP;       // keys between the slots of two groups
    const u32* gsrc = gthr_w + lane;
    auto refresh_gthr = [&]() {                           // two 4-byte LDS-DMAs: this wave's 128 shared thresholds
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                         (__attribute__((address_space(3))) void*)(ssm + C::GT_OFF + w * 512), 4, 0, 16);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc + 64),
                                         (__attribute__((address_space(3))) void*)(ssm + C::GT_OFF + w * 512 + 256), 4, 0, 16);
    };
    f16x8 qh[C::GR][S_TS];
    const u32 ql_addr = lds_addr(ssm + C::QL_OFF + w * C::QL_WAVE) + lane * 16u;
    {   // the last group's fragments go to the LDS first (before any LDS-DMA is in flight: plain stores)
        const int qi = q_base + 32 * C::GR + j;
        const char* qrow = (const char*)a.q + (size_t)(qi < a.nq ? qi : 0) * IMGB + h * 16;
#pragma unroll
        for (int T = 0; T < S_TS; ++T) {
            const f16x8 v = *(const f16x8*)(qrow + T * 32);
            *(f16x8*)(ssm + C::QL_OFF + w * C::QL_WAVE + T * 1024 + lane * 16) = v;
        }
    }
#pragma unroll
    for (int g = 0; g < C::GR; ++g) {
        const int qi = q_base + 32 * g + j;
        const char* qrow = (const char*)a.q + (size_t)(qi < a.nq ? qi : 0) * IMGB + h * 16;
#pragma unroll
        for (int T = 0; T < S_TS; ++T) qh[g][T] = *(const f16x8*)(qrow + T * 32);
    }
#pragma unroll
    for (int g = 0; g < C::GR; ++g)
#pragma unroll
        for (int T = 0; T < S_TS; ++T) asm volatile("" : "+v"(qh[g][T]));    // complete before any LDS-DMA (see scan_screen_kernel)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (each wave reads back only what it wrote)
    u32 dma_off[C::NIW];
#pragma unroll
    for (int n = 0; n < C::NIW; ++n) {
        const int f = (n * C::NW + w) * 64 + lane;
        const int i = f / S_U16, p = f % S_U16;
        dma_off[n] = (u32)(i * IMGB + (p ^ ((i >> 1) & 7)) * 16);
    }
    const int nchunks = 2 * ntiles;
    auto issue_part = [&](int cc, int n) {
        if (EXP & 1) return;
        const int ce = cc < nchunks ? cc : nchunks - 1;
        const char* sbase = img + ((t0 + (ce >> 1)) * S_RT) * (int64_t)IMGB + (ce & 1) * S_CKB;
        char* slot = ring + (cc % C::NR) * S_SLOT;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(sbase + dma_off[n]),
                                         (__attribute__((address_space(3))) void*)(slot + (n * C::NW + w) * 1024), 16, 0, 0);
    };
    int abase[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) abase[m] = j * S_CKB + (((2 * m + h) ^ ((j >> 1) & 7)) * 16);
    f16x8 fr[PRE], qf[PRE];                              // row fragments and the LDS group's query fragments, PRE steps ahead
#pragma unroll
    for (int m = 0; m < PRE; ++m) { fr[m] = f16x8{}; qf[m] = f16x8{}; }
    const u32 ring_addr = lds_addr(ring);
    // step gs of the tile (0..23): its row fragment (chunk step t of the slot at slot_off) and the LDS group's query fragment
    auto read_frag = [&](f16x8& dst, f16x8& qdst, int slot_off, int t, int gs) {
        if (EXP & 2) { asm volatile("" : "+v"(dst), "+v"(qdst)); return; }
        const u32 addr = ring_addr + (u32)(abase[t & 3] + slot_off);
        if ((t >> 2) == 0) asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr));
        else if ((t >> 2) == 1) asm volatile("ds_read_b128 %0, %1 offset:128" : "=v"(dst) : "v"(addr));
        else asm volatile("ds_read_b128 %0, %1 offset:256" : "=v"(dst) : "v"(addr));
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(qdst) : "v"(ql_addr), "n"(gs * 1024));
    };
    // A step first issues the reads of the step two ahead (into the buffer the PREVIOUS step's MFMAs have finished reading), then waits for its
    // own pair: a single in-order wave per SIMD gets the full two steps = 256 cycles of MFMA issue between a read and its use (issued behind
    // the step's MFMAs, ~100 cycles later, the same reads cost 3 ms of the 8.7-ms batch in exposed LDS latency)
    auto frag_wait = [&](f16x8& f, f16x8& q) {           // two reads per step: the oldest pair has landed when 2 (PRE - 1) are in flight
        if (EXP & 2) return;
        asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(f), "+v"(q) : "n"(2 * (PRE - 1)));
    };
    u32 d_slow = 0, d_comp = 0, d_app = 0;
    unsigned long long d_clk_slow = 0, d_clk_bar = 0, d_clk_vm = 0, d_clk_all = DBG ? clock64() : 0;
    constexpr int GRP = C::NIW;
    constexpr int WAITN = GRP * (C::NR - 3);               // at a chunk's barrier only chunks >= cc + 2 may be in flight
    if (ntiles > 0) {
        refresh_gthr();
#pragma unroll
        for (int c0 = 0; c0 < C::NR - 1; ++c0)
#pragma unroll
            for (int n = 0; n < C::NIW; ++n) issue_part(c0, n);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GRP * (C::NR - 2)) : "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int m = 0; m < PRE - 1; ++m) read_frag(fr[m], qf[m], 0, m, m);
        f32x16 acc[G];
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[g][r] = -INFINITY;
        const int64_t lane_r0 = a.row0 + t0 * S_RT + 4 * h;
        int cc = 0;
        // The MFMAs are inline asm with register-class constraints: the accumulators (64) and the query fragments of group 0 and half of group 1 (144) sit
        // in AGPRs and are read by the matrix core in place; left to itself the allocator treats AGPRs as a spill area for VGPR values (2.6
        // v_accvgpr_read per MFMA, reloads behind s_waitcnt vmcnt(0) inside the tile loop).  Volatile asm keeps program order; an accumulator is
        // read by VALU (filter) only three MFMAs = 96+ cycles after the last MFMA that wrote it.
        auto mfma = [&](f32x16& c, const f16x8& fa, const f16x8& qb, int g, int gs) {
            // (group 1: the allocator parks some of its fragments in the other register class and copies them over right in front of the
            // MFMA -- v_accvgpr_write / _read -> MFMA source needs two wait states, which nobody inserts for inline asm: without the s_nop
            // 20 of 1024 queries, all of group 1, lost a neighbour)
            if (g == 0) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c) : "v"(fa), "a"(qb));
            else if (g == 1 && gs < S_CS) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c) : "v"(fa), "a"(qb));
            else if (g == 1) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c) : "v"(fa), "v"(qb));
            else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c) : "v"(fa), "v"(qb));
        };
        auto mfma0 = [&](f32x16& c, const f16x8& fa, const f16x8& qb, int g) {
            if (g < 2) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=a"(c) : "v"(fa), "a"(qb));
            else asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=a"(c) : "v"(fa), "v"(qb));
        };
        auto qfrag = [&](int g, int gs, const f16x8& streamed) -> const f16x8& { return g < C::GR ? qh[g < C::GR ? g : 0][gs] : streamed; };
        // one half-k chunk: ring barrier, then 12 steps of four MFMAs; CI::value = chunk parity (compile time: it selects the query fragments)
        auto chunk = [&](auto CI, int tl) {
            constexpr int c = decltype(CI)::value;
            unsigned long long cb = 0;
            if (DBG) cb = clock64();
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAITN) : "memory");
            if (DBG) { const unsigned long long cv = clock64(); d_clk_vm += cv - cb; cb = cv; }
            __builtin_amdgcn_s_barrier();
            if (DBG) d_clk_bar += clock64() - cb;
            if (c == 0) {
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const u32 go = gt_lds[32 * g + j];
                    if (go && (a.share_thr & 1)) thr_s[g] = fmaxf(thr_s[g], rmu_ord2f(go - 1u) * 4096.0f);
                }
            } else {
                refresh_gthr();
            }
            const int cur_off = (cc % C::NR) * S_SLOT, nxt_off = ((cc + 1) % C::NR) * S_SLOT;
            if (c == 0) {
                // step 0 (peeled: a full unroll of the step loop must not carry four copies of the slow path per step): each group's scores
                // of the previous tile are filtered just before its accumulator starts over
                read_frag(fr[(PRE - 1) % PRE], qf[(PRE - 1) % PRE], cur_off, PRE - 1, PRE - 1);
                frag_wait(fr[0], qf[0]);
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    if (!(EXP & 8)) {
                        float mx = acc[g][0];
#pragma unroll
                        for (int r = 1; r < 15; r += 2) mx = fmaxf(mx, fmaxf(acc[g][r], acc[g][r + 1]));
                        mx = fmaxf(mx, acc[g][15]);
                        if (!(a.share_thr & 2) && __builtin_expect(__ballot(mx > thr_s[g]) != 0, 0)) {
                            const G4Slow u = g4_slow(acc[g], lane_r0 + (int64_t)(tl - 1) * S_RT, 0xffffu, cnt[g], thr_s[g], gslot0 + g * GSTRIDE,
                                                     gthr_w + 32 * g, a.k);
                            cnt[g] = u.cnt; thr_s[g] = u.thr_s;
                            if (DBG) ++d_slow;
                        }
                    }
                    mfma0(acc[g], fr[0], qfrag(g, 0, qf[0]), g);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int t = (c == 0 ? 1 : 0); t < S_CS; ++t) {
                constexpr int base = c * S_CS;
                constexpr int AH = PRE - 1;                                  // steps ahead
                if (t + AH < S_CS) read_frag(fr[(base + t + AH) % PRE], qf[(base + t + AH) % PRE], cur_off, t + AH, (base + t + AH) % S_TS);
                else read_frag(fr[(base + t + AH) % PRE], qf[(base + t + AH) % PRE], nxt_off, t + AH - S_CS, (base + t + AH) % S_TS);
                frag_wait(fr[(base + t) % PRE], qf[(base + t) % PRE]);
#pragma unroll
                for (int g = 0; g < G; ++g) mfma(acc[g], fr[(base + t) % PRE], qfrag(g, base + t, qf[(base + t) % PRE]), g, base + t);
                if (t % 4 == 1) issue_part(cc + C::NR - 1, t / 4);          // steps 1, 5, 9: this wave's DMA pieces of the chunk
                __builtin_amdgcn_sched_barrier(0);
            }
            ++cc;
        };
        for (int tl = 0; tl < ntiles; ++tl) {
            chunk(std::integral_constant<int, 0>{}, tl);
            chunk(std::integral_constant<int, 1>{}, tl);
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int m = 0; m < PRE; ++m) asm volatile("" : "+v"(fr[m]), "+v"(qf[m]));
        {   // the last tile: rows past the range are masked
            const int64_t rbl = lane_r0 + (int64_t)(ntiles - 1) * S_RT, row_end = a.row0 + a.n_rows;
            u32 inmask = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) inmask |= (rbl + (r & 3) + 8 * (r >> 2) < row_end) ? (1u << r) : 0u;
            if (!(a.share_thr & 2)) {
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const G4Slow u = g4_slow(acc[g], rbl, inmask, cnt[g], thr_s[g], gslot0 + g * GSTRIDE, gthr_w + 32 * g, a.k);
                    cnt[g] = u.cnt; thr_s[g] = u.thr_s;
                }
            }
        }
    }
    if (DBG) {
        u32 app = d_app;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) app += __shfl_xor(app, o);
        if (lane == 0) {
            atomicAdd((unsigned long long*)a.dbg + 0, (unsigned long long)d_slow);
            atomicAdd((unsigned long long*)a.dbg + 1, (unsigned long long)d_comp);
            atomicAdd((unsigned long long*)a.dbg + 2, (unsigned long long)app);
            atomicAdd((unsigned long long*)a.dbg + 3, (unsigned long long)ntiles);
            atomicAdd((unsigned long long*)a.dbg + 5, d_clk_slow);
            atomicAdd((unsigned long long*)a.dbg + 6, d_clk_bar);
            atomicAdd((unsigned long long*)a.dbg + 8, d_clk_vm);
            atomicAdd((unsigned long long*)a.dbg + 7, (unsigned long long)(clock64() - d_clk_all));
        }
    }
    // ---- emit: best K' approximate candidates of this (chunk, query), sorted.  Eight slots are read back per round trip.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int part = s_idx;
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const u64* gs = gslot0 + g * GSTRIDE;
        for (int j0 = 0; j0 < 32; j0 += 8) {
            if (q_base + 32 * g + j0 >= a.nq) break;
            u64 key[8][1];
            u32 nn[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int jj = j0 + e;
                nn[e] = (u32)__builtin_amdgcn_readlane((int)cnt[g], jj);
                const u64* slot = (const u64*)(((u64)(u32)__builtin_amdgcn_readlane((int)(u32)((u64)gs >> 32), jj) << 32) |
                                               (u64)(u32)__builtin_amdgcn_readlane((int)(u32)(u64)gs, jj));
                key[e][0] = (u32)lane < nn[e] ? __hip_atomic_load(slot + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int qq = q_base + 32 * g + j0 + e;
                if (qq < a.nq) {
                    u32 rank[1];
                    rank_keys<1>(key[e], nn[e], rank);
                    u64* dst = a.partial + ((size_t)part * a.nq + qq) * a.k;
                    if ((u32)lane < nn[e]) {
                        if (rank[0] < (u32)a.k) dst[rank[0]] = key[e][0];
                    } else if (lane < a.k) {
                        dst[lane] = 0ull;
                    }
                }
            }
        }
    }
}

// exact fp32 re-score of the K' candidates of each query, in the exact kernel's summation order:
// for t in 0..47, c in 0..3: acc = fma(x[8t+c], q[8t+c], acc); acc = fma(x[8t+4+c], q[8t+4+c], acc)
// (v_mfma_f32_32x32x2_f32 is a k-ordered fmaf chain; lanes < 32 hold k = 8t+c, lanes >= 32 hold k = 8t+4+c).
// L2 (RMU_METRIC_L2SQ; rows and queries `stride` = 768 floats apart, rows carry -|x|^2 in column 384, queries are (2q, 1): rmu_api.hip): the exact
// scan's chain runs on over the pad columns -- exact zeros except k = 384, the first term of step t = 48: S = fma(-|x|^2, 1, chain(2q, x)); the
// reported score is max(|q|^2 - S, 0) as in the exact path's merge (topk_merge.hip).  The candidates' approximate scores are HALF scores
// (q~.x~ - |x|^2 / 2: see scan_screen_lean3_kernel), the sufficiency test runs in those units: the image errors bound the q.x part as before,
// the norm is the SAME stored number on both sides, and the roundings that see it -- the screening chain starts at 2048 |x|^2 instead of 0
// (<= 408 roundings relative to |x||q| + |x|^2 / 2: 1.22e-5 |x|^2), the exact chain's last step rounds 2 q.x - |x|^2 once -- add 1.5e-5 |x|max^2.
template <bool L2, int NPL = 1>       // NPL: candidates per lane (K' <= 64 NPL); lane l holds candidates l, l + 64
__global__ __launch_bounds__(256) void k_rescore(const u64* __restrict__ cand, int kp, const float* __restrict__ x,
                                                 const float* __restrict__ q, int64_t nq, int k, float xnorm_max, float dx_max,
                                                 int64_t row_base, float* __restrict__ out_s, int64_t* __restrict__ out_r,
                                                 int* __restrict__ flagged /* [0] = number of queries that failed the test */,
                                                 int64_t* __restrict__ flagged_list /* their indices, in arrival order */,
                                                 float* __restrict__ eps_out /* optional [nq]: EPS(q) */, int stride,
                                                 const float* __restrict__ qn2_l2 /* L2: |q|^2 per query */) {
    const int lane = threadIdx.x & 63;
    const int64_t qi = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (qi >= nq) return;
    u64 ck[NPL];
    bool valid[NPL];
    float sa[NPL], acc[NPL];
    u32 row[NPL];
    const float* xv[NPL];
#pragma unroll
    for (int p = 0; p < NPL; ++p) {
        ck[p] = lane + 64 * p < kp ? cand[qi * kp + lane + 64 * p] : 0ull;
        valid[p] = ck[p] != 0ull;
        sa[p] = valid[p] ? rmu_key_score(ck[p]) : -INFINITY;     // approximate score (sorted descending over the candidate index)
        row[p] = valid[p] ? rmu_key_row(ck[p]) : 0u;
        xv[p] = x + (int64_t)row[p] * stride;
        acc[p] = 0.f;
    }
    const float* qv = q + qi * stride;
    float qn2 = 0.f, dq2 = 0.f;
    // (round 6) the row pieces of UN steps are requested TOGETHER, then eaten in order: left as one load pair per step the 48-step chain of a
    // candidate paid a memory round trip per step (24-27 us per launch behind every search; the summation order is untouched)
    constexpr int UN = NPL == 1 ? 24 : 12;      // (second session: 12 / 6 -> 24 / 12 -- two round trips per candidate instead of four; 192 of the 256 registers a wave may take)
    for (int t0 = 0; t0 < SD / 8; t0 += UN) {
        f32x4 xa[NPL][UN], xb[NPL][UN];
#pragma unroll
        for (int u = 0; u < UN; ++u)
#pragma unroll
            for (int p = 0; p < NPL; ++p) {
                xa[p][u] = *(const f32x4*)(xv[p] + 8 * (t0 + u));
                xb[p][u] = *(const f32x4*)(xv[p] + 8 * (t0 + u) + 4);
            }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int t = t0 + u;
            const f32x4 qa = *(const f32x4*)(qv + 8 * t), qb = *(const f32x4*)(qv + 8 * t + 4);
#pragma unroll
            for (int p = 0; p < NPL; ++p) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    acc[p] = fmaf(xa[p][u][c], qa[c], acc[p]);
                    acc[p] = fmaf(xb[p][u][c], qb[c], acc[p]);
                }
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float ua = L2 ? 0.5f * qa[c] : qa[c], ub = L2 ? 0.5f * qb[c] : qb[c];      // the query itself (the L2 index stores 2q)
                qn2 = fmaf(ua, ua, qn2);
                qn2 = fmaf(ub, ub, qn2);
                const float da = ua - (float)(_Float16)(ua * 64.0f) * (1.0f / 64.0f);
                const float db = ub - (float)(_Float16)(ub * 64.0f) * (1.0f / 64.0f);
                dq2 = fmaf(da, da, dq2);
                dq2 = fmaf(db, db, dq2);
            }
        }
    }
    if (L2) {
#pragma unroll
        for (int p = 0; p < NPL; ++p) acc[p] = fmaf(xv[p][SD], qv[SD], acc[p]);
    }
    // sufficiency test on the approximate scores
    int nvalid = 0;
#pragma unroll
    for (int p = 0; p < NPL; ++p) nvalid += __builtin_popcountll(__ballot(valid[p]));
    auto approx_at = [&](int idx) -> float {                    // approximate score of candidate `idx` (uniform)
        float v = __shfl(sa[0], idx & 63);
#pragma unroll
        for (int p = 1; p < NPL; ++p) { const float vp = __shfl(sa[p], idx & 63); v = (idx >> 6) == p ? vp : v; }
        return v;
    };
    const float tau = approx_at(k - 1);                         // k-th best approximate score (or -inf)
    const float smin = approx_at(kp - 1);                       // worst kept candidate
    const float qn = sqrtf(qn2) * 1.0001f, dq = sqrtf(dq2) * 1.0001f;
    const float eps = dx_max * qn + xnorm_max * dq + dx_max * dq + 5.0e-5f * xnorm_max * qn +   // see the header
                      (L2 ? 1.5e-5f * xnorm_max * xnorm_max : 0.f);
    const bool complete = nvalid < kp;                           // every live row was a candidate
    // eps must be finite: a query with |q_i| >= ~1000 overflows fp16(64 q), its approximate scores are inf/NaN and rows
    // scoring NaN are never appended (so even `complete` proves nothing) -- such a query always goes to the exact scan
    const bool ok = __builtin_isfinite(eps) && (complete || (smin < tau - 2.0f * eps));
    if (lane == 0) {
        if (!ok) flagged_list[atomicAdd(flagged, 1)] = qi;
        if (eps_out) eps_out[qi] = eps;
    }
    u64 key[NPL];
    u32 rank[NPL];
#pragma unroll
    for (int p = 0; p < NPL; ++p) key[p] = valid[p] ? rmu_make_key(acc[p] + 0.0f, row[p]) : 0ull;
    rank_keys<NPL>(key, (u32)(kp < 64 * NPL ? kp : 64 * NPL), rank);
    // keys of invalid lanes are 0 and rank below every valid one
#pragma unroll
    for (int p = 0; p < NPL; ++p) {
        if (valid[p] && rank[p] < (u32)k) {
            out_s[qi * k + rank[p]] = L2 ? fmaxf(qn2_l2[qi] - (acc[p] + 0.0f), 0.f) : acc[p] + 0.0f;
            out_r[qi * k + rank[p]] = (int64_t)row[p] + row_base;
        }
        const int le = lane + 64 * p;
        if (le < k && le >= nvalid) {
            out_s[qi * k + le] = L2 ? INFINITY : -INFINITY;
            out_r[qi * k + le] = -1;
        }
    }
}

}  // namespace

int rmu_split_launch(const float* src, void* dst, int64_t n_rows, hipStream_t s, int stride, float scale, u32* zero_a, int n_zero_a, u32* zero_b,
                     int n_zero_b) {
    const int64_t groups = n_rows * (SD / 8);
    if (groups <= 0) return (n_zero_a > 0 || n_zero_b > 0) ? RMU_E_INVALID : RMU_OK;      // (the zeroing rides on a launch that exists)
    if (stride < SD) return RMU_E_INVALID;
    hipLaunchKernelGGL(k_split_rows, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, s, src, (char*)dst, groups, stride, scale, zero_a,
                       zero_a ? n_zero_a : 0, zero_b, zero_b ? n_zero_b : 0);
    return hipGetLastError() == hipSuccess ? RMU_OK : RMU_E_HIP;
}

template <int G, int EXP = 0, int PRE = 4, int NRV = 0, int NT = 0, int NW = 4>
static int screen_launch_cfg(const ScanLaunch* p, hipStream_t s) {
    // function-local static: initialised exactly once, thread-safe (C++11)
    static const hipError_t attr_rc = hipFuncSetAttribute((const void*)scan_screen_kernel<G, EXP, PRE, NRV, NT, NW>,
                                                          hipFuncAttributeMaxDynamicSharedMemorySize, ScreenCfg<G, NRV, NW>::LDS_BYTES);
    if (attr_rc != hipSuccess) return RMU_E_HIP;
    constexpr int lds = ScreenCfg<G, NRV, NW>::LDS_BYTES;
    hipLaunchKernelGGL((scan_screen_kernel<G, EXP, PRE, NRV, NT, NW>), dim3(p->grid), dim3(64 * NW), lds, s, *p);
    return hipGetLastError() == hipSuccess ? RMU_OK : RMU_E_HIP;
}

int rmu_screen_lds_bytes(int qg) { return qg == 2 ? ScreenCfg<2>::LDS_BYTES : ScreenCfg<1>::LDS_BYTES; }

// geometry of one screening launch: p->qg 32-query groups per wave (2 when the batch fills 256-query workgroups),
// S row chunks (a multiple of 8 for the XCD-aware block map) so that grid = S * nqt fills the 256 CUs evenly
int rmu_screen_plan(ScanLaunch* p) {
    if (p->k < 1 || p->k > RMU_KS_CAP_DEEP - 8 || p->nq < 1 || p->n_rows < 0 || p->dpad != SD) return RMU_E_INVALID;   // (p->k is K': 32, 40, or up to 120 for 32 < k <= 104)
    // (round 5) The PRODUCT library carries exactly the kernels it takes: scan_screen_lean3_kernel<NW = 8> for full query tiles and
    // <NW = 4, nt> for one query tile.  The earlier forms of the same kernel (scan_screen_kernel in its 4- and 8-wave instantiations,
    // scan_screen_lean_kernel, scan_screen_lean2_kernel) and the switches that select them (RMU_SCREEN_G / _W8 / _LEAN / _LEAN4) exist in
    // debug builds only (python -m ragmeup_amd.build --debug-kernels), where tools/pace_probe.py reproduces DESIGN.md 4.5's A/B table.
#ifdef RMU_DEBUG_KERNELS
    static const int force_g_env = rmu_env("RMU_SCREEN_G") ? atoi(rmu_env("RMU_SCREEN_G")) : 0;
    const int force_g = p->k > 32 ? 0 : force_g_env;      // K' = 40 (24 < k <= 32) exists for the lean3 kernel only: slots of RMU_KS_CAP keys
#else
    constexpr int force_g = 0;
#endif
    p->qg = force_g == 1 || force_g == 2 ? force_g : (p->nq > 128 ? 2 : 1);
    p->wq = 4; p->kv = 0;
    // Round-4 experiments (debug builds only; all bit-identical to the product kernel, none faster -- DESIGN.md 4.5): RMU_SCREEN_G4=1 = one wave per
    // SIMD, 128 queries per wave, four MFMAs per LDS fragment (scan_screen_g4_kernel, batches >= 512); RMU_SCREEN_KS=1 = K-split pairs
    // (scan_screen_ks_kernel; RMU_SCREEN_KPP=0 for its interleaved form)
#ifdef RMU_DEBUG_KERNELS
    static const int g4_env = rmu_env("RMU_SCREEN_G4") ? atoi(rmu_env("RMU_SCREEN_G4")) : 0;
    static const int ks_env = rmu_env("RMU_SCREEN_KS") ? atoi(rmu_env("RMU_SCREEN_KS")) : 0;
    const int g4 = p->k > 32 ? 0 : g4_env, ks = p->k > 32 ? 0 : ks_env;
#else
    constexpr int g4 = 0, ks = 0;
#endif
    const bool use_g4 = g4 && !force_g && p->nq >= 512;
    // full query tiles (> 128 queries): 8 waves x 32 queries (two waves per SIMD) instead of 4 x 64 -- RMU_SCREEN_W8=0 keeps the 4-wave form
#ifdef RMU_DEBUG_KERNELS
    static const int w8_env = rmu_env("RMU_SCREEN_W8") ? atoi(rmu_env("RMU_SCREEN_W8")) : 1;
    const int w8 = p->k > 32 ? 1 : w8_env;
#else
    constexpr int w8 = 1;
#endif
    if (w8 && p->qg == 2 && !force_g) { p->qg = 1; p->wq = 8; }
    if (use_g4) { p->qg = 4; p->wq = 4; }
    const int qwg = use_g4 ? 512 : p->wq == 8 ? 256 : 128 * p->qg;
    p->nqt = (p->nq + qwg - 1) / qwg;
    const int64_t tiles_total = (p->n_rows + S_RT - 1) / S_RT;
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
    const int64_t used = (tiles_total + p->tiles_per_chunk - 1) / p->tiles_per_chunk;
    if (used > 0 && used < s) s = (int)used;
    p->s_chunks = s;
    p->grid = s * p->nqt;
    p->parts = s;
    static const int nt_env = rmu_env("RMU_NT") ? atoi(rmu_env("RMU_NT")) : 1;
    p->nt = (nt_env && p->nqt == 1 && p->qg == 1) ? 1 : 0;     // one query tile: each image byte is read by one workgroup
    // lean form with one barrier per tile and candidates in global memory (scan_screen_lean2_kernel): RMU_SCREEN_LEAN=2
#ifdef RMU_DEBUG_KERNELS
    static const int lean_env0 = rmu_env("RMU_SCREEN_LEAN") ? atoi(rmu_env("RMU_SCREEN_LEAN")) : 3;
    static const int lean4_env = rmu_env("RMU_SCREEN_LEAN4") ? atoi(rmu_env("RMU_SCREEN_LEAN4")) : 1;     // 0: round 3's kernel for one query tile
    const int lean_env = p->k > 32 ? 3 : lean_env0, lean4 = p->k > 32 ? 1 : lean4_env;
#else
    constexpr int lean_env = 3, lean4 = 1;
#endif
    const bool one_tile = p->wq == 4 && p->qg == 1 && p->nqt == 1;
    p->kv = use_g4 ? 2 : (ks && p->wq == 8) ? 1 : (lean_env == 3 && (p->wq == 8 || (lean4 && one_tile))) ? 4 : (lean_env == 2 && p->wq == 8) ? 3 : 0;
#ifndef RMU_DEBUG_KERNELS
    if (p->kv != 4) return RMU_E_INVALID;      // every product geometry is one of the two lean3 instantiations
#endif
    p->lds_bytes = p->kv == 4 ? (p->wq == 8 ? Lean3Cfg<8>::LDS_BYTES : Lean3Cfg<4>::LDS_BYTES)   /* (the DEEP forms take the same LDS) */ : p->kv == 3 ? Lean2Cfg::LDS_BYTES : p->kv == 2 ? G4Cfg::LDS_BYTES : p->kv ? KsCfg::LDS_BYTES : p->wq == 8 ? ScreenCfg<1, 0, 8>::LDS_BYTES : rmu_screen_lds_bytes(p->qg);
    // sibling pacing (see the kernel): query tiles of a chunk on one XCD, 2..4 of them, the whole grid resident at once (these
    // kernels take > 80 KiB of LDS: one workgroup per CU), and enough tiles per workgroup for drift to matter
    // window in tiles (0 = off).  Measured (tools/pace_probe.py, 10M x 1024): 0 / 4 / 8 / 16 / 32 all 7.82-7.86 ms of scan kernels -- the pacing
    // costs nothing -- and on the round-4 boxes the siblings did not drift without it either (FETCH_SIZE 7.75-7.80 GB per batch = 1.01x
    // the image, L2 hit 0.758 with pacing off AND on; round 3's boxes: 15.1 GB, 0.53): kept on as the bound on that drift.
    static const int pace_env = rmu_env("RMU_SCREEN_PACE") ? atoi(rmu_env("RMU_SCREEN_PACE")) : 8;
    static const int n_cu = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 0;
        return n;
    }();
    p->pace = (pace_env > 0 && p->qg == 1 && p->nqt >= 2 && p->nqt <= 4 && (p->s_chunks & 7) == 0 && p->grid <= n_cu &&
               p->tiles_per_chunk >= 4 * pace_env) ? pace_env : 0;
    return RMU_OK;
}

#ifdef RMU_DEBUG_KERNELS
template <int EXP>
static int screen_launch_ks(const ScanLaunch* p, hipStream_t s) {
    static const hipError_t attr_rc = hipFuncSetAttribute((const void*)scan_screen_ks_kernel<EXP>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                          KsCfg::LDS_BYTES);
    if (attr_rc != hipSuccess) return RMU_E_HIP;
    if (!p->gcand) return RMU_E_INVALID;
    hipLaunchKernelGGL((scan_screen_ks_kernel<EXP>), dim3(p->grid), dim3(512), KsCfg::LDS_BYTES, s, *p);
    return hipGetLastError() == hipSuccess ? RMU_OK : RMU_E_HIP;
}

template <int EXP>
static int screen_launch_g4(const ScanLaunch* p, hipStream_t s) {
    static const hipError_t attr_rc = hipFuncSetAttribute((const void*)scan_screen_g4_kernel<EXP>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                          G4Cfg::LDS_BYTES);
    if (attr_rc != hipSuccess) return RMU_E_HIP;
    if (!p->gcand) return RMU_E_INVALID;
    hipLaunchKernelGGL((scan_screen_g4_kernel<EXP>), dim3(p->grid), dim3(256), G4Cfg::LDS_BYTES, s, *p);
    return hipGetLastError() == hipSuccess ? RMU_OK : RMU_E_HIP;
}
#endif

template <int EXP, int NWV, int NT, int L2N, bool DEEP = false>
static int screen_launch_lean3(const ScanLaunch* p, hipStream_t s) {
    static const hipError_t attr_rc = hipFuncSetAttribute((const void*)scan_screen_lean3_kernel<EXP, NWV, NT, L2N, DEEP>,
                                                          hipFuncAttributeMaxDynamicSharedMemorySize, (Lean3Cfg<NWV, DEEP>::LDS_BYTES));
    if (attr_rc != hipSuccess) return RMU_E_HIP;
    constexpr int lds = Lean3Cfg<NWV, DEEP>::LDS_BYTES;
    hipLaunchKernelGGL((scan_screen_lean3_kernel<EXP, NWV, NT, L2N, DEEP>), dim3(p->grid), dim3(64 * NWV), lds, s, *p);
    return hipGetLastError() == hipSuccess ? RMU_OK : RMU_E_HIP;
}

int rmu_screen_launch(const ScanLaunch* p, hipStream_t s) {
    if (p->kv == 4) {
        if (!p->gcand) return RMU_E_INVALID;
        const bool l2 = p->nrm != nullptr;                // RMU_METRIC_L2SQ: the row norms enter the chain as its C operand
        if (p->k > RMU_KS_CAP - 8) {                      // deep K' (32 < k <= 104): slots of RMU_KS_CAP_DEEP keys
            if (p->wq == 4) {
                if (p->nt) return l2 ? screen_launch_lean3<0, 4, 1, 1, true>(p, s) : screen_launch_lean3<0, 4, 1, 0, true>(p, s);
                return l2 ? screen_launch_lean3<0, 4, 0, 1, true>(p, s) : screen_launch_lean3<0, 4, 0, 0, true>(p, s);
            }
            return l2 ? screen_launch_lean3<0, 8, 0, 1, true>(p, s) : screen_launch_lean3<0, 8, 0, 0, true>(p, s);
        }
        if (p->wq == 4) {                                 // one query tile
            if (p->nt) return l2 ? screen_launch_lean3<0, 4, 1, 1>(p, s) : screen_launch_lean3<0, 4, 1, 0>(p, s);
            return l2 ? screen_launch_lean3<0, 4, 0, 1>(p, s) : screen_launch_lean3<0, 4, 0, 0>(p, s);
        }
#ifdef RMU_DEBUG_KERNELS
        if (p->dbg && !l2) return screen_launch_lean3<4, 8, 0, 0>(p, s);
#endif
        return l2 ? screen_launch_lean3<0, 8, 0, 1>(p, s) : screen_launch_lean3<0, 8, 0, 0>(p, s);
    }
#ifndef RMU_DEBUG_KERNELS
    return RMU_E_INVALID;                      // rmu_screen_plan hands the product library kv == 4 only
#else
    if (p->kv == 3) {
        static const hipError_t attr_rc = hipFuncSetAttribute((const void*)scan_screen_lean2_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                              Lean2Cfg::LDS_BYTES);
        if (attr_rc != hipSuccess) return RMU_E_HIP;
        if (!p->gcand) return RMU_E_INVALID;
#ifdef RMU_DEBUG_KERNELS
        if (p->dbg) {
            static const hipError_t attr_d = hipFuncSetAttribute((const void*)scan_screen_lean2_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                                 Lean2Cfg::LDS_BYTES);
            if (attr_d != hipSuccess) return RMU_E_HIP;
            hipLaunchKernelGGL((scan_screen_lean2_kernel<4>), dim3(p->grid), dim3(512), Lean2Cfg::LDS_BYTES, s, *p);
            return hipGetLastError() == hipSuccess ? RMU_OK : RMU_E_HIP;
        }
#endif
        hipLaunchKernelGGL((scan_screen_lean2_kernel<0>), dim3(p->grid), dim3(512), Lean2Cfg::LDS_BYTES, s, *p);
        return hipGetLastError() == hipSuccess ? RMU_OK : RMU_E_HIP;
    }
#ifdef RMU_DEBUG_KERNELS
    if (p->kv == 2) {
        if (p->dbg) return screen_launch_g4<4>(p, s);
        static const int ex4 = rmu_env("RMU_SCREEN_EXP") ? atoi(rmu_env("RMU_SCREEN_EXP")) : 0;   // timing ablations (wrong results)
        if (ex4 == 1) return screen_launch_g4<1>(p, s);
        if (ex4 == 2) return screen_launch_g4<2>(p, s);
        if (ex4 == 3) return screen_launch_g4<3>(p, s);
        if (ex4 == 8) return screen_launch_g4<8>(p, s);
        if (ex4 == 11) return screen_launch_g4<11>(p, s);
        return screen_launch_g4<0>(p, s);
    }
    if (p->kv == 1) {
        if (p->dbg) return (rmu_env("RMU_SCREEN_KPP") && atoi(rmu_env("RMU_SCREEN_KPP")) == 0) ? screen_launch_ks<4>(p, s) : screen_launch_ks<20>(p, s);
        static const int kpp = rmu_env("RMU_SCREEN_KPP") ? atoi(rmu_env("RMU_SCREEN_KPP")) : 1;
        if (kpp) return screen_launch_ks<16>(p, s);
        return screen_launch_ks<0>(p, s);
    }
#else
    if (p->kv != 0) return RMU_E_INVALID;     // (3 and 4 were handled above)
#endif
#ifdef RMU_DEBUG_KERNELS      // timing ablations, ring / prefetch depth experiments, cycle counters (wrong results by design for EXP != 0):
                              // python -m ragmeup_amd.build --debug-kernels; tools/ablate_screen.sh
    static const int ex = rmu_env("RMU_SCREEN_EXP") ? atoi(rmu_env("RMU_SCREEN_EXP")) : 0;
    static const int pre = rmu_env("RMU_SCREEN_SPRE") ? atoi(rmu_env("RMU_SCREEN_SPRE")) : 4;
    if (p->wq == 8 && p->dbg) {
        static const int ppd = rmu_env("RMU_SCREEN_PP") ? atoi(rmu_env("RMU_SCREEN_PP")) : 0;
        return ppd ? screen_launch_cfg<1, 20, 4, 0, 0, 8>(p, s) : screen_launch_cfg<1, 4, 4, 0, 0, 8>(p, s);
    }
    if (p->dbg) {
        if (p->qg == 2 && ex == 8) return screen_launch_cfg<2, 12>(p, s);
        if (p->qg == 2 && ex == 9) return screen_launch_cfg<2, 13>(p, s);
        if (p->qg == 2 && ex == 10) return screen_launch_cfg<2, 14>(p, s);
        if (p->qg == 2 && ex == 11) return screen_launch_cfg<2, 15>(p, s);
        return p->qg == 2 ? screen_launch_cfg<2, 4>(p, s) : screen_launch_cfg<1, 4>(p, s);
    }
    if (p->qg == 2) {
        if (ex == 1) return screen_launch_cfg<2, 1>(p, s);
        if (ex == 2) return screen_launch_cfg<2, 2>(p, s);
        if (ex == 3) return screen_launch_cfg<2, 3>(p, s);
        if (ex == 8) return screen_launch_cfg<2, 8>(p, s);
        if (ex == 9) return screen_launch_cfg<2, 9>(p, s);
        if (ex == 10) return screen_launch_cfg<2, 10>(p, s);
        if (ex == 11) return screen_launch_cfg<2, 11>(p, s);
        static const int nrv = rmu_env("RMU_SCREEN_NR") ? atoi(rmu_env("RMU_SCREEN_NR")) : 0;
        if (nrv == 4) return screen_launch_cfg<2, 0, 4, 4>(p, s);
        if (nrv == 5) return screen_launch_cfg<2, 0, 4, 5>(p, s);
        if (pre == 6) return screen_launch_cfg<2, 0, 6>(p, s);
        if (pre == 3) return screen_launch_cfg<2, 0, 3>(p, s);
    }
#endif
    if (p->wq == 8) {                                                                 // full query tiles: 8 waves x 32 queries
        static const int lean = rmu_env("RMU_SCREEN_LEAN") ? atoi(rmu_env("RMU_SCREEN_LEAN")) : 1;   // 0: round 3's form of the same kernel
        if (lean) {
            static const hipError_t attr_rc = hipFuncSetAttribute((const void*)scan_screen_lean_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                                  LeanCfg::LDS_BYTES);
            if (attr_rc != hipSuccess) return RMU_E_HIP;
#ifdef RMU_DEBUG_KERNELS
            if (p->dbg) {
                static const hipError_t attr_d = hipFuncSetAttribute((const void*)scan_screen_lean_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                                     LeanCfg::LDS_BYTES);
                if (attr_d != hipSuccess) return RMU_E_HIP;
                hipLaunchKernelGGL((scan_screen_lean_kernel<4>), dim3(p->grid), dim3(512), LeanCfg::LDS_BYTES, s, *p);
                return hipGetLastError() == hipSuccess ? RMU_OK : RMU_E_HIP;
            }
#endif
            hipLaunchKernelGGL((scan_screen_lean_kernel<0>), dim3(p->grid), dim3(512), LeanCfg::LDS_BYTES, s, *p);
            return hipGetLastError() == hipSuccess ? RMU_OK : RMU_E_HIP;
        }
#ifdef RMU_DEBUG_KERNELS
        // the PING-PONG form (EXP bit 4; see the kernel): measured 8.27-8.32 ms of scan kernels per 10M x 1024 batch against 7.69-7.81 for the
        // interleaved form (compares in the compute segment; 9.3 with them in the load segment, 9.0-9.1 with the DMA there, 11.0 in the first cut)
        static const int pp = rmu_env("RMU_SCREEN_PP") ? atoi(rmu_env("RMU_SCREEN_PP")) : 0;
        if (pp) return screen_launch_cfg<1, 16, 4, 0, 0, 8>(p, s);
#endif
        return screen_launch_cfg<1, 0, 4, 0, 0, 8>(p, s);
    }
    if (p->qg == 2) return screen_launch_cfg<2>(p, s);                                // RMU_SCREEN_W8=0 / RMU_SCREEN_G=2: 4 waves x 64 queries
    return p->nt ? screen_launch_cfg<1, 0, 4, 0, 1>(p, s) : screen_launch_cfg<1>(p, s);
#endif
}

int rmu_img_err_launch(const float* x, int64_t n_rows, float* err2, hipStream_t s, int stride) {
    if (n_rows <= 0) return RMU_OK;
    if (stride < SD) return RMU_E_INVALID;
    hipLaunchKernelGGL(k_img_err, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), 0, s, x, n_rows, err2, stride);
    return hipGetLastError() == hipSuccess ? RMU_OK : RMU_E_HIP;
}

int rmu_rescore_launch(const u64* cand, int kp, const float* x, const float* q, int64_t nq, int k, float xnorm_max, float dx_max,
                       int64_t row_base, float* out_s, int64_t* out_r, int* flagged, int64_t* flagged_list, float* eps_out, hipStream_t s,
                       int stride, const float* qn2_l2) {
    if (kp < k || kp > 128 || stride < SD || (qn2_l2 && stride < SD + 1)) return RMU_E_INVALID;
    const dim3 grid((unsigned)((nq + 3) / 4));
    if (kp > 64) {       // deep K' (32 < k <= 104): two candidates per lane
        if (qn2_l2)
            hipLaunchKernelGGL((k_rescore<true, 2>), grid, dim3(256), 0, s, cand, kp, x, q, nq, k, xnorm_max, dx_max, row_base, out_s, out_r, flagged,
                               flagged_list, eps_out, stride, qn2_l2);
        else
            hipLaunchKernelGGL((k_rescore<false, 2>), grid, dim3(256), 0, s, cand, kp, x, q, nq, k, xnorm_max, dx_max, row_base, out_s, out_r, flagged,
                               flagged_list, eps_out, stride, qn2_l2);
    } else if (qn2_l2)
        hipLaunchKernelGGL((k_rescore<true, 1>), grid, dim3(256), 0, s, cand, kp, x, q, nq, k, xnorm_max, dx_max, row_base,
                           out_s, out_r, flagged, flagged_list, eps_out, stride, qn2_l2);
    else
        hipLaunchKernelGGL((k_rescore<false, 1>), grid, dim3(256), 0, s, cand, kp, x, q, nq, k, xnorm_max, dx_max, row_base,
                           out_s, out_r, flagged, flagged_list, eps_out, stride, qn2_l2);
    return hipGetLastError() == hipSuccess ? RMU_OK : RMU_E_HIP;
}

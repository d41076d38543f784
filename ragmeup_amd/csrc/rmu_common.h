// rmu_common.h -- shared device/host helpers for librmu.so (gfx950 only).
#pragma once
#include <mutex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;
typedef unsigned int u32;

// ---- candidate key: one u64 compares as (score desc, row asc) ---------------------------------
// hi 32 bits: order-preserving image of the fp32 score; lo 32 bits: ~row (smaller row = larger key).
// key 0 is the "no candidate" sentinel (smaller than every real key).
__host__ __device__ inline u32 rmu_f2ord(float f) {
    union { float f; u32 u; } c; c.f = f;
    return c.u ^ ((c.u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__host__ __device__ inline float rmu_ord2f(u32 o) {
    union { float f; u32 u; } c;
    c.u = o ^ ((o >> 31) ? 0x80000000u : 0xFFFFFFFFu);
    return c.f;
}
__host__ __device__ inline u64 rmu_make_key(float score, u32 row) {
    return ((u64)rmu_f2ord(score) << 32) | (u64)(u32)(~row);
}
__host__ __device__ inline float rmu_key_score(u64 k) { return rmu_ord2f((u32)(k >> 32)); }
__host__ __device__ inline u32 rmu_key_row(u64 k) { return ~(u32)k; }

// ---- wave-wide bitonic sort, descending, 64*NPL keys (lane holds elements lane + 64*p) ---------
template <int NPL>
__device__ __forceinline__ void rmu_bitonic_sort_desc(u64 (&key)[NPL], int lane) {
#pragma unroll
    for (int size = 2; size <= 64 * NPL; size <<= 1) {
#pragma unroll
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            if (stride >= 64) {
                const int ps = stride >> 6;
#pragma unroll
                for (int p = 0; p < NPL; ++p) {
                    if ((p & ps) == 0) {
                        const int p2 = p | ps;
                        const bool desc = (((lane + 64 * p) & size) == 0);
                        const u64 a = key[p], b = key[p2];
                        const u64 mx = a > b ? a : b, mn = a > b ? b : a;
                        key[p] = desc ? mx : mn;
                        key[p2] = desc ? mn : mx;
                    }
                }
            } else {
#pragma unroll
                for (int p = 0; p < NPL; ++p) {
                    const u64 other = __shfl_xor(key[p], stride);
                    const bool desc = (((lane + 64 * p) & size) == 0);
                    const bool lower = ((lane & stride) == 0);
                    const u64 mx = key[p] > other ? key[p] : other;
                    const u64 mn = key[p] > other ? other : key[p];
                    key[p] = (desc == lower) ? mx : mn;
                }
            }
        }
    }
}

// bitonic MERGE (input is a bitonic sequence of 64*NPL keys) -> sorted descending
template <int NPL>
__device__ __forceinline__ void rmu_bitonic_merge_desc(u64 (&key)[NPL], int lane) {
#pragma unroll
    for (int stride = 32 * NPL; stride > 0; stride >>= 1) {
        if (stride >= 64) {
            const int ps = stride >> 6;
#pragma unroll
            for (int p = 0; p < NPL; ++p) {
                if ((p & ps) == 0) {
                    const int p2 = p | ps;
                    const u64 a = key[p], b = key[p2];
                    key[p] = a > b ? a : b;
                    key[p2] = a > b ? b : a;
                }
            }
        } else {
#pragma unroll
            for (int p = 0; p < NPL; ++p) {
                const u64 other = __shfl_xor(key[p], stride);
                const bool lower = ((lane & stride) == 0);
                const u64 mx = key[p] > other ? key[p] : other;
                const u64 mn = key[p] > other ? other : key[p];
                key[p] = lower ? mx : mn;
            }
        }
    }
}

// Tuning / experiment switches (RMU_GEMM3, RMU_SCREEN_PACE, ...) are honoured ONLY when RMU_TUNING=1 is set as well: a stray
// RMU_* variable in a server's environment cannot change which kernels run.  Every switch is read once per process.  A process that
// sets RMU_* variables WITHOUT the master switch is told so once on stderr (they used to be honoured: silence would hide the change).
extern "C" char** environ;
inline const char* rmu_env(const char* name) {
    static const bool on = [] {
        const char* t = getenv("RMU_TUNING");
        const bool o = t != nullptr && atoi(t) == 1;
        if (!o && environ) {
            for (char** e = environ; *e; ++e)
                // (RMU_BENCH_* / RMU_REFERENCE_DIR belong to bench.py and the tests, not to the library)
                if (!strncmp(*e, "RMU_", 4) && strncmp(*e, "RMU_TUNING=", 11) && strncmp(*e, "RMU_SCREEN=", 11) && strncmp(*e, "RMU_GRAPH=", 10) &&
                    strncmp(*e, "RMU_BENCH_", 10) && strncmp(*e, "RMU_REFERENCE_DIR=", 18)) {
                    fprintf(stderr, "librmu: %.*s is set but ignored: tuning switches are honoured only with RMU_TUNING=1 (DESIGN.md 6.1)\n",
                            (int)(strchr(*e, '=') ? strchr(*e, '=') - *e : (long)strlen(*e)), *e);
                    break;
                }
        }
        return o;
    }();
    return on ? getenv(name) : nullptr;
}
// The two SAFETY kill switches deployments may rely on -- RMU_SCREEN=0 (no fp16 screening image: every search is the exact fp32 scan) and
// RMU_GRAPH=0 (rmu_bert_encode_host never captures a hipGraph) -- are honoured with or without the master switch: they only ever select
// the more conservative path.
inline const char* rmu_env_kill(const char* name) { return getenv(name); }

// ---- hipGraph captures in the same process (round 6; measured: tools/ubench/capture_probe.hip -> profiles/r06_capture_probe.txt) ----------
// The reference runs its LLM (PyTorch) on the same GPU and in the same process as this library (server/RAGHelper_local.py:42-105); a
// torch.cuda.graph capture there uses the GLOBAL capture mode.  On ROCm 7 a second thread whose capture-interaction mode is the default
// invalidates such a capture with nearly any synchronous call (hipMalloc, hipFree, hipStreamSynchronize, hipEventSynchronize,
// hipEventQuery, hipHostMalloc ...); with the thread's mode exchanged to RELAXED -- what PyTorch's own allocator does around
// cudaMalloc -- every one of them is harmless, EXCEPT hipDeviceSynchronize and the synchronous (pageable) hipMemcpy / hipMemset, which
// break a capture on another thread in every mode (and did break this library's own thread-local captures in round 5).  Hence:
//   1. every entry point runs under RMU_ENTRY(): the calling thread is in relaxed mode for the duration of the call;
//   2. the library never calls hipDeviceSynchronize, hipMemcpy or hipMemset: it waits for exactly the streams / events that used a
//      buffer (the index's reader events, a context's tail event) and copies with hipMemcpyAsync + hipStreamSynchronize.
struct RmuRelaxedCapture {
    hipStreamCaptureMode m = hipStreamCaptureModeRelaxed;
    RmuRelaxedCapture() { (void)hipThreadExchangeStreamCaptureMode(&m); }
    ~RmuRelaxedCapture() { (void)hipThreadExchangeStreamCaptureMode(&m); }
    RmuRelaxedCapture(const RmuRelaxedCapture&) = delete;
    RmuRelaxedCapture& operator=(const RmuRelaxedCapture&) = delete;
};
#define RMU_ENTRY() RmuRelaxedCapture rmu_relaxed_capture_
// This library's own captures (the host-path forwards of bert.hip, thread-local mode) are taken one at a time (round 5: two threads
// capturing at once failed 1 run in 4 on this runtime).
inline std::mutex& rmu_capture_mutex() {
    static std::mutex* mu = new std::mutex;      // (never destroyed: thread-local workspaces are released while the process winds down)
    return *mu;
}
// hipFree waits for the device by itself; in relaxed mode it leaves captures of other threads alone (probe above)
inline hipError_t rmu_free(void* p) {
    RMU_ENTRY();
    return hipFree(p);
}

// ---- launch descriptors shared between rmu_api.hip and the kernel translation units -------------
// Device-side launch predicate.  The screening path decides per query ON THE DEVICE whether the exact scan has to re-run
// it; the re-run launches are enqueued unconditionally and every workgroup of a launch whose predicate is false returns
// at once, so a search never needs a host round trip (and honours a caller-supplied stream).
//   p == null: always run.  Otherwise c = *p (number of flagged queries): run iff lo <= c <= hi; clamp != 0: only the first
//   min(nq, c) queries of the launch exist.
struct RmuCond {
    const int* p;
    int lo, hi, clamp;
};

struct ScanLaunch {
    const float* x;        // [n_rows, dpad] fp32 row-major, HBM resident
    int64_t n_rows;
    int64_t row0;          // first row of the scanned range: screening scan rows [row0, row0 + n_rows) of the image `x`; exact scan: `x`
                           // already points at row row0 and row0 is only added to the row ids of the keys
    int dpad;              // row stride in floats (multiple of 96)
    const float* q;        // [nq, dpad] fp32 device (padded like the rows)
    int nq;
    int k;
    u64* partial;          // [parts, nq, k] keys
    int share_thr;         // bit 0 = read the shared thresholds (0: publish only; debugging aid); bit 1 = no filter (timing ablation); bit 2 (screening scan) =
                           // emit slots of <= k entries unsorted (the launch's merge must be told: rmu_merge_to_keys_launch(..., unsorted = 1))
    u32* gthr;             // [nq] shared per-query threshold (order-preserving u32 image, 0 = none), zeroed per launch
    int parts;             // filled by the planner
    u64* dbg;              // optional debug counters (nullptr in production): [0] slow tiles, [1] compactions, [2] appends, [3] tiles
    // filled by rmu_scan_plan
    int wq, kv, s_chunks, nqt, tiles_per_chunk, grid, lds_bytes;
    int qg;                // screening scan only: 32-query groups per wave (rmu_screen_plan)
    int nt;                // 1 = stream the corpus with non-temporal loads (every byte is read by exactly one workgroup)
    RmuCond cond;          // exact scan only: device-side launch predicate (zero-initialised = always)
    // screening scan only: sibling pacing (scan_screen.hip).  prog = [s_chunks][4] progress words of the query-tile workgroups
    // of each row chunk (zeroed per launch; nullptr = off), pace = how many tiles a workgroup may run ahead of its slowest sibling
    u32* prog;
    int pace;
    // screening scan, K-split form only (kv == 1; scan_screen.hip): candidate slots in global memory, [parts][nq][RMU_KS_CAP] keys.  Needs no
    // initialisation (the slot counts live in registers); contents are dead once the launch has written its partials
    u64* gcand;
    // screening scan of an RMU_METRIC_L2SQ index: -2048 |x|^2 per image row (indexed like the image: row0 is added); nullptr = inner product
    const float* nrm;
};
#define RMU_KS_CAP 48     /* K' <= 40 kept candidates + 8 free slots between compactions (one key per lane in the rank: <= 64) */
#define RMU_KS_CAP_DEEP 128   /* (round 6) 32 < k <= 104: K' <= 120 kept candidates + 8 free slots, two keys per lane in the rank */

int rmu_scan_plan(ScanLaunch* p);                        // chooses geometry; returns 0 or RMU_E_INVALID
int rmu_scan_launch(const ScanLaunch* p, hipStream_t s); // launches the fused scan
int rmu_merge_final_launch(const u64* partial, int parts, int64_t nq, int k, int64_t row_base, int l2_out, const float* qnorm2,
                           float* out_scores, int64_t* out_rows, const int64_t* scatter /* or null */, const RmuCond* cond /* or null */,
                           hipStream_t s);
int rmu_merge_to_keys_launch(const u64* partial, int parts, int64_t nq, int k, u64* out_keys, u32* seed_thr /* or null */,
                             hipStream_t s, int unsorted = 0 /* the lists are compact but not sorted (ScanLaunch::share_thr bit 2) */);
// fp16 screening path (scan_screen.hip)
#define RMU_IMG_ROW_BYTES 768                          /* fp16(64 x) image of a 384-d row */
int rmu_split_launch(const float* src, void* dst, int64_t n_rows, hipStream_t s, int stride = 384,
                     float scale = 64.0f, u32* zero_a = nullptr, int n_zero_a = 0, u32* zero_b = nullptr, int n_zero_b = 0);                                             // fp32 [n, 384 of stride] -> fp16(scale x) image
int rmu_screen_launch(const ScanLaunch* p, hipStream_t s);                             // x/q = split images, k = K'
int rmu_screen_lds_bytes(int qg);
int rmu_screen_plan(ScanLaunch* p);                      // geometry of one screening launch (k = K' <= 32)
int rmu_img_err_launch(const float* x, int64_t n_rows, float* err2, hipStream_t s, int stride = 384);   // |x - image|^2 per row
// stride = floats between rows of x AND of q; qn2_l2 (RMU_METRIC_L2SQ: |q|^2 per query; x and q in the L2 index's augmented form) or null
int rmu_rescore_launch(const u64* cand, int kp, const float* x, const float* q, int64_t nq, int k, float xnorm_max, float dx_max,
                       int64_t row_base, float* out_s, int64_t* out_r, int* flagged /* [0] = count */, int64_t* flagged_list,
                       float* eps_out /* or null */, hipStream_t s, int stride = 384, const float* qn2_l2 = nullptr);
// shard lists [parts][nq, k]: part p's scores start at scores + p * stride_s (floats), rows at rows + p * stride_r (int64);
// smaller_better: distances (RMU_METRIC_L2SQ) instead of similarities
int rmu_merge_lists_launch(const float* scores, const int64_t* rows, int parts, int64_t stride_s, int64_t stride_r, int64_t nq, int k,
                           int smaller_better, float* out_scores, int64_t* out_rows, hipStream_t s);

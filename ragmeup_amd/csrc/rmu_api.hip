// rmu_api.hip -- C-ABI of librmu.so (see include/rmu.h): HBM-resident flat index + search orchestration.
//
// Reference call sites served (server/ = /root/reference/server):
//   rmu_index_create      RAGHelper.py:385-404   (Milvus.from_documents([],...) / PGVector(...))
//   rmu_index_add         RAGHelper.py:431, 525  (db.add_documents(documents, ids=ids))
//   rmu_index_search      RAGHelper.py:497-499   (dense retriever -> FLAT similarity search)
//   rmu_index_get_rows    RAGHelper.py:497-499   (search_type="mmr": re-fetch the fetch_k vectors)
//   rmu_index_remove_rows server.py:373-377      (collection.delete('source == ...'))
//   rmu_topk_merge        no counterpart (8-GPU shard merge)
//
// Data layout in HBM: one row-major [capacity, dpad] fp32 matrix, dpad = dim rounded up to 192/384/768
// (zero padded), base 256-B aligned so every 1536-B row of the 384-d flagship is 12 full 128-B lines.
// Tombstoned rows are NaN-poisoned in place: every score against them is NaN and fails the scan's
// `score > threshold` compare, so deletion costs nothing in the hot loop.
#include <mutex>
#include <chrono>
#include <shared_mutex>
#include <string>
#include <vector>
#include <cmath>
#include <cstdio>
#include <algorithm>
#include <cstring>
#include <cstdlib>

#include "rmu_common.h"
#include "../../include/rmu.h"

// ------------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------------
static thread_local std::string g_err;
static int fail(int code, const std::string& msg) { g_err = msg; return code; }
#define HIP_TRY(expr)                                                                            \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return fail(e_ == hipErrorOutOfMemory ? RMU_E_OOM : RMU_E_HIP,                       \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                      \
    } while (0)

extern "C" const char* rmu_last_error(void) { return g_err.c_str(); }
extern "C" void rmu_set_error_(const char* msg) { g_err = msg ? msg : ""; }   // used by bert.hip
extern "C" const char* rmu_version(void) { return "librmu 0.1 gfx950"; }

static int g_device = -1;
extern "C" int rmu_init(int device_ordinal) {
    RMU_ENTRY();
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    if (device_ordinal < 0 || device_ordinal >= n) return fail(RMU_E_INVALID, "rmu_init: no such device");
    HIP_TRY(hipSetDevice(device_ordinal));
    g_device = device_ordinal;
    return RMU_OK;
}

// ------------------------------------------------------------------------------------------------
// per-thread context: stream, events, grow-only workspace
// ------------------------------------------------------------------------------------------------
// Set once the process starts tearing the HIP runtime down (static destructors / atexit): thread-local destructors
// that run after that point must not call into HIP any more.
static bool g_runtime_down = false;
namespace { struct RuntimeGuard { ~RuntimeGuard() { g_runtime_down = true; } } g_runtime_guard; }

struct Buf {
    void* p = nullptr;
    size_t cap = 0;
    void release() {
        if (p && !g_runtime_down) (void)rmu_free(p);
        p = nullptr; cap = 0;
    }
    int ensure(size_t bytes) {
        if (bytes <= cap) return RMU_OK;
        if (p) (void)rmu_free(p);
        p = nullptr; cap = 0;
        size_t want = bytes + bytes / 4 + 256;
        if (hipMalloc(&p, want) != hipSuccess) { p = nullptr; return RMU_E_OOM; }
        cap = want;
        return RMU_OK;
    }
};
struct Tls {
    hipStream_t stream = nullptr;
    hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    Buf q, partial, out_s, out_r, in_s, in_r, qn, gthr, mscratch, qsplit, ckeys, flag, nrm, fbq, fb_s, fb_r, fb_i, mm_q, mm_s, mm_r, mm_p, gcand;
    std::vector<hipEvent_t> lev;   // per-launch events of the screening ladder
    int ensure_events(int n) {
        while ((int)lev.size() < n) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) return RMU_E_HIP;
            lev.push_back(e);
        }
        return RMU_OK;
    }
    int* hflag = nullptr;          // pinned landing word of the screening path's re-run count
    int* hflag_dev = nullptr;      // ... as the device sees it (written by k_gather_flagged)
    char* hpin = nullptr;          // pinned landing buffer of search_mmr_on's results (rows | scores in ONE device-to-host copy)
    size_t hpin_cap = 0;
    int ensure_hpin(size_t bytes) {
        if (bytes <= hpin_cap) return RMU_OK;
        if (hpin) (void)hipHostFree(hpin);
        hpin = nullptr; hpin_cap = 0;
        const size_t cap = bytes < 4096 ? 4096 : bytes * 2;
        if (hipHostMalloc((void**)&hpin, cap) != hipSuccess) { hpin = nullptr; (void)hipGetLastError(); return RMU_E_OOM; }
        hpin_cap = cap;
        return RMU_OK;
    }
    bool timing = false;
    float scan_ms = -1.f, search_ms = -1.f;
    int grid = 0, block = 0, lds = 0, passes = 0, screened = 0;
    int device = -1;
    // A search / merge given a caller stream and device buffers returns with its kernels still in flight -- on workspaces
    // (q, qsplit, partial, gthr, ckeys, flag, fb_*) that belong to this thread, not to that stream.  `pend_ev` marks the end
    // of the last such call: the next user of the workspaces on ANY OTHER stream is ordered behind it (same stream: stream
    // order already does that), so a second call can neither memset thresholds nor overwrite partials under running kernels.
    hipEvent_t pend_ev = nullptr;
    hipStream_t pend_stream = nullptr;
    bool pending = false;
    // `user`: the stream the caller's work will be enqueued on (nullptr = this thread's internal stream)
    int ensure_stream(hipStream_t user = nullptr) {
        if (g_device >= 0 && device != g_device) { (void)hipSetDevice(g_device); device = g_device; }
        if (!stream) {
            if (hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) != hipSuccess) return RMU_E_HIP;
            for (auto& e : ev)
                if (hipEventCreate(&e) != hipSuccess) return RMU_E_HIP;
            if (hipEventCreateWithFlags(&pend_ev, hipEventDisableTiming) != hipSuccess) return RMU_E_HIP;
        }
        hipStream_t s = user ? user : stream;
        if (pending && s != pend_stream && hipStreamWaitEvent(s, pend_ev, 0) != hipSuccess) return RMU_E_HIP;
        return RMU_OK;
    }
    // end of a call that used the workspaces on stream s: either it synchronised s (everything pending was ordered before
    // its own work, so nothing is in flight any more) or it leaves work in flight there
    void finished(hipStream_t s, bool drained) {
        if (drained) { pending = false; return; }
        if (hipEventRecord(pend_ev, s) == hipSuccess) { pend_stream = s; pending = true; }
        else { (void)hipStreamSynchronize(s); pending = false; }
    }
    // A server that spawns a thread per request (Flask's threaded dev server, server/server.py:394) creates one of
    // these per request: everything it owns goes back when the thread exits.
    ~Tls() {
        if (g_runtime_down) return;
        RMU_ENTRY();
        if (pending) (void)hipEventSynchronize(pend_ev);
        if (pend_ev) (void)hipEventDestroy(pend_ev);
        if (stream) (void)hipStreamSynchronize(stream);
        for (Buf* b : {&q, &partial, &out_s, &out_r, &in_s, &in_r, &qn, &gthr, &mscratch, &qsplit, &ckeys, &flag, &nrm, &fbq, &fb_s,
                       &fb_r, &fb_i, &mm_q, &mm_s, &mm_r, &mm_p, &gcand})
            b->release();
        for (auto& e : ev)
            if (e) (void)hipEventDestroy(e);
        for (auto& e : lev) (void)hipEventDestroy(e);
        if (hflag) (void)hipHostFree(hflag);
        if (hpin) (void)hipHostFree(hpin);
        hflag = nullptr; hflag_dev = nullptr; hpin = nullptr; hpin_cap = 0;
        if (stream) (void)hipStreamDestroy(stream);
        stream = nullptr;
    }
};
static thread_local Tls g_tls;

extern "C" int rmu_last_screened(void) { return g_tls.screened; }
extern "C" int rmu_set_timing(int on) { g_tls.timing = on != 0; return RMU_OK; }
extern "C" float rmu_last_scan_ms(void) { return g_tls.scan_ms; }
extern "C" float rmu_last_search_ms(void) { return g_tls.search_ms; }
extern "C" int rmu_last_scan_geometry(int* grid, int* block, int* lds_bytes, int* passes) {
    if (grid) *grid = g_tls.grid;
    if (block) *block = g_tls.block;
    if (lds_bytes) *lds_bytes = g_tls.lds;
    if (passes) *passes = g_tls.passes;
    return RMU_OK;
}

// ------------------------------------------------------------------------------------------------
// small kernels
// ------------------------------------------------------------------------------------------------
__global__ void k_poison_rows(float* x, int dpad, const int64_t* rows, int64_t n) {
    const int64_t r = blockIdx.x;
    if (r >= n) return;
    float* row = x + rows[r] * (int64_t)dpad;
    for (int c = threadIdx.x; c < dpad; c += blockDim.x) row[c] = __builtin_nanf("");
}

// one wave per row: x <- x / max(|x|, 1e-12) (rows of `dpad` floats, pad columns are zero);
// optionally also writes |x|^2 (before normalisation) to norm2[r]
__global__ void k_row_norm(float* x, int dpad, int64_t n, int normalise, float* norm2) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= n) return;
    float* row = x + r * (int64_t)dpad;
    float s = 0.f;
    for (int c = lane; c < dpad; c += 64) s = fmaf(row[c], row[c], s);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (norm2 && lane == 0) norm2[r] = s;
    if (normalise) {
        const float inv = 1.0f / fmaxf(sqrtf(s), 1e-12f);
        for (int c = lane; c < dpad; c += 64) row[c] *= inv;
    }
}

// RMU_METRIC_L2SQ: the scan ranks by inner product, and argmin |q - x|^2 = argmax (2 q.x - |x|^2).  Rows carry -|x|^2 in
// the first pad column (column `dim`; the L2 index pads dim + 1), queries are stored as (2q, 1): the scan's inner product
// is then 2 q.x - |x|^2 and the merge reports |q|^2 minus it.  One wave per row.
__global__ void k_l2_aug_rows(float* x, int dpad, int dim, int64_t n) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= n) return;
    float* row = x + r * (int64_t)dpad;
    float s = 0.f;
    for (int c = lane; c < dim; c += 64) s = fmaf(row[c], row[c], s);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) row[dim] = -s;
}
// queries (already copied into the dpad-wide buffer): qn2[r] = |q|^2, q <- 2q, column dim <- 1
__global__ void k_l2_aug_queries(float* q, int dpad, int dim, int64_t n, float* qn2) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= n) return;
    float* row = q + r * (int64_t)dpad;
    float s = 0.f;
    for (int c = lane; c < dim; c += 64) { const float v = row[c]; s = fmaf(v, v, s); row[c] = 2.0f * v; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) { qn2[r] = s; row[dim] = 1.0f; }
}

// L2 index with a screening image: nrm[r] = -2048 |x_r|^2 (the accumulator start of scan_screen_lean3_kernel's L2 form: the stored -|x|^2
// column times 2^11, exact) and norm2[r] = |x_r|^2 for the |x|max statistic
__global__ void k_l2_nrm(const float* __restrict__ x, int dpad, int dim, int64_t n, float* __restrict__ nrm, float* __restrict__ norm2) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const float v = x[r * (int64_t)dpad + dim];
    if (nrm) nrm[r] = 2048.0f * v;
    norm2[r] = -v;
}

// max over rows of |x|^2 (non-negative floats order like their bit patterns)
__global__ void k_max_norm2(const float* __restrict__ norm2, int64_t n, unsigned* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    float v = i < n ? norm2[i] : 0.f;
    if (!(v == v)) v = 0.f;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(v));
}

__global__ void k_gather_rows(const float* x, int dpad, int dim, const int64_t* rows, int64_t n, float* out) {
    const int64_t r = blockIdx.x;
    if (r >= n) return;
    const float* row = x + rows[r] * (int64_t)dpad;
    for (int c = threadIdx.x; c < dim; c += blockDim.x) out[r * dim + c] = row[c];
}

// the queries the screening path flagged (list `pos`, count *count) -> a dense [count, dpad] block for the exact re-run
// (round 6, second session) ... and the two other things the conditional re-runs need, in the same launch (every dependent stream operation
// costs ~4.5 us of kernel boundary behind a search): the shared thresholds of the re-run launches zeroed (was a memset), the count landed in
// pinned host memory for rmu_last_screened (was a device-to-host copy at the end).  gather_n = 0: a batch of <= 32 queries re-runs as a whole.
__global__ void k_gather_flagged(const float* __restrict__ q, int dpad, const int64_t* __restrict__ pos, const int* __restrict__ count,
                                 float* __restrict__ out, int gather_n, u32* __restrict__ zero_words, int n_zero, int* __restrict__ host_count) {
    const int64_t r = blockIdx.x;
    const int cnt = *count;
    if (r == 0) {
        for (int i = threadIdx.x; i < n_zero; i += blockDim.x) zero_words[i] = 0u;
        if (threadIdx.x == 0 && host_count) *host_count = cnt;
    }
    if (r >= gather_n || r >= cnt) return;
    const float* row = q + pos[r] * (int64_t)dpad;
    for (int c = threadIdx.x; c < dpad; c += blockDim.x) out[r * dpad + c] = row[c];
}

// Batched greedy MMR (langchain_core `maximal_marginal_relevance`, the reference retriever's search_type="mmr",
// server/RAGHelper.py:497-499): one wave per query, lane i = candidate i (fetch_k <= 64), fp64 like the numpy original.
//   first pick = argmax_i cos(q, x_i); then repeatedly argmax_i  lambda*cos(q, x_i) - (1-lambda)*max_{s picked} cos(x_i, x_s),
//   strict '>' so the lowest index wins ties; cos = dot / (|a||b|), NaN/inf -> 0.
// rows: [nq, fetch_k] index-local row ids (-1 = absent, as rmu_index_search pads them); out_pos: [nq, k] positions in the
// candidate list, -1 past the number of candidates.
__global__ __launch_bounds__(256) void k_mmr(const float* __restrict__ x, int dpad, int dim, int64_t n_rows,
                                             const float* __restrict__ q, const int64_t* __restrict__ rows, int64_t nq,
                                             int fetch_k, int k, double lambda, int* __restrict__ out_pos) {
    const int lane = threadIdx.x & 63;
    const int64_t qi = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (qi >= nq) return;
    const int64_t row = lane < fetch_k ? rows[qi * fetch_k + lane] : -1;
    const bool valid = row >= 0 && row < n_rows;
    const float* xr = x + (valid ? row : 0) * (int64_t)dpad;
    const float* qv = q + qi * dim;
    double dq = 0.0, nx = 0.0, nqq = 0.0;
    for (int c = 0; c < dim; ++c) {
        const double a = (double)xr[c], b = (double)qv[c];
        dq = fma(a, b, dq);
        nx = fma(a, a, nx);
        nqq = fma(b, b, nqq);
    }
    nx = sqrt(nx); nqq = sqrt(nqq);
    double sim_q = dq / (nx * nqq);
    if (!(sim_q == sim_q) || isinf(sim_q)) sim_q = 0.0;
    const int n_valid = __builtin_popcountll(__ballot(valid));
    const int want = k < n_valid ? k : n_valid;
    // wave argmax with the lowest lane on ties
    auto argmax = [&](double v, bool ok) -> int {
        double m = ok ? v : -INFINITY;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o));
        const unsigned long long eq = __ballot(ok && v == m);
        return eq ? __builtin_ctzll(eq) : -1;
    };
    bool picked = false;
    double red = -INFINITY;          // max cosine to the picked set
    int sel = argmax(sim_q, valid);
    for (int t = 0; t < want && sel >= 0; ++t) {
        if (lane == 0) out_pos[qi * k + t] = sel;
        if (lane == sel) picked = true;
        if (t + 1 == want) break;
        // cosine of every candidate to the newly picked one
        const int64_t srow = __shfl(row, sel);
        const double ns = __shfl(nx, sel);
        const float* xs = x + srow * (int64_t)dpad;
        double d = 0.0;
        for (int c = 0; c < dim; ++c) d = fma((double)xr[c], (double)xs[c], d);
        double cs = d / (nx * ns);
        if (!(cs == cs) || isinf(cs)) cs = 0.0;
        red = fmax(red, cs);
        sel = argmax(lambda * sim_q - (1.0 - lambda) * red, valid && !picked);
    }
    for (int t = want + lane; t < k; t += 64) out_pos[qi * k + t] = -1;
}

// The same selection for ONE query per workgroup with the candidates staged in LDS and their GRAM MATRIX computed up front by all 256
// threads (round 4).  k_mmr's lane walks its own row in global memory one float at a time (a dependent load per fp64 fma: 0.32 ms for
// the reference's per-request call, 20 candidates, 10 picks); round 3 staged the rows in LDS and ran the unchanged chain on wave 0 --
// still one 384-step fp64 chain per pick and lane, 58 us.  Every cosine the greedy loop can ask for is an entry of G = X X^T, so:
//   1. all threads copy the <= 64 rows (coalesced) into LDS rows of dim + 1 floats (odd stride: conflict-free column walks) and q;
//   2. one task per thread: G[i][j] for i <= j, q . x_i, q . q -- fp64 fma over the dims, FOUR interleaved partial chains
//      ((s0 + s1) + (s2 + s3): the chain length, not the issue rate, bounded the old kernel); 231 tasks for 20 candidates = one round;
//   3. wave 0 runs the greedy loop on look-ups: cos(x_i, x_sel) = G[i][sel] / (|x_i| |x_sel|).
// Same rule and tie order as k_mmr (langchain's); the fp64 sums differ from k_mmr's in the last bits only (summation order).
// dim <= 384: 98.6 KiB of rows + 32.5 KiB of G.
__global__ __launch_bounds__(256) void k_mmr_lds(const float* __restrict__ x, int dpad, int dim, int64_t n_rows,
                                                 const float* __restrict__ q, const int64_t* __restrict__ rows, int64_t nq,
                                                 int fetch_k, int k, double lambda, int* __restrict__ out_pos) {
    extern __shared__ __attribute__((aligned(16))) float mm[];
    const int ld = dim + 1;
    float* qs = mm + 64 * ld;
    double* G = (double*)(mm + 64 * 385 + 384);               // [65][64]: rows 0..63 = X X^T (upper triangle), row 64 = q . x_j; then q . q
    const int64_t qi = blockIdx.x;
    for (int i = threadIdx.x; i < fetch_k * dim; i += 256) {
        const int r = i / dim, c = i - r * dim;
        const int64_t row = rows[qi * fetch_k + r];
        mm[r * ld + c] = (row >= 0 && row < n_rows) ? x[row * (int64_t)dpad + c] : 0.f;
    }
    for (int c = threadIdx.x; c < dim; c += 256) qs[c] = q[qi * dim + c];
    __syncthreads();
    const int n = fetch_k;
    const int n_tri = n * (n + 1) / 2, n_tasks = n_tri + n + 1;
    for (int p = threadIdx.x; p < n_tasks; p += 256) {
        const float *a, *b;
        double* dst;
        if (p < n_tri) {
            int i = 0, rem = p;
            while (rem >= n - i) { rem -= n - i; ++i; }
            const int j = i + rem;
            a = mm + i * ld; b = mm + j * ld; dst = G + i * 64 + j;
        } else if (p < n_tri + n) {
            const int j = p - n_tri;
            a = qs; b = mm + j * ld; dst = G + 64 * 64 + j;
        } else {
            a = qs; b = qs; dst = G + 65 * 64;
        }
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        int c = 0;
        for (; c + 4 <= dim; c += 4) {
            s0 = fma((double)a[c], (double)b[c], s0);
            s1 = fma((double)a[c + 1], (double)b[c + 1], s1);
            s2 = fma((double)a[c + 2], (double)b[c + 2], s2);
            s3 = fma((double)a[c + 3], (double)b[c + 3], s3);
        }
        for (; c < dim; ++c) s0 = fma((double)a[c], (double)b[c], s0);
        *dst = (s0 + s1) + (s2 + s3);
    }
    __syncthreads();
    if (threadIdx.x >= 64) return;
    const int lane = threadIdx.x;
    const int64_t row = lane < fetch_k ? rows[qi * fetch_k + lane] : -1;
    const bool valid = row >= 0 && row < n_rows;
    const int li = lane < fetch_k ? lane : 0;
    const double nx = sqrt(G[li * 64 + li]), nqq = sqrt(G[65 * 64]);
    double sim_q = G[64 * 64 + li] / (nx * nqq);
    if (!(sim_q == sim_q) || isinf(sim_q)) sim_q = 0.0;
    const int n_valid = __builtin_popcountll(__ballot(valid));
    const int want = k < n_valid ? k : n_valid;
    auto argmax = [&](double v, bool ok) -> int {
        double m = ok ? v : -INFINITY;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o));
        const unsigned long long eq = __ballot(ok && v == m);
        return eq ? __builtin_ctzll(eq) : -1;
    };
    bool picked = false;
    double red = -INFINITY;
    int sel = argmax(sim_q, valid);
    for (int t = 0; t < want && sel >= 0; ++t) {
        if (lane == 0) out_pos[qi * k + t] = sel;
        if (lane == sel) picked = true;
        if (t + 1 == want) break;
        const double ns = __shfl(nx, sel);
        const double d = li <= sel ? G[li * 64 + sel] : G[sel * 64 + li];
        double cs = d / (nx * ns);
        if (!(cs == cs) || isinf(cs)) cs = 0.0;
        red = fmax(red, cs);
        sel = argmax(lambda * sim_q - (1.0 - lambda) * red, valid && !picked);
    }
    for (int t = want + lane; t < k; t += 64) out_pos[qi * k + t] = -1;
}

// picks -> result rows / scores of rmu_index_search_mmr: out[qi, t] = candidate list entry pos[qi, t] (+ row_base), -1 where pos is -1
__global__ void k_take_picks(const int* __restrict__ pos, const int64_t* __restrict__ rows, const float* __restrict__ scores, int64_t nq,
                             int fetch_k, int k, int64_t row_base, int64_t* __restrict__ out_rows, float* __restrict__ out_scores) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq * k) return;
    const int64_t qi = i / k;
    const int p = pos ? pos[i] : (int)(i - qi * k);      // pos == nullptr: no selection, the first k candidates in score order
    const int64_t r = p >= 0 ? rows[qi * fetch_k + p] : -1;
    out_rows[i] = r >= 0 ? r + row_base : -1;
    out_scores[i] = p >= 0 ? scores[qi * fetch_k + p] : -INFINITY;
}

// ------------------------------------------------------------------------------------------------
// index
// ------------------------------------------------------------------------------------------------
struct rmu_index {
    int dim = 0, dpad = 0, metric = 0;
    int64_t n = 0, cap = 0, n_live = 0;
    float* x = nullptr;
    char* split = nullptr;          // fp16(64 x) image of x (screening pass), 768 B per row; nullptr = disabled
    float* nrm = nullptr;           // RMU_METRIC_L2SQ with an image: -2048 |x|^2 per row (NaN past the last row); exists iff split does
    float xnorm_max = 0.f;          // max row norm (bounds the screening error)
    float dx_max = 0.f;             // max row norm of (x - screening image): the measured rounding error
    unsigned stat_host[2] = {0, 0}; // landing pair of update_image_stats (|x|^2 max, |dx|^2 max of the rows just added)
    bool stat_pending = false;
    int64_t grow_count = 0;         // re-allocations of the corpus matrix (+ image) by rmu_index_add, and their wall time
    double grow_ms = 0.0;
    bool screen_enabled = true;     // RMU_OPT_SCREEN: searches may take the screening path (when `split` exists)
    int ladder_ratio = 0, ladder_first = 0;   // RMU_OPT_LADDER_RATIO / _FIRST (0 = defaults; tools/ladder_sweep.py)
    int64_t screen_min_nq = 0;      // RMU_OPT_SCREEN_MIN_NQ: > 0 = screen every batch of at least this many queries, whatever the corpus size
    std::vector<uint8_t> alive;
    std::shared_mutex mu;
    // Streams on which scans of this index may still be running AFTER their call returned (a search handed a caller stream and device
    // buffers only enqueues: rmu_index_search's `drained == false`).  One event per stream, re-recorded behind each such search.  A
    // writer (growth, row removal, free) holds `mu` exclusively -- nothing new can start -- and waits for exactly these events instead of
    // the whole device (round 6: hipDeviceSynchronize breaks hipGraph captures of other threads, rmu_common.h).
    struct Reader { hipStream_t s; hipEvent_t ev; };
    std::vector<Reader> readers;
    std::mutex readers_mu;
};

// (shared lock held) the scans just enqueued on `s` stay in flight behind the caller's back
static void mark_reader(rmu_index* idx, hipStream_t s) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess) { (void)hipGetLastError(); return; }
    if (st != hipStreamCaptureStatusNone) return;   // a search captured into the caller's graph: replays are the caller's to order against writers
    std::lock_guard<std::mutex> g(idx->readers_mu);
    for (auto& r : idx->readers)
        if (r.s == s) { (void)hipEventRecord(r.ev, s); return; }
    if (idx->readers.size() >= 64) {                // streams come and go (a thread per request): drop the marks whose work has ended
        size_t w = 0;
        for (auto& r : idx->readers) {
            if (hipEventQuery(r.ev) == hipSuccess) (void)hipEventDestroy(r.ev);
            else idx->readers[w++] = r;
        }
        (void)hipGetLastError();
        idx->readers.resize(w);
    }
    hipEvent_t ev = nullptr;
    if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); (void)hipStreamSynchronize(s); return; }
    if (hipEventRecord(ev, s) != hipSuccess) { (void)hipGetLastError(); (void)hipEventDestroy(ev); (void)hipStreamSynchronize(s); return; }
    idx->readers.push_back({s, ev});
}
// (exclusive lock held) host-side wait for every scan still in flight; `on` != nullptr: order stream `on` behind them instead (no host wait)
static int wait_readers(rmu_index* idx, hipStream_t on = nullptr) {
    std::lock_guard<std::mutex> g(idx->readers_mu);
    for (auto& r : idx->readers) {
        const hipError_t e = on ? hipStreamWaitEvent(on, r.ev, 0) : hipEventSynchronize(r.ev);
        if (e != hipSuccess) return fail(RMU_E_HIP, std::string("waiting for the scans in flight: ") + hipGetErrorString(e));
    }
    return RMU_OK;
}

// The scans' LDS-DMA rings run past the last row: the exact scan reads whole 128-row tiles; the screening scan (scan_screen_lean3_kernel)
// looks four 32-row tiles ahead of the tile it computes and does not clamp, i.e. it touches up to (ceil(n / 32) + 4) * 32 - n <= 159 rows
// past row n of the image.  Keep this many allocated (zero-filled) rows past the capacity of both matrices.
static const int64_t kSlackRows = 256;
static_assert(kSlackRows >= 160, "scan_screen_lean3_kernel's unclamped look-ahead");
static int pad_dim(int d) { return d <= 192 ? 192 : (d <= 384 ? 384 : (d <= 768 ? 768 : -1)); }
// L2SQ rows carry -|x|^2 in one extra column (see k_l2_aug_rows)
static int pad_dim_metric(int d, int metric) { return pad_dim(metric == RMU_METRIC_L2SQ ? d + 1 : d); }

extern "C" int rmu_index_create(rmu_index_t** out, int dim, int metric, int64_t capacity_hint) {
    RMU_ENTRY();
    if (!out) return fail(RMU_E_INVALID, "rmu_index_create: out is null");
    if (dim < 1 || dim > RMU_MAX_DIM) return fail(RMU_E_INVALID, "rmu_index_create: dim must be in [1, 768]");
    if (metric != RMU_METRIC_IP && metric != RMU_METRIC_COSINE && metric != RMU_METRIC_L2SQ)
        return fail(RMU_E_INVALID, "rmu_index_create: metric must be RMU_METRIC_IP, RMU_METRIC_COSINE or RMU_METRIC_L2SQ");
    if (pad_dim_metric(dim, metric) < 0)
        return fail(RMU_E_INVALID, "rmu_index_create: RMU_METRIC_L2SQ needs one spare column: dim must be <= 767");
    int rc = g_tls.ensure_stream();
    if (rc) return fail(rc, "rmu_index_create: stream");
    auto* idx = new (std::nothrow) rmu_index();
    if (!idx) return fail(RMU_E_OOM, "rmu_index_create: host alloc");
    idx->dim = dim;
    idx->dpad = pad_dim_metric(dim, metric);
    idx->metric = metric;
    int64_t cap = capacity_hint > 0 ? capacity_hint : 4096;
    hipError_t e = hipMalloc((void**)&idx->x, (size_t)(cap + kSlackRows) * idx->dpad * sizeof(float));
    if (e != hipSuccess) { delete idx; return fail(RMU_E_OOM, "rmu_index_create: hipMalloc"); }
    // zero fill ON OUR STREAM and wait: a null-stream hipMemset is asynchronous to the host and is not ordered
    // with a hipStreamNonBlocking stream, so it could land after (and wipe) the first rows uploaded on it
    if (hipMemsetAsync(idx->x, 0, (size_t)(cap + kSlackRows) * idx->dpad * sizeof(float), g_tls.stream) != hipSuccess ||
        hipStreamSynchronize(g_tls.stream) != hipSuccess) {
        (void)hipFree(idx->x);
        delete idx;
        return fail(RMU_E_HIP, "rmu_index_create: zero fill");
    }
    idx->cap = cap;
    // screening image (+50% corpus memory): 384-wide rows only; RMU_SCREEN=0 disables
    static const bool screen_on = !(rmu_env_kill("RMU_SCREEN") && atoi(rmu_env_kill("RMU_SCREEN")) == 0);
    // (round 5) ... and the native L2 index at dim 384 (rows 768 floats apart: 384 + the -|x|^2 column, padded), with the row norms beside it
    const bool l2 = metric == RMU_METRIC_L2SQ;
    if (screen_on && dim == 384 && (idx->dpad == 384 || l2)) {
        if (hipMalloc((void**)&idx->split, (size_t)(cap + kSlackRows) * RMU_IMG_ROW_BYTES) != hipSuccess) {
            idx->split = nullptr;   // not fatal: exact path only
            (void)hipGetLastError();
        } else if (l2 && hipMalloc((void**)&idx->nrm, (size_t)(cap + kSlackRows) * sizeof(float)) != hipSuccess) {
            (void)hipGetLastError();
            (void)hipFree(idx->split);
            idx->split = nullptr; idx->nrm = nullptr;
        } else {
            (void)hipMemsetAsync(idx->split, 0, (size_t)(cap + kSlackRows) * RMU_IMG_ROW_BYTES, g_tls.stream);
            if (idx->nrm) (void)hipMemsetAsync(idx->nrm, 0xFF, (size_t)(cap + kSlackRows) * sizeof(float), g_tls.stream);   // NaN: never a candidate
            (void)hipStreamSynchronize(g_tls.stream);
        }
    }
    *out = idx;
    return RMU_OK;
}

extern "C" int rmu_index_free(rmu_index_t* idx) {
    RMU_ENTRY();
    if (!idx) return RMU_OK;
    {
        std::unique_lock<std::shared_mutex> lk(idx->mu);
        (void)wait_readers(idx);          // scans still in flight on callers' streams (everything else ended under the shared lock)
        if (idx->x) (void)hipFree(idx->x);
        if (idx->split) (void)hipFree(idx->split);
        if (idx->nrm) (void)hipFree(idx->nrm);
        idx->x = nullptr;
        idx->split = nullptr;
        idx->nrm = nullptr;
        for (auto& r : idx->readers) (void)hipEventDestroy(r.ev);
        idx->readers.clear();
    }
    delete idx;
    return RMU_OK;
}

extern "C" int rmu_index_size(rmu_index_t* idx, int64_t* n_rows) {
    if (!idx || !n_rows) return fail(RMU_E_INVALID, "rmu_index_size: null");
    std::shared_lock<std::shared_mutex> lk(idx->mu);
    *n_rows = idx->n;
    return RMU_OK;
}
extern "C" int rmu_index_stat(rmu_index_t* idx, int what, double* out) {
    if (!idx || !out) return fail(RMU_E_INVALID, "rmu_index_stat: null");
    std::shared_lock<std::shared_mutex> lk(idx->mu);
    switch (what) {
        case RMU_STAT_CAPACITY: *out = (double)idx->cap; break;
        case RMU_STAT_GROW_COUNT: *out = (double)idx->grow_count; break;
        case RMU_STAT_GROW_MS: *out = idx->grow_ms; break;
        case RMU_STAT_LIVE_ROWS: *out = (double)idx->n_live; break;
        default: return fail(RMU_E_INVALID, "rmu_index_stat: unknown statistic");
    }
    return RMU_OK;
}
extern "C" int rmu_index_dim(rmu_index_t* idx, int* dim) {
    if (!idx || !dim) return fail(RMU_E_INVALID, "rmu_index_dim: null");
    *dim = idx->dim;
    return RMU_OK;
}

static int grow(rmu_index* idx, int64_t need) {
    if (need <= idx->cap) return RMU_OK;
    const auto t0 = std::chrono::steady_clock::now();
    struct Tick {
        rmu_index* i; std::chrono::steady_clock::time_point t;
        ~Tick() { i->grow_count++; i->grow_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count(); }
    } tick{idx, t0};
    int64_t cap = idx->cap;
    while (cap < need) cap = cap + cap / 2 + 1024;
    float* nx = nullptr;
    if (hipMalloc((void**)&nx, (size_t)(cap + kSlackRows) * idx->dpad * sizeof(float)) != hipSuccess) {
        cap = need;  // retry with the exact size before giving up
        if (hipMalloc((void**)&nx, (size_t)(cap + kSlackRows) * idx->dpad * sizeof(float)) != hipSuccess)
            return fail(RMU_E_OOM, "rmu_index_add: hipMalloc for growth");
    }
    // nobody may still be scanning the old matrix when it is freed below: searches on internal streams ended under the shared lock, the
    // ones left in flight on callers' streams are waited for one by one -- never the whole device (rmu_common.h: captures)
    int wrc = wait_readers(idx);
    if (wrc) { (void)rmu_free(nx); return wrc; }
    hipStream_t s = g_tls.stream;     // same stream as the uploads that follow (see rmu_index_create)
    HIP_TRY(hipMemsetAsync(nx + idx->n * idx->dpad, 0, (size_t)(cap + kSlackRows - idx->n) * idx->dpad * sizeof(float), s));
    if (idx->n)
        HIP_TRY(hipMemcpyAsync(nx, idx->x, (size_t)idx->n * idx->dpad * sizeof(float), hipMemcpyDeviceToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (idx->split) {
        char* ns = nullptr;
        float* nn = nullptr;
        const size_t rowb = RMU_IMG_ROW_BYTES;
        bool got = hipMalloc((void**)&ns, (size_t)(cap + kSlackRows) * rowb) == hipSuccess;
        if (got && idx->nrm && hipMalloc((void**)&nn, (size_t)(cap + kSlackRows) * sizeof(float)) != hipSuccess) {
            (void)rmu_free(ns);
            got = false;
        }
        if (got) {
            HIP_TRY(hipMemsetAsync(ns + idx->n * rowb, 0, (size_t)(cap + kSlackRows - idx->n) * rowb, s));
            if (idx->n) HIP_TRY(hipMemcpyAsync(ns, idx->split, (size_t)idx->n * rowb, hipMemcpyDeviceToDevice, s));
            if (nn) {
                HIP_TRY(hipMemsetAsync(nn + idx->n, 0xFF, (size_t)(cap + kSlackRows - idx->n) * sizeof(float), s));
                if (idx->n) HIP_TRY(hipMemcpyAsync(nn, idx->nrm, (size_t)idx->n * sizeof(float), hipMemcpyDeviceToDevice, s));
            }
            HIP_TRY(hipStreamSynchronize(s));
            (void)rmu_free(idx->split);
            if (idx->nrm) (void)rmu_free(idx->nrm);
            idx->split = ns;
            idx->nrm = nn;
        } else {
            (void)hipGetLastError();
            (void)rmu_free(idx->split);   // no room for the screening image: exact path only from now on
            if (idx->nrm) (void)rmu_free(idx->nrm);
            idx->split = nullptr;
            idx->nrm = nullptr;
        }
    }
    (void)rmu_free(idx->x);
    idx->x = nx;
    idx->cap = cap;
    return RMU_OK;
}

extern "C" int rmu_index_reserve(rmu_index_t* idx, int64_t rows) {
    RMU_ENTRY();
    if (!idx || rows < 0) return fail(RMU_E_INVALID, "rmu_index_reserve: bad argument");
    if (rows > 0xFFFFFFF0ll) return fail(RMU_E_INVALID, "rmu_index_reserve: row ids are 32-bit inside the scan");
    int rc = g_tls.ensure_stream();
    if (rc) return fail(rc, "rmu_index_reserve: stream");
    std::unique_lock<std::shared_mutex> lk(idx->mu);
    return grow(idx, rows);           // (rmu_index_add's geometric policy: a caller that reserves round by round does not re-allocate every round)
}

// after `n` rows at `dst` were converted into the screening image: |x|max and the measured image error |dx|max of those rows.
// ENQUEUES only (two reductions into one 8-byte device pair + one 8-byte copy to idx->stat_host): the caller's own final
// stream synchronisation covers it, then fold_image_stats() folds the pair into the index -- one host round trip per
// rmu_index_add instead of three (the reference inserts in 1000-document calls, server/RAGHelper.py:423-434).
static int update_image_stats(rmu_index_t* idx, const float* dst, int64_t n, hipStream_t s, float* nrm_out = nullptr) {
    Buf& nb = g_tls.nrm;
    if (nb.ensure((size_t)n * sizeof(float) + 16)) return fail(RMU_E_OOM, "rmu_index_add: norm workspace");
    unsigned* mx = (unsigned*)((char*)nb.p + (size_t)n * sizeof(float));
    HIP_TRY(hipMemsetAsync(mx, 0, 2 * sizeof(unsigned), s));
    if (idx->metric == RMU_METRIC_L2SQ) {      // |x|^2 is the augmented column (k_l2_aug_rows), not the norm of the augmented row
        hipLaunchKernelGGL(k_l2_nrm, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dst, idx->dpad, idx->dim, n, nrm_out, (float*)nb.p);
        hipLaunchKernelGGL(k_max_norm2, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const float*)nb.p, n, mx);
    } else if (idx->metric != RMU_METRIC_COSINE) {
        hipLaunchKernelGGL(k_row_norm, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, const_cast<float*>(dst), idx->dpad, n, 0, (float*)nb.p);
        hipLaunchKernelGGL(k_max_norm2, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const float*)nb.p, n, mx);
    }
    int rc = rmu_img_err_launch(dst, n, (float*)nb.p, s, idx->dpad);
    if (rc) return fail(rc, "rmu_index_add: image error");
    hipLaunchKernelGGL(k_max_norm2, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const float*)nb.p, n, mx + 1);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(idx->stat_host, mx, 2 * sizeof(unsigned), hipMemcpyDeviceToHost, s));
    idx->stat_pending = true;
    return RMU_OK;
}

// rows [r0, r0 + n) of idx->x (final form: normalised / augmented) -> screening image (+ row norms of an L2 index) + statistics.  Enqueues only.
static int build_image(rmu_index_t* idx, int64_t r0, int64_t n, hipStream_t s) {
    if (!idx->split) return RMU_OK;
    const float* rows = idx->x + r0 * (int64_t)idx->dpad;
    int rc = rmu_split_launch(rows, idx->split + (size_t)r0 * RMU_IMG_ROW_BYTES, n, s, idx->dpad, 64.0f);
    if (rc) return fail(rc, "rmu_index_add: split image");
    return update_image_stats(idx, rows, n, s, idx->nrm ? idx->nrm + r0 : nullptr);
}

// after the stream that ran update_image_stats was synchronised
static void fold_image_stats(rmu_index_t* idx) {
    if (!idx->stat_pending) return;
    idx->stat_pending = false;
    float f;
    if (idx->metric == RMU_METRIC_COSINE) {
        idx->xnorm_max = 1.0f;
    } else {
        memcpy(&f, &idx->stat_host[0], 4);
        f = sqrtf(f);
        if (f > idx->xnorm_max) idx->xnorm_max = f;
    }
    memcpy(&f, &idx->stat_host[1], 4);
    f = sqrtf(f) * 1.0001f;
    if (f > idx->dx_max) idx->dx_max = f;
}

extern "C" int rmu_index_add(rmu_index_t* idx, const float* vecs, int64_t n, int is_device, int64_t* first_row) {
    RMU_ENTRY();
    if (!idx || (!vecs && n > 0) || n < 0) return fail(RMU_E_INVALID, "rmu_index_add: bad argument");
    int rc = g_tls.ensure_stream();
    if (rc) return fail(rc, "rmu_index_add: stream");
    std::unique_lock<std::shared_mutex> lk(idx->mu);
    if (first_row) *first_row = idx->n;
    if (n == 0) return RMU_OK;
    if (idx->n + n > 0xFFFFFFF0ll) return fail(RMU_E_INVALID, "rmu_index_add: row ids are 32-bit inside the scan");
    rc = grow(idx, idx->n + n);
    if (rc) return rc;
    float* dst = idx->x + idx->n * (int64_t)idx->dpad;
    hipStream_t s = g_tls.stream;
    const hipMemcpyKind kind = is_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    if (idx->dpad == idx->dim)
        HIP_TRY(hipMemcpyAsync(dst, vecs, (size_t)n * idx->dim * sizeof(float), kind, s));
    else
        HIP_TRY(hipMemcpy2DAsync(dst, (size_t)idx->dpad * sizeof(float), vecs, (size_t)idx->dim * sizeof(float),
                                 (size_t)idx->dim * sizeof(float), (size_t)n, kind, s));
    if (idx->metric == RMU_METRIC_COSINE) {
        const int wpb = 4;
        hipLaunchKernelGGL(k_row_norm, dim3((unsigned)((n + wpb - 1) / wpb)), dim3(64 * wpb), 0, s, dst, idx->dpad, n, 1,
                           (float*)nullptr);
        HIP_TRY(hipGetLastError());
    }
    if (idx->metric == RMU_METRIC_L2SQ) {
        hipLaunchKernelGGL(k_l2_aug_rows, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, dst, idx->dpad, idx->dim, n);
        HIP_TRY(hipGetLastError());
    }
    rc = build_image(idx, idx->n, n, s);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(s));
    fold_image_stats(idx);
    idx->alive.resize((size_t)(idx->n + n), 1);
    idx->n += n;
    idx->n_live += n;
    return RMU_OK;
}

extern "C" int rmu_index_remove_rows(rmu_index_t* idx, const int64_t* rows, int64_t n, int64_t* n_removed) {
    RMU_ENTRY();
    if (!idx || (!rows && n > 0) || n < 0) return fail(RMU_E_INVALID, "rmu_index_remove_rows: bad argument");
    int rc = g_tls.ensure_stream();
    if (rc) return fail(rc, "rmu_index_remove_rows: stream");
    std::unique_lock<std::shared_mutex> lk(idx->mu);
    std::vector<int64_t> todo;
    for (int64_t i = 0; i < n; ++i) {
        const int64_t r = rows[i];
        if (r < 0 || r >= idx->n) return fail(RMU_E_INVALID, "rmu_index_remove_rows: row out of range");
        if (idx->alive[(size_t)r]) { idx->alive[(size_t)r] = 0; todo.push_back(r); }
    }
    if (n_removed) *n_removed = (int64_t)todo.size();
    if (todo.empty()) return RMU_OK;
    Buf& b = g_tls.in_r;
    if (b.ensure(todo.size() * sizeof(int64_t))) return fail(RMU_E_OOM, "rmu_index_remove_rows: workspace");
    hipStream_t s = g_tls.stream;
    rc = wait_readers(idx, s);        // the poison kernels are ordered behind the scans still in flight on callers' streams
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(b.p, todo.data(), todo.size() * sizeof(int64_t), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_poison_rows, dim3((unsigned)todo.size()), dim3(128), 0, s, idx->x, idx->dpad,
                       (const int64_t*)b.p, (int64_t)todo.size());
    if (idx->split)   // fp32 NaN pattern = one fp16 NaN per pair: every screening score of the row is NaN as well
        hipLaunchKernelGGL(k_poison_rows, dim3((unsigned)todo.size()), dim3(128), 0, s, (float*)idx->split, RMU_IMG_ROW_BYTES / 4,
                           (const int64_t*)b.p, (int64_t)todo.size());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(s));
    idx->n_live -= (int64_t)todo.size();
    return RMU_OK;
}

extern "C" int rmu_index_get_rows(rmu_index_t* idx, const int64_t* rows, int64_t n, float* out_host) {
    RMU_ENTRY();
    if (!idx || (!rows && n > 0) || (!out_host && n > 0) || n < 0)
        return fail(RMU_E_INVALID, "rmu_index_get_rows: bad argument");
    if (n == 0) return RMU_OK;
    int rc = g_tls.ensure_stream();
    if (rc) return fail(rc, "rmu_index_get_rows: stream");
    std::shared_lock<std::shared_mutex> lk(idx->mu);
    for (int64_t i = 0; i < n; ++i)
        if (rows[i] < 0 || rows[i] >= idx->n) return fail(RMU_E_INVALID, "rmu_index_get_rows: row out of range");
    Buf& br = g_tls.in_r;
    Buf& bo = g_tls.out_s;
    if (br.ensure((size_t)n * sizeof(int64_t)) || bo.ensure((size_t)n * idx->dim * sizeof(float)))
        return fail(RMU_E_OOM, "rmu_index_get_rows: workspace");
    hipStream_t s = g_tls.stream;
    HIP_TRY(hipMemcpyAsync(br.p, rows, (size_t)n * sizeof(int64_t), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_gather_rows, dim3((unsigned)n), dim3(128), 0, s, idx->x, idx->dpad, idx->dim,
                       (const int64_t*)br.p, n, (float*)bo.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out_host, bo.p, (size_t)n * idx->dim * sizeof(float), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return RMU_OK;
}

static void launch_mmr(const rmu_index* idx, const float* dq, const int64_t* dr, int64_t nq, int fetch_k, int k, double lambda_mult, int* dout,
                       hipStream_t s) {
    static const bool lds_off = rmu_env("RMU_MMR_LDS") && atoi(rmu_env("RMU_MMR_LDS")) == 0;
    if (idx->dim <= 384 && nq <= 65535 && !lds_off) {          // one workgroup per query, candidates staged in LDS (bit-identical picks)
        // rows [64][dim + 1] + q [dim] at the head of a (64 * 385 + 384)-float area, then the fp64 Gram block [65][64] + 1
        const size_t lds = (size_t)(64 * 385 + 384) * sizeof(float) + (size_t)(65 * 64 + 1) * sizeof(double);
        static const hipError_t attr_rc = hipFuncSetAttribute((const void*)k_mmr_lds, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                              (64 * 385 + 384) * 4 + (65 * 64 + 1) * 8);
        (void)attr_rc;
        hipLaunchKernelGGL(k_mmr_lds, dim3((unsigned)nq), dim3(256), lds, s, idx->x, idx->dpad, idx->dim, idx->n, dq, dr, nq, fetch_k, k, lambda_mult, dout);
    } else {
        hipLaunchKernelGGL(k_mmr, dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, s, idx->x, idx->dpad, idx->dim, idx->n, dq, dr, nq, fetch_k,
                           k, lambda_mult, dout);
    }
}

extern "C" int rmu_index_mmr(rmu_index_t* idx, const float* q, int64_t nq, const int64_t* rows, int fetch_k, int k,
                             double lambda_mult, unsigned flags, int32_t* out_pos) {
    RMU_ENTRY();
    if (!idx || !q || !rows || !out_pos) return fail(RMU_E_INVALID, "rmu_index_mmr: null pointer");
    if (nq < 1 || fetch_k < 1 || fetch_k > 64 || k < 1) return fail(RMU_E_INVALID, "rmu_index_mmr: nq >= 1, fetch_k in [1, 64], k >= 1");
    Tls& t = g_tls;
    int rc = t.ensure_stream();
    if (rc) return fail(rc, "rmu_index_mmr: stream");
    hipStream_t s = t.stream;
    const bool q_dev = flags & RMU_F_Q_DEVICE, io_dev = flags & RMU_F_OUT_DEVICE;
    std::shared_lock<std::shared_mutex> lk(idx->mu);
    const float* dq = q;
    const int64_t* dr = rows;
    int* dout = (int*)out_pos;
    if (!q_dev) {
        if (t.q.ensure((size_t)nq * idx->dim * sizeof(float))) return fail(RMU_E_OOM, "rmu_index_mmr: q workspace");
        HIP_TRY(hipMemcpyAsync(t.q.p, q, (size_t)nq * idx->dim * sizeof(float), hipMemcpyHostToDevice, s));
        dq = (const float*)t.q.p;
    }
    if (!io_dev) {
        if (t.in_r.ensure((size_t)nq * fetch_k * sizeof(int64_t)) || t.out_r.ensure((size_t)nq * k * sizeof(int)))
            return fail(RMU_E_OOM, "rmu_index_mmr: workspace");
        HIP_TRY(hipMemcpyAsync(t.in_r.p, rows, (size_t)nq * fetch_k * sizeof(int64_t), hipMemcpyHostToDevice, s));
        dr = (const int64_t*)t.in_r.p;
        dout = (int*)t.out_r.p;
    }
    launch_mmr(idx, dq, dr, nq, fetch_k, k, lambda_mult, dout, s);
    HIP_TRY(hipGetLastError());
    if (!io_dev) HIP_TRY(hipMemcpyAsync(out_pos, dout, (size_t)nq * k * sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return RMU_OK;
}

// dense top-fetch_k + greedy selection of `nq` queries (host or device fp32 [nq, dim]) on stream s; HOST results; drains s.
// lambda_mult < 0: no selection -- the top-k (k <= fetch_k) in score order, as rmu_index_search returns them.
static int search_mmr_on(rmu_index_t* idx, const float* q, bool q_dev, int64_t nq, int fetch_k, int k, double lambda_mult, int64_t row_base,
                         int64_t* out_rows, float* out_scores, hipStream_t s, const char* who) {
    Tls& t = g_tls;
    const size_t nf = (size_t)nq * fetch_k, nk = (size_t)nq * k;
    // candidate scores / rows, the raw queries (the search normalises its own copy for COSINE), picks, and the results (rows | scores)
    if (t.mm_s.ensure(nf * sizeof(float)) || t.mm_r.ensure(nf * sizeof(int64_t)) || t.mm_q.ensure((size_t)nq * idx->dim * sizeof(float)) ||
        t.mm_p.ensure(nk * (sizeof(int) + sizeof(int64_t) + sizeof(float))))
        return fail(RMU_E_OOM, std::string(who) + ": workspace");
    int64_t* d_rows = (int64_t*)t.mm_p.p;
    float* d_sc = (float*)(d_rows + nk);
    int* d_pos = (int*)(d_sc + nk);
    // (round 5: every stream operation of the one-query path is ~4.4 us of dependent-kernel boundary.)  No selection and fetch_k == k: the
    // search writes the result rows (row_base applied) and scores where the copy below reads them -- no k_take_picks launch.
    const bool direct = lambda_mult < 0.0 && fetch_k == k;
    if (t.ensure_hpin(nk * (sizeof(int64_t) + sizeof(float)))) return fail(RMU_E_OOM, std::string(who) + ": pinned result buffer");
    // the search on the given stream, results left on the device (no synchronisation inside)
    int rc = direct ? rmu_index_search(idx, q, nq, k, RMU_F_OUT_DEVICE | (q_dev ? RMU_F_Q_DEVICE : 0u), row_base, d_sc, d_rows, (uint64_t)(uintptr_t)s)
                    : rmu_index_search(idx, q, nq, fetch_k, RMU_F_OUT_DEVICE | (q_dev ? RMU_F_Q_DEVICE : 0u), 0, (float*)t.mm_s.p, (int64_t*)t.mm_r.p,
                                       (uint64_t)(uintptr_t)s);
    if (rc) return rc;
    std::shared_lock<std::shared_mutex> lk(idx->mu);
    if (!direct) {
        const float* dq = q;
        if (!q_dev && lambda_mult >= 0.0) {
            HIP_TRY(hipMemcpyAsync(t.mm_q.p, q, (size_t)nq * idx->dim * sizeof(float), hipMemcpyHostToDevice, s));
            dq = (const float*)t.mm_q.p;
        }
        if (lambda_mult >= 0.0) launch_mmr(idx, dq, (const int64_t*)t.mm_r.p, nq, fetch_k, k, lambda_mult, d_pos, s);
        hipLaunchKernelGGL(k_take_picks, dim3((unsigned)((nk + 255) / 256)), dim3(256), 0, s, lambda_mult >= 0.0 ? (const int*)d_pos : (const int*)nullptr,
                           (const int64_t*)t.mm_r.p, (const float*)t.mm_s.p, nq, fetch_k, k, row_base, d_rows, d_sc);
        HIP_TRY(hipGetLastError());
    }
    // rows | scores are contiguous on the device: ONE copy into pinned memory (a copy into the caller's pageable arrays is staged by the
    // runtime, one staging round per call), split on the host behind the synchronisation
    HIP_TRY(hipMemcpyAsync(t.hpin, d_rows, nk * (sizeof(int64_t) + sizeof(float)), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    memcpy(out_rows, t.hpin, nk * sizeof(int64_t));
    if (out_scores) memcpy(out_scores, t.hpin + nk * sizeof(int64_t), nk * sizeof(float));
    t.finished(s, true);
    return RMU_OK;
}

extern "C" int rmu_index_search_mmr(rmu_index_t* idx, const float* q, int64_t nq, int fetch_k, int k, double lambda_mult, int64_t row_base,
                                    int64_t* out_rows, float* out_scores) {
    RMU_ENTRY();
    if (!idx || !q || !out_rows) return fail(RMU_E_INVALID, "rmu_index_search_mmr: null pointer");
    if (nq < 1 || fetch_k < 1 || fetch_k > 64 || k < 1 || k > fetch_k)
        return fail(RMU_E_INVALID, "rmu_index_search_mmr: nq >= 1, fetch_k in [1, 64], k in [1, fetch_k]");
    if (lambda_mult < 0.0) return fail(RMU_E_INVALID, "rmu_index_search_mmr: lambda_mult must be >= 0");
    Tls& t = g_tls;
    int rc = t.ensure_stream();
    if (rc) return fail(rc, "rmu_index_search_mmr: stream");
    return search_mmr_on(idx, q, false, nq, fetch_k, k, lambda_mult, row_base, out_rows, out_scores, t.stream, "rmu_index_search_mmr");
}

// bert.hip (rmu_bert_search_mmr): the same behind an encoder forward -- DEVICE queries on the encoder's stream
extern "C" int rmu_index_search_mmr_dev_(rmu_index_t* idx, const float* q_dev, int64_t nq, int fetch_k, int k, double lambda_mult, int64_t row_base,
                                         int64_t* out_rows, float* out_scores, void* hip_stream) {
    RMU_ENTRY();
    if (!idx || !q_dev || !out_rows || !hip_stream) return fail(RMU_E_INVALID, "rmu_bert_search_mmr: null pointer");
    if (nq < 1 || fetch_k < 1 || fetch_k > 64 || k < 1 || k > fetch_k)
        return fail(RMU_E_INVALID, "rmu_bert_search_mmr: nq >= 1, fetch_k in [1, 64], k in [1, fetch_k]");
    if (idx->dim != 384) return fail(RMU_E_INVALID, "rmu_bert_search_mmr: the index must hold 384-d rows (the encoder's width)");
    Tls& t = g_tls;
    int rc = t.ensure_stream((hipStream_t)hip_stream);
    if (rc) return fail(rc, "rmu_bert_search_mmr: stream");
    return search_mmr_on(idx, q_dev, true, nq, fetch_k, k, lambda_mult, row_base, out_rows, out_scores, (hipStream_t)hip_stream, "rmu_bert_search_mmr");
}

// ------------------------------------------------------------------------------------------------
// persistence
// ------------------------------------------------------------------------------------------------
struct RmuFileHeader {
    char magic[8];          // "RMUIDX01"
    int32_t dim, dpad, metric, reserved;
    int64_t n, n_live;
    float xnorm_max;
    char pad[64 - 8 - 16 - 16 - 4];
};
static_assert(sizeof(RmuFileHeader) == 64, "header");

extern "C" int rmu_index_save(rmu_index_t* idx, const char* path) {
    RMU_ENTRY();
    if (!idx || !path) return fail(RMU_E_INVALID, "rmu_index_save: null argument");
    int rc = g_tls.ensure_stream();
    if (rc) return fail(rc, "rmu_index_save: stream");
    std::shared_lock<std::shared_mutex> lk(idx->mu);
    FILE* f = fopen(path, "wb");
    if (!f) return fail(RMU_E_INVALID, std::string("rmu_index_save: cannot open ") + path);
    RmuFileHeader h{};
    memcpy(h.magic, "RMUIDX01", 8);
    h.dim = idx->dim; h.dpad = idx->dpad; h.metric = idx->metric; h.n = idx->n; h.n_live = idx->n_live; h.xnorm_max = idx->xnorm_max;
    bool ok = fwrite(&h, sizeof(h), 1, f) == 1;
    if (idx->n) ok = ok && fwrite(idx->alive.data(), 1, (size_t)idx->n, f) == (size_t)idx->n;
    const size_t rowb = (size_t)idx->dpad * sizeof(float);
    const int64_t chunk = 65536;                       // rows per staging copy (<= 192 MiB)
    std::vector<char> host(ok ? (size_t)std::min<int64_t>(chunk, std::max<int64_t>(idx->n, 1)) * rowb : 0);
    for (int64_t r0 = 0; ok && r0 < idx->n; r0 += chunk) {
        const int64_t nr = std::min<int64_t>(chunk, idx->n - r0);
        if (hipMemcpyAsync(host.data(), (const char*)idx->x + r0 * rowb, (size_t)nr * rowb, hipMemcpyDeviceToHost, g_tls.stream) != hipSuccess ||
            hipStreamSynchronize(g_tls.stream) != hipSuccess) {
            fclose(f);
            return fail(RMU_E_HIP, "rmu_index_save: device read");
        }
        ok = fwrite(host.data(), rowb, (size_t)nr, f) == (size_t)nr;
    }
    ok = (fclose(f) == 0) && ok;
    return ok ? RMU_OK : fail(RMU_E_INVALID, std::string("rmu_index_save: write failed: ") + path);
}

extern "C" int rmu_index_load(rmu_index_t** out, const char* path) {
    RMU_ENTRY();
    if (!out || !path) return fail(RMU_E_INVALID, "rmu_index_load: null argument");
    FILE* f = fopen(path, "rb");
    if (!f) return fail(RMU_E_INVALID, std::string("rmu_index_load: cannot open ") + path);
    RmuFileHeader h{};
    if (fread(&h, sizeof(h), 1, f) != 1 || memcmp(h.magic, "RMUIDX01", 8) != 0 || h.n < 0 || h.dim < 1 ||
        h.dpad != pad_dim_metric(h.dim, h.metric)) {
        fclose(f);
        return fail(RMU_E_INVALID, std::string("rmu_index_load: not an RMUIDX01 file: ") + path);
    }
    rmu_index_t* idx = nullptr;
    int rc = rmu_index_create(&idx, h.dim, h.metric, h.n > 0 ? h.n : 4096);
    if (rc) { fclose(f); return rc; }
    idx->alive.assign((size_t)h.n, 1);
    bool ok = h.n == 0 || fread(idx->alive.data(), 1, (size_t)h.n, f) == (size_t)h.n;
    const size_t rowb = (size_t)h.dpad * sizeof(float);
    const int64_t chunk = 65536;
    std::vector<char> host(ok ? (size_t)std::min<int64_t>(chunk, std::max<int64_t>(h.n, 1)) * rowb : 0);
    hipStream_t s = g_tls.stream;
    for (int64_t r0 = 0; ok && r0 < h.n; r0 += chunk) {
        const int64_t nr = std::min<int64_t>(chunk, h.n - r0);
        ok = fread(host.data(), rowb, (size_t)nr, f) == (size_t)nr;
        if (ok && hipMemcpyAsync((char*)idx->x + r0 * rowb, host.data(), (size_t)nr * rowb, hipMemcpyHostToDevice, s) != hipSuccess) ok = false;
        if (ok) ok = build_image(idx, r0, nr, s) == RMU_OK;
        if (ok && hipStreamSynchronize(s) != hipSuccess) ok = false;
        if (ok) fold_image_stats(idx);
    }
    fclose(f);
    if (!ok) { rmu_index_free(idx); return fail(RMU_E_INVALID, std::string("rmu_index_load: truncated or unreadable: ") + path); }
    idx->n = h.n; idx->n_live = h.n_live;
    if (h.xnorm_max > idx->xnorm_max) idx->xnorm_max = h.xnorm_max;
    *out = idx;
    return RMU_OK;
}

// ------------------------------------------------------------------------------------------------
// search
// ------------------------------------------------------------------------------------------------
static const int64_t kMaxQueriesPerLaunch = 8192;
static const int kScreenKp = 32;     // K': candidates the screening pass keeps per query (k <= 24)
// (round 5) 24 < k <= 32: K' = 40 -- the same eight spare candidates behind the k-th for the sufficiency test; the slots hold RMU_KS_CAP = 48
// (round 6) 32 < k <= 104 (BASELINE config 5's dense top-100): K' = k + max(8, k / 5), at most 120 -- the sufficiency test needs the K'-th approximate
// score 2 EPS below the k-th, and the order statistics crowd together as k grows (16..20 spare candidates at k = 100; 8 sufficed at k <= 32)
static const int kScreenMaxK = 104;
static inline int screen_kp(int k) {
    if (k <= 24) return kScreenKp;
    if (k <= 32) return 40;
    const int kp = k + (k / 5 > 8 ? k / 5 : 8);
    return kp < RMU_KS_CAP_DEEP - 8 ? kp : RMU_KS_CAP_DEEP - 8;
}

static u64* g_dbg = nullptr;         // RMU_SCAN_EXP=7: cycle / event counters of the scan kernels (diagnostics only)
static u64* dbg_buffer() {
#ifndef RMU_DEBUG_KERNELS
    return nullptr;          // the counter instantiations exist only in a --debug-kernels build
#endif
    static std::once_flag once;
    std::call_once(once, [] {
        if (rmu_env("RMU_SCAN_EXP") && atoi(rmu_env("RMU_SCAN_EXP")) == 7) (void)hipMalloc((void**)&g_dbg, 128);
    });
    return g_dbg;
}
static void dbg_dump(const char* what, int64_t rows, hipStream_t s) {
    u64 h[16];
    (void)hipStreamSynchronize(s);
    (void)hipMemcpy(h, g_dbg, 128, hipMemcpyDeviceToHost);
    fprintf(stderr, "[rmu dbg %s: %lld rows] slow_tiles=%llu compactions=%llu appends=%llu wave_tiles=%llu rounds=%llu clk_slow=%llu clk_bar=%llu clk_all=%llu clk_vmwait=%llu seg=%llu/%llu/%llu/%llu/%llu/%llu/%llu\n",
            what, (long long)rows, h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], h[8], h[9], h[10], h[11], h[12], h[13], h[14], h[15]);
    (void)hipMemset(g_dbg, 0, 128);
}

// row ranges of the threshold ladder over n rows (see the comment in screen_enqueue)
// first range of the ladder for batches of <= 128 queries (one query tile).  tools/ladder_sweep.py, 10M rows, same box, step ms at
// (ratio : first) 8:256 / 8:2048 / 8:16384 / 16:16384 / 32:16384 -- B = 1: 1.302 / 1.277 / 1.269 / 1.303 / 1.275; B = 32: 1.414 / 1.380 /
// 1.649 / 2.027 / 1.823; B = 128: 1.551 / 1.485 / 1.788 / 2.276 / 2.004 (profiles/r05_ladder_sweep.txt): one launch + merge less pays,
// a colder start does not.  Full batches keep 256 (1.25M rows x 1024: 1.245 ms at 256, 1.372 at 2048, 1.794 at 16384).
static constexpr int LADDER_FIRST_SMALL = 2048;
static std::vector<int64_t> ladder_bounds(int64_t n, int64_t nb, int opt_ratio = 0, int opt_first = 0) {
    static const int lvl_min_env = rmu_env("RMU_SCREEN_MINLVL") ? atoi(rmu_env("RMU_SCREEN_MINLVL")) : 0;
    static const int lvl_ratio_env0 = rmu_env("RMU_SCREEN_RATIO") ? atoi(rmu_env("RMU_SCREEN_RATIO")) : 0;   // <= 1: single launch
    // (per-index options RMU_OPT_LADDER_RATIO / _FIRST first, then the environment, then the defaults)
    const int lvl_ratio_env = opt_ratio ? opt_ratio : lvl_ratio_env0;
    const int lvl_min = opt_first ? opt_first : lvl_min_env ? lvl_min_env : (nb <= 128 ? LADDER_FIRST_SMALL : 256);
    // ratio 3 for full batches; small batches (one query tile, HBM-bound: 7.68 GB image per batch) have few appends to
    // save and pay for every launch gap and merge, so they climb faster
    // (round 4: from 4 096 rows up -- was 262 144.  A single COLD launch over 200k rows x 1024 queries takes 1.74 ms, twice what the ladder
    // needs for 1M rows: every score that beats a still-empty threshold is appended.  With the ladder: 200k 0.45 ms, 20k 0.27, 8k 0.29.)
    static const int64_t ladder_min = rmu_env("RMU_SCREEN_LADDER_MIN") ? atoll(rmu_env("RMU_SCREEN_LADDER_MIN")) : 4096;
    auto build = [&](int ratio) {
        std::vector<int64_t> b{n};
        if (ratio > 1 && n >= ladder_min) {
            int64_t c = n / ratio / 32 * 32;
            for (; c >= 65536 && b.size() < 24; c = c / ratio / 32 * 32) b.insert(b.begin(), c);
            // below 64k rows a range is a handful of tiles per workgroup and its appends cost next to nothing: ratio 8, down
            // to a first range so small (<= lvl_min rows) that its cold start -- every score is appended -- does not matter
            for (c = b.front() / 8 / 32 * 32; c >= lvl_min && b.size() < 24; c = c / 8 / 32 * 32) b.insert(b.begin(), c);
        }
        return b;
    };
    if (lvl_ratio_env) return build(lvl_ratio_env);
    if (nb <= 128) return build(8);
    // (round 6, second session) Full batches over a corpus below ~6M rows: a ladder level costs 40-60 us whatever its range (launch, query
    // fragments, emit, merge) while a seeded level's appends are cheap -- the ratio with the FEWEST levels wins there (ties: the smaller ratio).
    // tools/ladder_sweep.py, same box, step ms at ratio 3 / 4 / 8: 1M x 1024 0.970 / 1.004 / 0.949, x 512 0.632 / 0.643 / 0.604, x 256 0.434 / 0.445 / 0.402;
    // 1.25M x 1024 1.169 / 1.144 / 1.161, x 512 0.762 / 0.727 / 0.729, x 256 0.530 / 0.488 / 0.495; 2.5M x 1024 at 3 / 5: 1.97 / 1.92, x 256 0.699 / 0.670;
    // 4.5M x 1024 at 3 / 6: 3.30 / 3.25, x 256 1.092 / 1.022; but 10M x 1024 6.71 / 6.77 / 6.78 at 3 / 4 / 8 (ratio 6: 6.77): there the appends of a
    // wider level cost more than the level it saves, and the ratio stays 3.
    std::vector<int64_t> best = build(3);
    if (n < 6000000)
        for (int r : {4, 5, 6, 8}) {
            std::vector<int64_t> b = build(r);
            if (b.size() < best.size()) best = std::move(b);
        }
    return best;
}

// Enqueue the screening ladder for `nb` device queries (fp32, [nb, 384]): on return (stream order) t.ckeys holds the best
// K' approximate candidates per query, sorted.  No host synchronisation.  scan_ms_events: record per-launch events.
// Threshold ladder: the corpus is scanned in row ranges of geometrically growing size (256 rows, x8 up to 64k, then x3);
// after each range its candidates are merged with the running top-K' and the K'-th best seeds the shared per-query
// thresholds of the next launch.  A cold launch appends K' ln(rows/K') candidates per query and CHUNK, a seeded one only
// K' (ratio - 1) per query in total, and every append stalls a whole workgroup for ~1-3k cycles: this cut
// the filter overhead of the 10M x 1024 scan from 5.2 to ~1.5 ms.
static int screen_enqueue(rmu_index* idx, Tls& t, const float* qdev, int64_t nb, hipStream_t s, bool timed, int* n_launches,
                          ScanLaunch* last_geom, int kp = kScreenKp, u32* zero_word = nullptr /* one more word the query conversion zeroes (the re-run count) */) {
    const int dpad = idx->dpad;
    static const int share = rmu_env("RMU_NO_SHARED_THR") ? 0 : 1;
    const std::vector<int64_t> bounds = ladder_bounds(idx->n, nb, idx->ladder_ratio, idx->ladder_first);
    const int nl = (int)bounds.size();
    std::vector<ScanLaunch> lv((size_t)nl);
    int slots = nl - 1;
    for (int l = 0; l < nl; ++l) {
        ScanLaunch& S = lv[(size_t)l];
        S = ScanLaunch{};
        S.x = (const float*)idx->split; S.row0 = l ? bounds[(size_t)l - 1] : 0; S.n_rows = bounds[(size_t)l] - S.row0;
        S.dpad = 384; S.nq = (int)nb; S.k = kp;       // (the image's geometry; an L2 index keeps its fp32 rows 768 floats apart)
        S.nrm = idx->nrm;
        if (rmu_screen_plan(&S) != RMU_OK) return fail(RMU_E_INVALID, "rmu_index_search: screening geometry");
        slots += S.parts;
    }
    const size_t part_keys = (size_t)nb * kp;
    const size_t gwords = (size_t)((nb + 255) / 256 * 256 + 64);
    size_t pwords = 0;                                   // sibling-pacing progress words: [s_chunks][4] per paced launch
    for (int l = 0; l < nl; ++l)
        if (lv[(size_t)l].pace) pwords += (size_t)lv[(size_t)l].s_chunks * 4;
    const size_t gbytes = (gwords + pwords) * sizeof(u32);    // one memset zeroes thresholds and progress words
    size_t gcand_bytes = 0;                              // K-split launches: global candidate slots (reused by every launch of the ladder)
    for (int l = 0; l < nl; ++l)
        if (lv[(size_t)l].kv >= 1)
            gcand_bytes = std::max(gcand_bytes, (size_t)lv[(size_t)l].parts * (size_t)nb * (kp > RMU_KS_CAP - 8 ? RMU_KS_CAP_DEEP : RMU_KS_CAP) * sizeof(u64));
    if (gcand_bytes && t.gcand.ensure(gcand_bytes)) return fail(RMU_E_OOM, "rmu_index_search: screening candidate slots");
    if (t.partial.ensure((size_t)slots * part_keys * sizeof(u64)) || t.qsplit.ensure((size_t)nb * RMU_IMG_ROW_BYTES) ||
        t.gthr.ensure(gbytes) || t.ckeys.ensure(part_keys * sizeof(u64)) || t.ensure_events(2 * nl))
        return fail(RMU_E_OOM, "rmu_index_search: screening workspace");
    static const int nofilter = rmu_env("RMU_SCREEN_NOFILTER") != nullptr ? 2 : 0;
    const int sflags = share | nofilter;
    // (an L2 index holds its queries as (2q, 1): the image is fp16(64 q) all the same)
    // (round 6) the conversion's first workgroup also zeroes the ladder's thresholds + pacing words and the caller's word: was two memsets
    int rc = rmu_split_launch(qdev, t.qsplit.p, nb, s, dpad, idx->metric == RMU_METRIC_L2SQ ? 32.0f : 64.0f, (u32*)t.gthr.p,
                              (int)(gbytes / sizeof(u32)), zero_word, zero_word ? 1 : 0);
    if (rc) return fail(rc, "rmu_index_search: query conversion");
    u64* base = (u64*)t.partial.p;
    int cursor = 0;     // slot index: [merged keys of the ranges so far][this range's parts] ...
    u32* prog_next = (u32*)t.gthr.p + gwords;
    for (int l = 0; l < nl; ++l) {
        ScanLaunch& S = lv[(size_t)l];
        const int first = cursor;               // slot of the running top-K' (l > 0), else of this range's first part
        if (l > 0) cursor += 1;
        S.prog = nullptr;
        if (S.pace) { S.prog = prog_next; prog_next += (size_t)S.s_chunks * 4; }
        S.partial = base + (size_t)cursor * part_keys;
        // (round 6) the FIRST launch is cold -- one tile per chunk, every row a candidate -- and its slots leave unsorted (share_thr bit 2; its merge
        // below is told): sorting 32 keys for each of a workgroup's 256 queries was ~25 us of a launch that scans 2 000 rows.  RMU_EMIT_RAW=0: never.
        static const int emit_raw = (rmu_env("RMU_EMIT_RAW") && atoi(rmu_env("RMU_EMIT_RAW")) == 0) ? 0 : 4;
        const int raw = l == 0 ? emit_raw : 0;
        S.gthr = (u32*)t.gthr.p; S.share_thr = sflags | raw; S.dbg = g_dbg; S.q = (const float*)t.qsplit.p;
        S.gcand = S.kv >= 1 ? (u64*)t.gcand.p : nullptr;
        if (timed) HIP_TRY(hipEventRecord(t.lev[(size_t)(2 * l)], s));
        rc = rmu_screen_launch(&S, s);
        if (rc) return fail(rc, "rmu_index_search: screening launch");
        if (timed) HIP_TRY(hipEventRecord(t.lev[(size_t)(2 * l + 1)], s));
        cursor += S.parts;
        if (g_dbg) dbg_dump("screen range", S.n_rows, s);
        u64* merged = l + 1 < nl ? base + (size_t)cursor * part_keys : (u64*)t.ckeys.p;
        rc = rmu_merge_to_keys_launch(base + (size_t)first * part_keys, cursor - first, nb, kp, merged,
                                      l + 1 < nl ? (u32*)t.gthr.p : nullptr /* merge + seed in one launch */, s, raw ? 1 : 0);
        if (rc) return fail(rc, "rmu_index_search: screening merge / threshold seeding");
    }
    *n_launches = nl;
    if (last_geom) *last_geom = lv.back();
    return RMU_OK;
}

static bool screen_applies(const rmu_index* idx, int64_t nb, int k) {
    // (read once: the per-index switch is rmu_index_set_option(RMU_OPT_SCREEN_MIN_NQ))
    static const bool env_set = rmu_env("RMU_SCREEN_MIN_NQ") != nullptr;
    static const int env_min_nq = env_set ? atoi(rmu_env("RMU_SCREEN_MIN_NQ")) : 1;
    const bool min_nq_set = env_set || idx->screen_min_nq > 0;
    const int64_t screen_min_nq = idx->screen_min_nq > 0 ? idx->screen_min_nq : env_min_nq;
    // small batches are HBM-bound either way: the screen reads half the bytes (768 vs 1536 B per row) but pays for the
    // ladder's launches and merges per batch, which only pays off on a large enough corpus
    // (deep k: the alternative is the exact 128-deep ladder at ~0.3 of the HBM roof -- the screen pays from the size that ladder starts at)
    const bool screen_pays = nb >= 128 || idx->n >= 3000000 || (nb > 64 && idx->n >= 1000000) || min_nq_set || (k > 32 && idx->n >= 262144);
    const bool geom = idx->dim == 384 && (idx->metric == RMU_METRIC_L2SQ ? idx->nrm != nullptr : idx->dpad == 384);
    return idx->split && idx->screen_enabled && geom &&
           nb >= screen_min_nq && screen_pays && k <= kScreenMaxK && idx->n > 0 && idx->xnorm_max > 0.f &&
           idx->xnorm_max < 500.f;   // fp16(64*x) must not overflow
}

// deep k (the 128-deep candidate geometry) over a large corpus takes the exact scan as a threshold ladder (see rmu_index_search)
static bool deep_applies(const rmu_index* idx, int k) {
    static const bool off = rmu_env("RMU_DEEP") && atoi(rmu_env("RMU_DEEP")) == 0;
    return !off && k > 32 && idx->n >= 262144;
}
static std::vector<int64_t> deep_bounds(int64_t n) {
    static const int first = rmu_env("RMU_DEEP_FIRST") ? atoi(rmu_env("RMU_DEEP_FIRST")) : 8192;
    static const int ratio = rmu_env("RMU_DEEP_RATIO") ? atoi(rmu_env("RMU_DEEP_RATIO")) : 4;
    std::vector<int64_t> b{n};
    for (int64_t c = n / ratio / 128 * 128; c >= first && b.size() < 8; c = c / ratio / 128 * 128) b.insert(b.begin(), c);
    return b;
}

extern "C" int rmu_index_search(rmu_index_t* idx, const float* q, int64_t nq, int k, unsigned flags, int64_t row_base,
                                float* out_scores, int64_t* out_rows, uint64_t hip_stream) {
    RMU_ENTRY();
    if (!idx || !q || !out_scores || !out_rows) return fail(RMU_E_INVALID, "rmu_index_search: null pointer");
    if (nq < 1) return fail(RMU_E_INVALID, "rmu_index_search: nq must be >= 1");
    if (k < 1 || k > RMU_MAX_K) return fail(RMU_E_INVALID, "rmu_index_search: k must be in [1, 112]");
    Tls& t = g_tls;
    int rc = t.ensure_stream((hipStream_t)hip_stream);
    if (rc) return fail(rc, "rmu_index_search: stream");
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : t.stream;
    const bool q_dev = flags & RMU_F_Q_DEVICE, out_dev = flags & RMU_F_OUT_DEVICE;
    const bool timed = t.timing && !hip_stream;
    (void)dbg_buffer();

    std::shared_lock<std::shared_mutex> lk(idx->mu);
    const int dpad = idx->dpad, dim = idx->dim;
    const bool l2 = idx->metric == RMU_METRIC_L2SQ;     // dpad > dim by construction: queries are always copied and augmented
    t.scan_ms = -1.f; t.search_ms = -1.f; t.passes = 0; t.screened = 0;
    float scan_total = 0.f;
    bool any_screened = false;
    int rerun_total = 0;
    if (timed) HIP_TRY(hipEventRecord(t.ev[0], s));
    if (!t.hflag) {
        if (hipHostMalloc((void**)&t.hflag, sizeof(int)) != hipSuccess) { t.hflag = nullptr; return fail(RMU_E_OOM, "rmu_index_search: pinned flag"); }
        *t.hflag = 0;
        if (hipHostGetDevicePointer((void**)&t.hflag_dev, t.hflag, 0) != hipSuccess) { (void)hipGetLastError(); t.hflag_dev = t.hflag; }   // (unified addressing: the same pointer)
    }

    for (int64_t q0 = 0; q0 < nq; q0 += kMaxQueriesPerLaunch) {
        const int64_t nb = (nq - q0) < kMaxQueriesPerLaunch ? (nq - q0) : kMaxQueriesPerLaunch;
        // ---- queries -> device, padded to dpad, normalised for COSINE, augmented for L2SQ -----------------------------
        const float* qsrc = q + q0 * dim;
        const float* qdev = qsrc;
        const bool need_copy = !q_dev || dpad != dim || idx->metric == RMU_METRIC_COSINE;
        if (need_copy) {
            if (t.q.ensure((size_t)nb * dpad * sizeof(float))) return fail(RMU_E_OOM, "rmu_index_search: q workspace");
            if (dpad != dim) HIP_TRY(hipMemsetAsync(t.q.p, 0, (size_t)nb * dpad * sizeof(float), s));
            HIP_TRY(hipMemcpy2DAsync(t.q.p, (size_t)dpad * sizeof(float), qsrc, (size_t)dim * sizeof(float),
                                     (size_t)dim * sizeof(float), (size_t)nb,
                                     q_dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s));
            if (idx->metric == RMU_METRIC_COSINE) {
                hipLaunchKernelGGL(k_row_norm, dim3((unsigned)((nb + 3) / 4)), dim3(256), 0, s, (float*)t.q.p, dpad, nb, 1,
                                   (float*)nullptr);
                HIP_TRY(hipGetLastError());
            }
            if (l2) {
                if (t.qn.ensure((size_t)nb * sizeof(float))) return fail(RMU_E_OOM, "rmu_index_search: |q|^2 workspace");
                hipLaunchKernelGGL(k_l2_aug_queries, dim3((unsigned)((nb + 3) / 4)), dim3(256), 0, s, (float*)t.q.p, dpad, dim, nb,
                                   (float*)t.qn.p);
                HIP_TRY(hipGetLastError());
            }
            qdev = (const float*)t.q.p;
        }
        float* d_s = out_scores + q0 * k;
        int64_t* d_r = out_rows + q0 * k;
        if (!out_dev) {
            if (t.out_s.ensure((size_t)nb * k * sizeof(float)) || t.out_r.ensure((size_t)nb * k * sizeof(int64_t)))
                return fail(RMU_E_OOM, "rmu_index_search: output workspace");
            d_s = (float*)t.out_s.p;
            d_r = (int64_t*)t.out_r.p;
        }
        static const int share = rmu_env("RMU_NO_SHARED_THR") ? 0 : 1;
        bool exact_timed = false;
        // ---- the exact fp32 fused scan + merge of `nqq` device queries.  plan first (all workspace is sized before anything
        // is enqueued: a grow-only buffer must not be re-allocated under work already in flight), then run.  `cond`
        // (optional) makes both launches conditional on the device-side count of flagged queries; `scatter` redirects
        // result row i to batch row scatter[i]. ------------------------------------------------------------------------
        size_t need_partial = 0, need_gthr = 0;
        auto plan_exact = [&](const float* qd, int64_t nqq, const RmuCond* cond, ScanLaunch* L, int kk = 0) -> int {
            *L = ScanLaunch{};
            if (kk <= 0) kk = k;
            L->x = idx->x; L->n_rows = idx->n; L->dpad = dpad; L->q = qd; L->nq = (int)nqq; L->k = kk; L->dbg = g_dbg;
            if (cond) L->cond = *cond;
            const int rc2 = rmu_scan_plan(L);
            if (rc2) return fail(rc2, "rmu_index_search: no scan geometry for this (dim, k)");
            need_partial = std::max(need_partial, (size_t)L->parts * nqq * kk * sizeof(u64));
            // shared per-query thresholds: padded to whole 128-query tiles, zero = no bound yet
            need_gthr = std::max(need_gthr, (size_t)((nqq + 127) / 128 * 128 + 64) * sizeof(u32));
            return RMU_OK;
        };
        // zero_gthr_bytes: bytes of the shared thresholds to zero in front of the launch (0: an earlier launch of a mutually exclusive
        // group already did -- only ONE launch of such a group ever runs, and a launch that does not run touches nothing)
        auto run_exact = [&](ScanLaunch& L, float* os, int64_t* orr, const int64_t* scatter, bool time_it, size_t zero_gthr_bytes) -> int {
            const bool has_cond = L.cond.p != nullptr;
            const size_t pbytes = (size_t)L.parts * L.nq * L.k * sizeof(u64);
            L.partial = (u64*)t.partial.p;
            L.gthr = (u32*)t.gthr.p;
            L.share_thr = share;
            if (zero_gthr_bytes) HIP_TRY(hipMemsetAsync(t.gthr.p, 0, zero_gthr_bytes, s));
            int rc2;
            if (idx->n > 0) {
                if (time_it) HIP_TRY(hipEventRecord(t.ev[2], s));
                rc2 = rmu_scan_launch(&L, s);
                if (rc2) return fail(rc2, std::string("rmu_index_search: scan launch: ") + hipGetErrorString(hipGetLastError()));
                if (time_it) { HIP_TRY(hipEventRecord(t.ev[3], s)); exact_timed = true; }
                if (!has_cond) { t.grid = L.grid; t.block = 256; t.lds = L.lds_bytes; t.passes += 1; }
            } else {
                HIP_TRY(hipMemsetAsync(L.partial, 0, pbytes, s));
            }
            rc2 = rmu_merge_final_launch(L.partial, L.parts, L.nq, L.k, row_base, l2 ? 1 : 0, l2 ? (const float*)t.qn.p : nullptr, os, orr,
                                         scatter, has_cond ? &L.cond : nullptr, s);
            if (rc2) return fail(rc2, "rmu_index_search: merge launch");
            if (g_dbg && !has_cond) dbg_dump("exact", idx->n, s);
            return RMU_OK;
        };
        const bool screened = screen_applies(idx, nb, k);
        if (screened) {
            // ---- screened path: fp16 scan proposes K' = 32 candidates, exact fp32 re-score decides --------------------------
            // Flagged queries -> exact scan, decided ON THE DEVICE: three mutually exclusive conditional launches
            //   1..32 flagged: the 32-query HBM-bound geometry; 33..nb/8: a batch of nb/8; more: the whole batch.
            // With nothing flagged (the normal case) each costs one empty grid (~2 us); no host round trip either way, so
            // the whole search is asynchronous on the stream it was given.
            // (round 5) Every dependent launch of a search costs ~4.4 us of kernel boundary on this part, run or not, and a batch of <= 32
            // queries already IS the 32-query geometry: one class there -- "anything flagged: the exact scan of the whole batch" -- without
            // the gather (3 stream operations behind the re-score instead of 7); the classes of a larger batch share ONE threshold memset.
            const bool one_class = nb <= 32;
            const int small_n = (int)(nb < 32 ? nb : 32), mid_n = one_class ? 0 : (int)(nb / 8);
            const int gather_n = mid_n > small_n ? mid_n : small_n;
            const int lim_small = one_class ? 0 : (mid_n > small_n ? small_n : mid_n);    // no mid launch: the small one covers 1..nb/8
            if (t.flag.ensure(2 * sizeof(int)) || t.fb_i.ensure((size_t)nb * sizeof(int64_t)) ||
                t.fbq.ensure((size_t)gather_n * dpad * sizeof(float)))
                return fail(RMU_E_OOM, "rmu_index_search: re-run workspace");
            const int* cnt = (const int*)t.flag.p;
            const RmuCond c1{cnt, 1, lim_small, 1}, c2{cnt, small_n + 1, mid_n, 1}, c3{cnt, (lim_small > 0 || mid_n > small_n ? mid_n : 0) + 1, 0x7fffffff, 0};
            ScanLaunch L1{}, L2{}, L3{};
            if (lim_small > 0 && (rc = plan_exact((const float*)t.fbq.p, small_n, &c1, &L1))) return rc;
            if (mid_n > small_n && (rc = plan_exact((const float*)t.fbq.p, mid_n, &c2, &L2))) return rc;
            if ((rc = plan_exact(qdev, nb, &c3, &L3))) return rc;
            if (t.partial.ensure(need_partial) || t.gthr.ensure(need_gthr)) return fail(RMU_E_OOM, "rmu_index_search: re-run partials");
            int nl = 0;
            ScanLaunch lastg{};
            rc = screen_enqueue(idx, t, qdev, nb, s, timed, &nl, &lastg, screen_kp(k), (u32*)t.flag.p);    // (flag[0] = 0 by the query conversion)
            if (rc) return rc;
            // |s~ - s_fp32| <= EPS(q) from the measured image errors (derivation in scan_screen.hip); queries failing the
            // sufficiency test are appended to the list fb_i (count in flag[0])
            rc = rmu_rescore_launch((const u64*)t.ckeys.p, screen_kp(k), idx->x, qdev, nb, k, idx->xnorm_max, idx->dx_max, row_base, d_s, d_r,
                                    (int*)t.flag.p, (int64_t*)t.fb_i.p, nullptr, s, dpad, l2 ? (const float*)t.qn.p : nullptr);
            if (rc) return fail(rc, "rmu_index_search: re-score launch");
            t.grid = lastg.grid; t.block = 256; t.lds = lastg.lds_bytes; t.passes += nl;
            {
                // gather (batches above 32 queries) + the re-runs' threshold zeroing + the count to the host, one launch (see the kernel)
                const int g_n = (lim_small > 0 || mid_n > small_n) ? gather_n : 0;
                hipLaunchKernelGGL(k_gather_flagged, dim3((unsigned)(g_n > 0 ? g_n : 1)), dim3(128), 0, s, qdev, dpad, (const int64_t*)t.fb_i.p, cnt,
                                   (float*)t.fbq.p, g_n, (u32*)t.gthr.p, (int)(need_gthr / sizeof(u32)), t.hflag_dev);
                HIP_TRY(hipGetLastError());
            }
            size_t zero_once = 0;                  // (the thresholds of the mutually exclusive re-run launches were zeroed by the launch above)
            if (lim_small > 0) { if ((rc = run_exact(L1, d_s, d_r, (const int64_t*)t.fb_i.p, false, zero_once))) return rc; zero_once = 0; }
            if (mid_n > small_n) { if ((rc = run_exact(L2, d_s, d_r, (const int64_t*)t.fb_i.p, false, zero_once))) return rc; zero_once = 0; }
            if ((rc = run_exact(L3, d_s, d_r, nullptr, false, zero_once))) return rc;
            if (timed)
                for (int l = 0; l < nl; ++l) {
                    HIP_TRY(hipEventSynchronize(t.lev[(size_t)(2 * l + 1)]));
                    float ms = 0.f;
                    if (hipEventElapsedTime(&ms, t.lev[(size_t)(2 * l)], t.lev[(size_t)(2 * l + 1)]) == hipSuccess) scan_total += ms;
                }
        } else if (deep_applies(idx, k)) {
            // ---- deep k (32 < k <= 112) over a large corpus (BASELINE config 5's dense top-100): the 128-deep scan is slow COLD --
            // k ln(rows / k) appends per query and chunk, each stalling a barrier-coupled workgroup -- not per byte.  So the
            // threshold ladder of the screening path, with exact arithmetic: the corpus is scanned in row ranges of growing size
            // (~8k rows, then x4), after each range the candidates are merged with the running top-k (keys) and its k-th best
            // seeds the shared per-query thresholds of the next launch; the last merge writes the results.  Every launch is
            // the exact fp32 scan, every bound is a real k-th best: results are those of the single launch, bit for bit.
            const std::vector<int64_t> bounds = deep_bounds(idx->n);
            const int nl = (int)bounds.size();
            std::vector<ScanLaunch> lv((size_t)nl);
            int slots = nl - 1;
            size_t gthr_need = 0;
            for (int l = 0; l < nl; ++l) {
                ScanLaunch& S = lv[(size_t)l];
                S = ScanLaunch{};
                S.row0 = l ? bounds[(size_t)l - 1] : 0;
                S.x = idx->x + S.row0 * dpad; S.n_rows = bounds[(size_t)l] - S.row0; S.dpad = dpad; S.q = qdev; S.nq = (int)nb; S.k = k; S.dbg = g_dbg;
                if ((rc = rmu_scan_plan(&S))) return fail(rc, "rmu_index_search: no scan geometry for this (dim, k)");
                slots += S.parts;
                gthr_need = std::max(gthr_need, (size_t)((nb + 127) / 128 * 128 + 64) * sizeof(u32));
            }
            const size_t part_keys = (size_t)nb * k;
            if (t.partial.ensure((size_t)slots * part_keys * sizeof(u64)) || t.gthr.ensure(gthr_need))
                return fail(RMU_E_OOM, "rmu_index_search: ladder workspace");
            HIP_TRY(hipMemsetAsync(t.gthr.p, 0, gthr_need, s));
            u64* base = (u64*)t.partial.p;
            int cursor = 0;
            if (timed) HIP_TRY(hipEventRecord(t.ev[2], s));
            for (int l = 0; l < nl; ++l) {
                ScanLaunch& S = lv[(size_t)l];
                const int first = cursor;           // slot of the running top-k (l > 0), else of this range's first part
                if (l > 0) cursor += 1;
                S.partial = base + (size_t)cursor * part_keys; S.gthr = (u32*)t.gthr.p; S.share_thr = share;
                rc = rmu_scan_launch(&S, s);
                if (rc) return fail(rc, std::string("rmu_index_search: scan launch: ") + hipGetErrorString(hipGetLastError()));
                cursor += S.parts;
                if (l + 1 < nl)
                    rc = rmu_merge_to_keys_launch(base + (size_t)first * part_keys, cursor - first, nb, k, base + (size_t)cursor * part_keys,
                                                  (u32*)t.gthr.p, s);
                else
                    rc = rmu_merge_final_launch(base + (size_t)first * part_keys, cursor - first, nb, k, row_base, l2 ? 1 : 0,
                                                l2 ? (const float*)t.qn.p : nullptr, d_s, d_r, nullptr, nullptr, s);
                if (rc) return fail(rc, "rmu_index_search: ladder merge");
            }
            if (timed) { HIP_TRY(hipEventRecord(t.ev[3], s)); exact_timed = true; }
            t.grid = lv.back().grid; t.block = 256; t.lds = lv.back().lds_bytes; t.passes += nl;
        } else {
            ScanLaunch L{};
            if ((rc = plan_exact(qdev, nb, nullptr, &L))) return rc;
            if (t.partial.ensure(need_partial) || t.gthr.ensure(need_gthr)) return fail(RMU_E_OOM, "rmu_index_search: partial workspace");
            if ((rc = run_exact(L, d_s, d_r, nullptr, timed, need_gthr))) return rc;
        }
        if (!out_dev) {
            HIP_TRY(hipMemcpyAsync(out_scores + q0 * k, d_s, (size_t)nb * k * sizeof(float), hipMemcpyDeviceToHost, s));
            HIP_TRY(hipMemcpyAsync(out_rows + q0 * k, d_r, (size_t)nb * k * sizeof(int64_t), hipMemcpyDeviceToHost, s));
        }
        // workspace is reused by the next query block (and host outputs must land): drain per block.  With a caller stream,
        // device outputs and a single block nothing here waits: the work is merely ordered on that stream.
        const bool drained = !hip_stream || q0 + nb < nq || !out_dev;
        if (drained) HIP_TRY(hipStreamSynchronize(s));
        else mark_reader(idx, s);
        t.finished(s, drained);
        if (screened) {   // the count of re-run queries is unknown while a caller's stream still runs: reported as 0 then
            any_screened = true;
            if (drained) rerun_total += *t.hflag;
        }
        if (exact_timed) {
            HIP_TRY(hipEventSynchronize(t.ev[3]));
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, t.ev[2], t.ev[3]) == hipSuccess) scan_total += ms;
        }
    }
    t.screened = any_screened ? (rerun_total ? -rerun_total : 1) : 0;
    if (timed) {
        HIP_TRY(hipEventRecord(t.ev[1], s));
        HIP_TRY(hipEventSynchronize(t.ev[1]));
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, t.ev[0], t.ev[1]) == hipSuccess) t.search_ms = ms;
        t.scan_ms = scan_total;
    }
    return RMU_OK;
}

// Test hook (not a product entry point): the screening pass's candidate set for each query -- approximate scores s~,
// row ids, and the per-query error bound EPS(q) the sufficiency test uses -- so the bound |s~ - s_fp32| <= EPS can be
// checked ON THE HARDWARE against fp32 scores computed independently (tests/test_search_gpu.py).
extern "C" int rmu_index_screen_candidates(rmu_index_t* idx, const float* q_host, int64_t nq, float* out_approx, int64_t* out_rows,
                                           float* out_exact, float* out_eps) {
    RMU_ENTRY();
    if (!idx || !q_host || !out_approx || !out_rows || !out_exact || !out_eps) return fail(RMU_E_INVALID, "rmu_index_screen_candidates: null pointer");
    if (nq < 1 || nq > kMaxQueriesPerLaunch) return fail(RMU_E_INVALID, "rmu_index_screen_candidates: 1 <= nq <= 8192");
    Tls& t = g_tls;
    int rc = t.ensure_stream();
    if (rc) return fail(rc, "rmu_index_screen_candidates: stream");
    hipStream_t s = t.stream;
    std::shared_lock<std::shared_mutex> lk(idx->mu);
    if (!idx->split || idx->dim != 384 || idx->n <= 0) return fail(RMU_E_INVALID, "rmu_index_screen_candidates: index has no screening image");
    if (idx->metric == RMU_METRIC_L2SQ) return fail(RMU_E_INVALID, "rmu_index_screen_candidates: inner-product / cosine indexes only");
    const int kp = kScreenKp;
    if (t.q.ensure((size_t)nq * 384 * sizeof(float)) || t.flag.ensure((size_t)(nq + 1) * sizeof(int)) || t.fb_i.ensure((size_t)nq * sizeof(int64_t)) ||
        t.out_s.ensure((size_t)nq * kp * sizeof(float)) || t.out_r.ensure((size_t)nq * kp * sizeof(int64_t)) || t.nrm.ensure((size_t)nq * sizeof(float)))
        return fail(RMU_E_OOM, "rmu_index_screen_candidates: workspace");
    HIP_TRY(hipMemcpyAsync(t.q.p, q_host, (size_t)nq * 384 * sizeof(float), hipMemcpyHostToDevice, s));
    if (idx->metric == RMU_METRIC_COSINE) hipLaunchKernelGGL(k_row_norm, dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, s, (float*)t.q.p, 384, nq, 1, (float*)nullptr);
    HIP_TRY(hipMemsetAsync(t.flag.p, 0, sizeof(int), s));
    int nl = 0;
    rc = screen_enqueue(idx, t, (const float*)t.q.p, nq, s, false, &nl, nullptr);
    if (rc) return rc;
    // k = K': the re-score kernel writes the exact fp32 score of EVERY candidate (rank order) and EPS(q)
    rc = rmu_rescore_launch((const u64*)t.ckeys.p, kp, idx->x, (const float*)t.q.p, nq, kp, idx->xnorm_max, idx->dx_max, 0, (float*)t.out_s.p,
                            (int64_t*)t.out_r.p, (int*)t.flag.p, (int64_t*)t.fb_i.p, (float*)t.nrm.p, s);
    if (rc) return fail(rc, "rmu_index_screen_candidates: re-score");
    std::vector<u64> keys((size_t)nq * kp);
    std::vector<float> ex((size_t)nq * kp);
    std::vector<int64_t> er((size_t)nq * kp);
    HIP_TRY(hipMemcpyAsync(keys.data(), t.ckeys.p, keys.size() * sizeof(u64), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(ex.data(), t.out_s.p, ex.size() * sizeof(float), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(er.data(), t.out_r.p, er.size() * sizeof(int64_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(out_eps, t.nrm.p, (size_t)nq * sizeof(float), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    for (int64_t i = 0; i < nq; ++i)
        for (int c = 0; c < kp; ++c) {
            const u64 key = keys[(size_t)(i * kp + c)];
            out_approx[i * kp + c] = key ? rmu_key_score(key) : -INFINITY;
            out_rows[i * kp + c] = key ? (int64_t)rmu_key_row(key) : -1;
            // exact fp32 score of that same row: look it up in the re-scored (exact-rank-ordered) list
            float e = -INFINITY;
            for (int c2 = 0; c2 < kp && key; ++c2)
                if (er[(size_t)(i * kp + c2)] == (int64_t)rmu_key_row(key)) { e = ex[(size_t)(i * kp + c2)]; break; }
            out_exact[i * kp + c] = e;
        }
    return RMU_OK;
}

extern "C" int rmu_index_set_option(rmu_index_t* idx, int option, int64_t value) {
    if (!idx) return fail(RMU_E_INVALID, "rmu_index_set_option: null index");
    std::unique_lock<std::shared_mutex> lk(idx->mu);
    switch (option) {
        case RMU_OPT_SCREEN: idx->screen_enabled = value != 0; return RMU_OK;
        case RMU_OPT_SCREEN_MIN_NQ: idx->screen_min_nq = value > 0 ? value : 0; return RMU_OK;
        case RMU_OPT_LADDER_RATIO: idx->ladder_ratio = value > 0 && value <= 4096 ? (int)value : 0; return RMU_OK;
        case RMU_OPT_LADDER_FIRST: idx->ladder_first = value > 0 && value <= (1 << 30) ? (int)value : 0; return RMU_OK;
        default: return fail(RMU_E_INVALID, "rmu_index_set_option: unknown option");
    }
}
extern "C" int rmu_index_metric(rmu_index_t* idx, int* metric) {
    if (!idx || !metric) return fail(RMU_E_INVALID, "rmu_index_metric: null");
    *metric = idx->metric;
    return RMU_OK;
}

extern "C" int rmu_topk_merge(const float* scores, const int64_t* rows, int parts, int64_t nq, int k, unsigned flags,
                              float* out_scores, int64_t* out_rows, uint64_t hip_stream) {
    RMU_ENTRY();
    if (!scores || !rows || !out_scores || !out_rows) return fail(RMU_E_INVALID, "rmu_topk_merge: null pointer");
    if (parts < 1 || nq < 1 || k < 1 || k > 128) return fail(RMU_E_INVALID, "rmu_topk_merge: parts/nq >= 1, k in [1,128]");
    Tls& t = g_tls;
    int rc = t.ensure_stream((hipStream_t)hip_stream);
    if (rc) return fail(rc, "rmu_topk_merge: stream");
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : t.stream;
    const bool in_dev = flags & RMU_F_Q_DEVICE, out_dev = flags & RMU_F_OUT_DEVICE;
    const size_t cnt = (size_t)parts * nq * k;
    const float* ds = scores;
    const int64_t* dr = rows;
    if (!in_dev) {
        if (t.in_s.ensure(cnt * sizeof(float)) || t.in_r.ensure(cnt * sizeof(int64_t)))
            return fail(RMU_E_OOM, "rmu_topk_merge: input workspace");
        HIP_TRY(hipMemcpyAsync(t.in_s.p, scores, cnt * sizeof(float), hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(t.in_r.p, rows, cnt * sizeof(int64_t), hipMemcpyHostToDevice, s));
        ds = (const float*)t.in_s.p;
        dr = (const int64_t*)t.in_r.p;
    }
    float* os = out_scores;
    int64_t* orr = out_rows;
    if (!out_dev) {
        if (t.out_s.ensure((size_t)nq * k * sizeof(float)) || t.out_r.ensure((size_t)nq * k * sizeof(int64_t)))
            return fail(RMU_E_OOM, "rmu_topk_merge: output workspace");
        os = (float*)t.out_s.p;
        orr = (int64_t*)t.out_r.p;
    }
    rc = rmu_merge_lists_launch(ds, dr, parts, nq * k, nq * k, nq, k, (flags & RMU_F_SMALLER_BETTER) ? 1 : 0, os, orr, s);
    if (rc) return fail(rc, "rmu_topk_merge: launch");
    if (!out_dev) {
        HIP_TRY(hipMemcpyAsync(out_scores, os, (size_t)nq * k * sizeof(float), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(out_rows, orr, (size_t)nq * k * sizeof(int64_t), hipMemcpyDeviceToHost, s));
    }
    const bool drained = !hip_stream || !out_dev || !in_dev;
    if (drained) HIP_TRY(hipStreamSynchronize(s));
    t.finished(s, drained);
    return RMU_OK;
}

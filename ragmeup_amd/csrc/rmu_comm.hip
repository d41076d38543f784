// rmu_comm.hip -- the ONE exchange step of the row-sharded search, behind the C-ABI.
//
// Rank r owns corpus rows [r*N/W, (r+1)*N/W) and has run rmu_index_search with row_base = its offset; every rank then
// holds [nq, k] (score, global row) lists.  rmu_shard_allgather_topk packs them into one struct-of-arrays byte buffer,
// issues a single RCCL all-gather over xGMI (nq*k*12 B per rank: 120 KB at nq = 1024, k = 10 -- latency-bound, nowhere
// near the 7 x 153 GB/s links) and merges the W lists on the device.  No reference counterpart: the reference is
// single-process (the local step serves server/RAGHelper.py:497-499).
//
// RCCL is bound at run time (dlopen) so that librmu.so carries no link-time dependency on it and shares the copy a host
// framework (PyTorch bundles its own librccl.so) has already loaded, instead of bringing a second one into the process.
#include <dlfcn.h>
#include <cstring>
#include <mutex>
#include <new>
#include <string>

#include <rccl/rccl.h>

#include "rmu_common.h"
#include "../../include/rmu.h"

extern "C" void rmu_set_error_(const char* msg);
static int cfail(int code, const std::string& m) { rmu_set_error_(m.c_str()); return code; }

namespace {

struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;          // optional: rmu_comm_world asks the communicator itself
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
    std::string error;
};
RcclApi g_rccl;
std::once_flag g_rccl_once;

const RcclApi& rccl() {
    std::call_once(g_rccl_once, [] {
        const char* names[] = {"librccl.so", "librccl.so.1"};
        for (const char* n : names)                       // already in the process (e.g. loaded by torch)?
            if (!g_rccl.handle) g_rccl.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
        const char* paths[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char* n : paths)
            if (!g_rccl.handle) g_rccl.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (!g_rccl.handle) { g_rccl.error = std::string("librccl.so not found: ") + (dlerror() ? dlerror() : ""); return; }
        g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))dlsym(g_rccl.handle, "ncclGetUniqueId");
        g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))dlsym(g_rccl.handle, "ncclCommInitRank");
        g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))dlsym(g_rccl.handle, "ncclCommDestroy");
        g_rccl.AllGather = (decltype(g_rccl.AllGather))dlsym(g_rccl.handle, "ncclAllGather");
        g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))dlsym(g_rccl.handle, "ncclGetErrorString");
        g_rccl.CommCount = (decltype(g_rccl.CommCount))dlsym(g_rccl.handle, "ncclCommCount");
        g_rccl.CommUserRank = (decltype(g_rccl.CommUserRank))dlsym(g_rccl.handle, "ncclCommUserRank");
        if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.CommDestroy || !g_rccl.AllGather || !g_rccl.GetErrorString)
            g_rccl.error = "librccl.so lacks a required symbol";
    });
    return g_rccl;
}

// local [nq, k] lists -> one send buffer [scores (nq*k fp32, padded to 8 B) | rows (nq*k int64)]
__global__ void k_pack_lists(const float* __restrict__ s, const int64_t* __restrict__ r, int64_t n, int64_t rows_off, char* __restrict__ dst) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    ((float*)dst)[i] = s[i];
    ((int64_t*)(dst + rows_off))[i] = r[i];
}

}  // namespace

struct rmu_comm {
    ncclComm_t comm = nullptr;
    int world = 0, rank = 0, device = 0;
    hipStream_t stream = nullptr;
    std::mutex mu;                   // one collective at a time per communicator
    void *send = nullptr, *recv = nullptr;
    float* in_s = nullptr; int64_t* in_r = nullptr; float* out_s = nullptr; int64_t* out_r = nullptr;
    size_t cap_send = 0, cap_io = 0;
};

static_assert(RMU_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "unique-id size");

extern "C" int rmu_comm_unique_id(void* id_out) {
    RMU_ENTRY();
    if (!id_out) return cfail(RMU_E_INVALID, "rmu_comm_unique_id: null");
    const RcclApi& api = rccl();
    if (!api.error.empty()) return cfail(RMU_E_RCCL, api.error);
    ncclUniqueId id;
    const ncclResult_t r = api.GetUniqueId(&id);
    if (r != ncclSuccess) return cfail(RMU_E_RCCL, std::string("ncclGetUniqueId: ") + api.GetErrorString(r));
    memcpy(id_out, id.internal, NCCL_UNIQUE_ID_BYTES);
    return RMU_OK;
}

// Every entry point runs on the communicator's device, whichever device is current for the calling thread (a Flask or pool
// worker thread starts on device 0: on ranks > 0 the pack / merge kernels and the buffers would land on the wrong GPU).
struct CommDeviceScope {
    int prev = -1;
    bool switched = false;
    explicit CommDeviceScope(int dev) {
        if (hipGetDevice(&prev) == hipSuccess && prev != dev) switched = hipSetDevice(dev) == hipSuccess;
    }
    ~CommDeviceScope() { if (switched) (void)hipSetDevice(prev); }
};

extern "C" int rmu_comm_free(rmu_comm_t* c) {
    RMU_ENTRY();
    if (!c) return RMU_OK;
    CommDeviceScope dev(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm) (void)rccl().CommDestroy(c->comm);
    for (void* p : {c->send, c->recv, (void*)c->in_s, (void*)c->in_r, (void*)c->out_s, (void*)c->out_r})
        if (p) (void)hipFree(p);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return RMU_OK;
}

extern "C" int rmu_comm_init(rmu_comm_t** out, const void* id, int world, int rank) {
    RMU_ENTRY();
    if (!out || !id) return cfail(RMU_E_INVALID, "rmu_comm_init: null argument");
    if (world < 1 || rank < 0 || rank >= world) return cfail(RMU_E_INVALID, "rmu_comm_init: need 0 <= rank < world");
    const RcclApi& api = rccl();
    if (!api.error.empty()) return cfail(RMU_E_RCCL, api.error);
    auto* c = new (std::nothrow) rmu_comm();
    if (!c) return cfail(RMU_E_OOM, "rmu_comm_init: host alloc");
    c->world = world; c->rank = rank;
    if (hipGetDevice(&c->device) != hipSuccess || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        c->comm = nullptr;
        rmu_comm_free(c);                            // one cleanup path (a created stream goes back too)
        return cfail(RMU_E_HIP, "rmu_comm_init: stream");
    }
    ncclUniqueId uid;
    memcpy(uid.internal, id, NCCL_UNIQUE_ID_BYTES);
    const ncclResult_t r = api.CommInitRank(&c->comm, world, uid, rank);
    if (r != ncclSuccess) {
        c->comm = nullptr;
        rmu_comm_free(c);
        return cfail(RMU_E_RCCL, std::string("ncclCommInitRank: ") + api.GetErrorString(r));
    }
    *out = c;
    return RMU_OK;
}

extern "C" int rmu_comm_world(rmu_comm_t* c, int* world, int* rank) {
    RMU_ENTRY();
    if (!c) return cfail(RMU_E_INVALID, "rmu_comm_world: null");
    // what RCCL says the communicator is (ncclCommCount / ncclCommUserRank), not what rmu_comm_init was asked for
    int w = c->world, r = c->rank;
    const RcclApi& api = rccl();
    if (c->comm && api.CommCount && api.CommUserRank) {
        ncclResult_t e = api.CommCount(c->comm, &w);
        if (e == ncclSuccess) e = api.CommUserRank(c->comm, &r);
        if (e != ncclSuccess) return cfail(RMU_E_RCCL, std::string("ncclCommCount/ncclCommUserRank: ") + api.GetErrorString(e));
    }
    if (world) *world = w;
    if (rank) *rank = r;
    return RMU_OK;
}

extern "C" int rmu_shard_allgather_topk(rmu_comm_t* c, const float* scores, const int64_t* rows, int64_t nq, int k, unsigned flags,
                                        float* out_scores, int64_t* out_rows, uint64_t hip_stream) {
    RMU_ENTRY();
    if (!c || !scores || !rows || !out_scores || !out_rows) return cfail(RMU_E_INVALID, "rmu_shard_allgather_topk: null pointer");
    if (nq < 1 || k < 1 || k > 128) return cfail(RMU_E_INVALID, "rmu_shard_allgather_topk: nq >= 1, k in [1,128]");
    const RcclApi& api = rccl();
    std::lock_guard<std::mutex> lk(c->mu);
    CommDeviceScope dev(c->device);
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : c->stream;
    const bool in_dev = flags & RMU_F_Q_DEVICE, out_dev = flags & RMU_F_OUT_DEVICE;
    const int64_t n = nq * k;
    const size_t rows_off = ((size_t)n * sizeof(float) + 7) & ~(size_t)7;
    const size_t per_rank = rows_off + (size_t)n * sizeof(int64_t);
#define C_TRY(expr)                                                                                                  \
    do {                                                                                                             \
        hipError_t e_ = (expr);                                                                                      \
        if (e_ != hipSuccess) return cfail(e_ == hipErrorOutOfMemory ? RMU_E_OOM : RMU_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)
    if (per_rank > c->cap_send) {
        C_TRY(hipStreamSynchronize(s));
        if (c->send) (void)hipFree(c->send);
        if (c->recv) (void)hipFree(c->recv);
        c->send = c->recv = nullptr; c->cap_send = 0;
        C_TRY(hipMalloc(&c->send, per_rank));
        C_TRY(hipMalloc(&c->recv, per_rank * (size_t)c->world));
        c->cap_send = per_rank;
    }
    if ((!in_dev || !out_dev) && (size_t)n > c->cap_io) {
        C_TRY(hipStreamSynchronize(s));
        for (void* p : {(void*)c->in_s, (void*)c->in_r, (void*)c->out_s, (void*)c->out_r})
            if (p) (void)hipFree(p);
        c->in_s = c->out_s = nullptr; c->in_r = c->out_r = nullptr; c->cap_io = 0;
        C_TRY(hipMalloc((void**)&c->in_s, (size_t)n * sizeof(float)));
        C_TRY(hipMalloc((void**)&c->in_r, (size_t)n * sizeof(int64_t)));
        C_TRY(hipMalloc((void**)&c->out_s, (size_t)n * sizeof(float)));
        C_TRY(hipMalloc((void**)&c->out_r, (size_t)n * sizeof(int64_t)));
        c->cap_io = (size_t)n;
    }
    const float* ds = scores;
    const int64_t* dr = rows;
    if (!in_dev) {
        C_TRY(hipMemcpyAsync(c->in_s, scores, (size_t)n * sizeof(float), hipMemcpyHostToDevice, s));
        C_TRY(hipMemcpyAsync(c->in_r, rows, (size_t)n * sizeof(int64_t), hipMemcpyHostToDevice, s));
        ds = c->in_s; dr = c->in_r;
    }
    hipLaunchKernelGGL(k_pack_lists, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, ds, dr, n, (int64_t)rows_off, (char*)c->send);
    C_TRY(hipGetLastError());
    // the single exchange of the path: every rank receives every rank's [scores | rows] block
    const ncclResult_t r = api.AllGather(c->send, c->recv, per_rank, ncclChar, c->comm, s);
    if (r != ncclSuccess) return cfail(RMU_E_RCCL, std::string("ncclAllGather: ") + api.GetErrorString(r));
    float* os = out_dev ? out_scores : c->out_s;
    int64_t* orr = out_dev ? out_rows : c->out_r;
    // ranks hold ascending row ranges, so "lower part first on ties" is the (score, row) order of the single-GPU search
    const int rc = rmu_merge_lists_launch((const float*)c->recv, (const int64_t*)((const char*)c->recv + rows_off), c->world,
                                          (int64_t)(per_rank / sizeof(float)), (int64_t)(per_rank / sizeof(int64_t)), nq, k,
                                          (flags & RMU_F_SMALLER_BETTER) ? 1 : 0, os, orr, s);
    if (rc) return cfail(rc, "rmu_shard_allgather_topk: merge launch");
    if (!out_dev) {
        C_TRY(hipMemcpyAsync(out_scores, os, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, s));
        C_TRY(hipMemcpyAsync(out_rows, orr, (size_t)n * sizeof(int64_t), hipMemcpyDeviceToHost, s));
    }
    if (!hip_stream || !out_dev || !in_dev) C_TRY(hipStreamSynchronize(s));
#undef C_TRY
    return RMU_OK;
}

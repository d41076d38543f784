// Micro-benchmark: how fast can all 256 CUs of an MI355X fill LDS (global_load_lds, 16 B per lane) or registers
// (global_load_dwordx4) from an L2-resident / MALL-resident / HBM-resident region?  The answer bounds every LDS-staged
// GEMM of bert.hip (flop per staged byte x this rate).  Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/fill tools/ubench/fill.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
extern __shared__ __attribute__((aligned(16))) char gsm[];
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// MODE 0: LDS-DMA; MODE 1: plain 16-B loads to registers.  ROWB: bytes per "row" a group of lanes covers contiguously
// (64 = four lanes per 64-B row piece with a row stride of `stride` bytes, like a K-slab of a row-major matrix; 1024 = fully
// contiguous 1 KiB per wave instruction).
template <int MODE, int DEPTH>
__global__ __launch_bounds__(256) void k_fill(const char* __restrict__ src, size_t footprint, int iters, int rowb, int stride, int lds_kb, float* sink) {
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    f32x4 acc = {0, 0, 0, 0};
    f32x4 vv[7];
    uint32_t lane_off;
    if (rowb >= 1024) lane_off = lane * 16;
    else lane_off = (lane >> 2) * stride + (lane & 3) * 16;                     // 16 rows x 64 B per instruction
    const uint32_t inst_bytes = rowb >= 1024 ? 1024 : 16 * stride;              // address advance per instruction
    const uint32_t mask = (uint32_t)(footprint - 1);                            // footprints are powers of two <= 4 GiB
    uint32_t pos = (uint32_t)__builtin_amdgcn_readfirstlane((int)((((size_t)blockIdx.x * 4 + w) * inst_bytes * 7) & mask));
    const int wrap = lds_kb >= 112 ? 3 : lds_kb >= 56 ? 1 : 0;                  // ring of 28-KiB blocks that fit the allocation
    for (int it = 0; it < iters; ++it) {
        char* blk = gsm + (it & wrap) * 28672 + w * 1024;
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            pos = (pos + inst_bytes) & mask;
            if (pos + 65536 > mask) pos = 0;                                    // keep lane offsets inside the footprint
            const char* p = src + pos;                                          // scalar 64-bit base
            if (MODE == 0) {
                uint32_t o = lane_off;
                asm volatile("" : "+v"(o));
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + o),
                                                 (__attribute__((address_space(3))) void*)(blk + i * 4096), 16, 0, 0);
            } else {
                asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(vv[i]) : "v"(lane_off), "s"(p) : "memory");
            }
        }
        if (MODE == 1) {   // registers of loads in flight must stay allocated until they land: wait, then consume all seven
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(vv[0]), "+v"(vv[1]), "+v"(vv[2]), "+v"(vv[3]), "+v"(vv[4]), "+v"(vv[5]), "+v"(vv[6]));
#pragma unroll
            for (int i = 0; i < 7; ++i) acc += vv[i];
        }
        if (MODE == 0) wait_vm<DEPTH>();
    }
    wait_vm<0>();
    if (MODE == 0) { __syncthreads(); acc[0] = *(float*)(gsm + threadIdx.x * 4); }
    if (acc[0] == 12345.678f) sink[0] = acc[0] + acc[1];
}

template <int MODE, int DEPTH>
static void run(const char* name, const char* src, size_t fp, int lds, int grid, int iters, int rowb, int stride, float* sink) {
    hipFuncSetAttribute((const void*)k_fill<MODE, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_fill<MODE, DEPTH>), dim3(grid), dim3(256), lds, 0, src, fp, iters / 4, rowb, stride, lds >> 10, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_fill<MODE, DEPTH>), dim3(grid), dim3(256), lds, 0, src, fp, iters, rowb, stride, lds >> 10, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)grid * 4 * 7 * 1024 * iters;
    printf("%-10s depth<=%2d+7 lds=%3dKB grid=%5d footprint=%8.1f MB rowb=%4d : %7.2f TB/s  (%.1f GB/s per CU)\n", name, DEPTH, lds >> 10, grid,
           fp / 1048576.0, rowb, bytes / ms / 1e9, bytes / ms / 1e6 / 256);
    fflush(stdout);
    hipError_t e = hipGetLastError(); if (e != hipSuccess) { printf("ERR %s\n", hipGetErrorString(e)); fflush(stdout); }
}

int main() {
    const size_t big = (size_t)4 << 30;
    char* src; hipMalloc((void**)&src, big); hipMemset(src, 1, big);
    float* sink; hipMalloc((void**)&sink, 64);
    const int it = 400;
    for (size_t fp : {(size_t)1 << 20, (size_t)32 << 20, (size_t)4 << 30}) {
        for (int rowb : {1024, 64}) {
            const int stride = 768;
            run<0, 21>("lds-dma", src, fp, 140 << 10, 256 * 8, it, rowb, stride, sink);
            run<0, 21>("lds-dma", src, fp, 70 << 10, 256 * 8, it, rowb, stride, sink);
            run<0, 21>("lds-dma", src, fp, 35 << 10, 256 * 8, it, rowb, stride, sink);
            run<1, 0>("vgpr-load", src, fp, 140 << 10, 256 * 8, it, rowb, stride, sink);
            run<1, 0>("vgpr-load", src, fp, 70 << 10, 256 * 8, it, rowb, stride, sink);
            run<1, 0>("vgpr-load", src, fp, 35 << 10, 256 * 8, it, rowb, stride, sink);
        }
    }
    return 0;
}

// Micro-benchmark: how fast can one CU / the chip move L2-resident data into LDS?
//   mode 0: global_load_lds_dwordx4 (LDS-DMA)      mode 1: global_load_dwordx4 -> VGPR -> ds_write_b128
// Each workgroup (256 threads) sweeps its own `span` bytes `iters` times; span small => L2/MALL hits.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_fill.hip -o /tmp/ubench_fill ; run: /tmp/ubench_fill
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
extern __shared__ __attribute__((aligned(16))) char lds[];
template <int MODE>
__global__ __launch_bounds__(256) void k(const char* __restrict__ src, size_t span, int iters, float* sink) {
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const char* base = src + (size_t)blockIdx.x * span;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        for (size_t off = 0; off < span; off += 64 * 1024) {          // 64 KiB per inner step: 16 x 1 KiB per wave
#pragma unroll
            for (int n = 0; n < 16; ++n) {
                const char* p = base + off + (size_t)((n * 4 + w) * 1024 + lane * 16);
                if (MODE == 0) {
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                        (__attribute__((address_space(3))) void*)(lds + (n * 4 + w) * 1024), 16, 0, 0);
                } else {
                    const float4 v = *(const float4*)p;
                    *(float4*)(lds + (n * 4 + w) * 1024 + lane * 16) = v;
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            acc += *(const float*)(lds + threadIdx.x * 4);
            __syncthreads();
        }
    }
    if (acc == 123.456f) sink[0] = acc;
}
int main() {
    const size_t span = 1 << 20;   // 1 MiB per workgroup
    for (int grid : {1, 32, 256, 512}) {
        char* d; float* sink;
        hipMalloc(&d, span * grid); hipMemset(d, 1, span * grid); hipMalloc(&sink, 4);
        for (int mode = 0; mode < 2; ++mode) {
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            const int iters = 200;
            auto launch = [&]() {
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 65536, 0, d, span, iters, sink);
                else hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 65536, 0, d, span, iters, sink);
            };
            hipFuncSetAttribute((const void*)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
            hipFuncSetAttribute((const void*)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
            launch(); hipDeviceSynchronize();
            hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            const double bytes = (double)span * iters * grid;
            printf("grid %4d mode %d (%s): %.1f GB/s total, %.1f GB/s per WG\n", grid, mode, mode ? "vgpr+ds_write" : "lds-dma",
                   bytes / ms / 1e6, bytes / ms / 1e6 / grid);
        }
        hipFree(d); hipFree(sink);
    }
    return 0;
}

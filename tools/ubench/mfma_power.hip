// mfma_power.hip -- what does the WHOLE chip sustain on v_mfma_f32_32x32x16_{f16,bf16} when the operands are real data?
// mfma_issue.hip measured 2.46-2.50 PFLOP/s for back-to-back MFMAs, but with the SAME operand registers in every instruction (and
// mostly zeros): nothing toggles in the multiplier arrays.  The screening kernel (f16, random unit vectors x 64) runs at MFMA busy
// 0.72 x 1.42 GHz, the bare MFMA stream of k_ffn3 at ~1.08 "GHz of matrix pipe": power.  This program runs nothing BUT MFMAs -- two
// waves per SIMD on every CU, 8 accumulators per wave, operands rotating over 8 x 8 different fragments held in registers -- for
// about a second per variant and prints the rate of every launch, so the power manager's steady state is what is read.
// Variants: zeros | constant operands | random f16 (N(0, 3.3), the image's distribution) | random bf16 | random f16 + one
// ds_read_b128 per MFMA (the fragment stream of the real kernels, without anything else).
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/mfma_power tools/ubench/mfma_power.hip && tools/ubench/mfma_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <cstring>
#include <algorithm>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

extern __shared__ __attribute__((aligned(16))) char lds_[];

// MODE 0: f16, 1: bf16, 2: f16 + ds_read_b128 per MFMA (the A operand comes from LDS, 8 fragments per wave, rotating)
template <int MODE>
__global__ __launch_bounds__(512) void k(const u32x4* __restrict__ src, float* __restrict__ out, int iters) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    u32x4 a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        a[i] = src[(size_t)((blockIdx.x * 8 + w) * 16 + i) * 64 + lane];
        b[i] = src[(size_t)((blockIdx.x * 8 + w) * 16 + 8 + i) * 64 + lane];
    }
    if (MODE == 2) {
#pragma unroll
        for (int i = 0; i < 8; ++i) *(u32x4*)(lds_ + (size_t)((w * 8 + i) * 64 + lane) * 16) = a[i];
        __syncthreads();
    }
    f32x16 acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            u32x4 av = a[i];
            if (MODE == 2) av = *(const u32x4*)(lds_ + (size_t)((w * 8 + i) * 64 + lane) * 16);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (MODE == 1)
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, b[(i + j) & 7]), acc[j], 0, 0, 0);
                else
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, av), __builtin_bit_cast(f16x8, b[(i + j) & 7]), acc[j], 0, 0, 0);
            }
            if (MODE == 2) asm volatile("" ::: "memory");     // (keep the LDS read inside the loop)
        }
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[j][e];
    out[(size_t)blockIdx.x * 512 + threadIdx.x] = s;
}

// MODE 3..6: the A operand of EVERY RD-th MFMA is a fresh 1-KiB fragment out of LDS (64 KiB of fragments shared by the eight waves, as
// the waves of the screening kernel share a row tile), read four fragments ahead with counted waits; FILL: each wave also issues one
// 1-KiB LDS-DMA piece (global_load_lds from a 64-MiB region, i.e. mostly L2 / MALL hits) per 8 MFMAs into a region nobody reads --
// the screening kernel's fill rate (24 pieces per 32-row tile = 3 per wave and 24 MFMAs).
template <int RD, bool FILL>
__global__ __launch_bounds__(512) void kl(const u32x4* __restrict__ src, float* __restrict__ out, int iters, const char* __restrict__ stream, unsigned stream_mask) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    u32x4 b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) b[i] = src[(size_t)((blockIdx.x * 8 + w) * 16 + 8 + i) * 64 + lane];
    // 64 fragments: wave w writes fragments 8 w .. 8 w + 7
#pragma unroll
    for (int i = 0; i < 8; ++i) *(u32x4*)(lds_ + (size_t)((w * 8 + i) * 64 + lane) * 16) = src[(size_t)((blockIdx.x * 8 + w) * 16 + i) * 64 + lane];
    __syncthreads();
    const unsigned lbase = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)lds_ + (unsigned)lane * 16u;
    char* fill_dst = lds_ + 64 * 1024 + w * 4096;
    unsigned fo = ((unsigned)blockIdx.x * 8u + (unsigned)w) * 65536u + (unsigned)lane * 16u;
    f32x16 acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    constexpr int NR = 64 / RD;                // reads per iteration
    u32x4 f[4];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f[r]) : "v"(lbase), "n"(((r * 17) & 63) * 1024));
#pragma unroll
        for (int n = 0; n < 64; ++n) {
            constexpr int dummy = 0; (void)dummy;
            const int r = n / RD;
            if (n % RD == 0) {
                // fragment r has landed once at most min(3, NR - 1 - r) younger reads are in flight
                if (NR - 1 - r >= 3) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(f[r & 3]));
                else if (NR - 1 - r == 2) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(f[r & 3]));
                else if (NR - 1 - r == 1) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(f[r & 3]));
                else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[r & 3]));
            }
            acc[n & 7] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, f[r & 3]), __builtin_bit_cast(f16x8, b[(n + (n >> 3)) & 7]), acc[n & 7], 0, 0, 0);
            if (n % RD == RD - 1 && r + 4 < NR)
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f[r & 3]) : "v"(lbase), "n"((((r + 4) * 17) & 63) * 1024));
            if (FILL && (n & 7) == 7) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(stream + (fo & stream_mask)),
                                                 (__attribute__((address_space(3))) void*)(fill_dst + ((n >> 3) & 3) * 1024), 16, 0, 0);
                fo += 1024u;
            }
        }
        if (FILL) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[j][e];
    out[(size_t)blockIdx.x * 512 + threadIdx.x] = s;
}

static unsigned short f2h(float f) { _Float16 h = (_Float16)f; unsigned short u; memcpy(&u, &h, 2); return u; }
static unsigned short f2b(float f) { unsigned int u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (unsigned short)(u >> 16); }
static float gauss() {
    const float u1 = (rand() + 1.0f) / (RAND_MAX + 2.0f), u2 = rand() / (RAND_MAX + 1.0f);
    return sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2);
}

int main(int argc, char** argv) {
    const double secs = argc > 1 ? atof(argv[1]) : 1.0;
    int ncu = 256;
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0); ncu = p.multiProcessorCount;
    const int grid = ncu;                      // one 8-wave workgroup per CU: two waves per SIMD
    const size_t nfrag = (size_t)grid * 8 * 16, nhalf = nfrag * 64 * 8;
    std::vector<unsigned short> h(nhalf);
    u32x4* d; float* o;
    hipMalloc(&d, nhalf * 2); hipMalloc(&o, (size_t)grid * 512 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;                    // 64 MFMAs per iteration and wave: 256k MFMAs per wave and launch (~ 4-8 ms)
    const double flop = (double)grid * 8 * iters * 64 * 32768.0;
    struct V { const char* name; int mode; int data; } vs[] = {
        {"f16  zeros", 0, 0}, {"f16  constant 1.5", 0, 1}, {"f16  random N(0,3.3)", 0, 2}, {"bf16 random N(0,3.3)", 1, 3},
        {"f16  random N(0,0.05) (unscaled unit-vector elements)", 0, 4}, {"f16  random + ds_read_b128 per 8 MFMAs", 2, 2},
        {"f16  random, A fragment from LDS: 1 read per 4 MFMAs", 13, 2}, {"f16  random, A fragment from LDS: 1 read per 2 MFMAs", 12, 2},
        {"f16  random, A fragment from LDS: 1 read per MFMA", 11, 2}, {"f16  random, 1 read per MFMA + LDS-DMA fill (1 KiB / 8 MFMAs)", 21, 2},
        {"f16  random, 1 read per 4 MFMAs + LDS-DMA fill", 23, 2}};
    char* stream; hipMalloc(&stream, 64u << 20); hipMemset(stream, 1, 64u << 20);
    const unsigned smask = (64u << 20) - 1024u;
    for (const V& v : vs) {
        for (size_t i = 0; i < nhalf; ++i) {
            const float g = v.data == 0 ? 0.f : v.data == 1 ? 1.5f : v.data == 4 ? 0.05f * gauss() : 3.3f * gauss();
            h[i] = v.mode == 1 ? f2b(g) : f2h(g);
        }
        hipMemcpy(d, h.data(), nhalf * 2, hipMemcpyHostToDevice);
        double elapsed = 0;
        std::vector<float> rates;
        while (elapsed < secs * 1e3) {
            hipEventRecord(e0, 0);
            if (v.mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(512), 0, 0, d, o, iters);
            else if (v.mode == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(512), 0, 0, d, o, iters);
            else if (v.mode == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(512), 64 * 1024, 0, d, o, iters);
            else if (v.mode == 13) hipLaunchKernelGGL((kl<4, false>), dim3(grid), dim3(512), 96 * 1024, 0, d, o, iters, stream, smask);
            else if (v.mode == 12) hipLaunchKernelGGL((kl<2, false>), dim3(grid), dim3(512), 96 * 1024, 0, d, o, iters, stream, smask);
            else if (v.mode == 11) hipLaunchKernelGGL((kl<1, false>), dim3(grid), dim3(512), 96 * 1024, 0, d, o, iters, stream, smask);
            else if (v.mode == 21) hipLaunchKernelGGL((kl<1, true>), dim3(grid), dim3(512), 96 * 1024, 0, d, o, iters, stream, smask);
            else hipLaunchKernelGGL((kl<4, true>), dim3(grid), dim3(512), 96 * 1024, 0, d, o, iters, stream, smask);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            elapsed += ms;
            rates.push_back((float)(flop / ms / 1e9));
        }
        const size_t n = rates.size();
        double tail = 0; size_t nt = 0;
        for (size_t i = n - n / 4; i < n; ++i) { tail += rates[i]; ++nt; }
        printf("%-58s launches %3zu  first %7.1f  min %7.1f  last-quarter mean %7.1f TFLOP/s  (%.3f of 2500; %.2f GHz-equivalent of matrix pipe)\n", v.name, n,
               rates[0], *std::min_element(rates.begin(), rates.end()), tail / nt, tail / nt / 2500.0, tail / nt / 2500.0 * 2.4);
        fflush(stdout);
    }
    return 0;
}

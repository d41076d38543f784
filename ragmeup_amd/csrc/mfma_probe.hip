// mfma_probe.hip -- rmu_probe_mfma_rate: what THIS GPU sustains on v_mfma_f32_32x32x16_{f16,bf16} when the operands are data.
//
// The roofs of MI355X_MICROARCH.md (2.5 PFLOP/s dense f16 / bf16) are reached by back-to-back MFMAs on constant operands (2.46-2.47 measured,
// tools/ubench/mfma_power.hip).  With operands that CHANGE from instruction to instruction -- random values, the image's distribution --
// the multiplier arrays toggle and the part is power-limited: 1.60-1.67 PFLOP/s f16, 1.70-1.76 bf16 with nothing but MFMAs in the loop;
// 1.44 with one 1-KiB LDS fragment read per MFMA and the screening kernel's LDS-DMA fill rate beside them (profiles/r06_mfma_power.txt).
// bench.py reports these next to the nominal peak (`roofline.sustained`), measured on the box and in the run the bench line comes from:
// the fraction of the NOMINAL peak is what the contract asks for, the fraction of the sustained rate says how much a kernel leaves.
// A diagnostic like rmu_last_scan_ms, not a step of the path.  (No reference counterpart.)
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "rmu_common.h"
#include "../../include/rmu.h"

extern "C" void rmu_set_error_(const char* msg);
static int pfail(int code, const std::string& m) { rmu_set_error_(m.c_str()); return code; }

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

extern __shared__ __attribute__((aligned(16))) char probe_lds[];

// 8 waves per workgroup (two per SIMD), 64 MFMAs per iteration on 8 accumulators; wave w holds B fragments 16 w + 8 .. 16 w + 15 of `src`
// in registers.  LDSF = false: the A fragments 16 w .. 16 w + 7 in registers as well (each feeds 8 consecutive MFMAs).  LDSF = true: the
// A operand of EVERY MFMA is a 1-KiB fragment out of LDS (64 KiB of fragments shared by the eight waves, read four ahead with counted
// waits) and each wave issues one 1-KiB LDS-DMA piece per 8 MFMAs into a region nobody reads (scan_screen_lean3_kernel's fill rate:
// 24 pieces per 32-row tile = 3 per wave and 24 MFMAs), from `stream` (64 MiB: mostly L2 / MALL hits).
template <bool BF, bool LDSF>
__global__ __launch_bounds__(512) void k_mfma_probe(const u32x4* __restrict__ src, float* __restrict__ out, int iters, const char* __restrict__ stream,
                                                    unsigned stream_mask) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    u32x4 a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        a[i] = src[(size_t)(w * 16 + i) * 64 + lane];
        b[i] = src[(size_t)(w * 16 + 8 + i) * 64 + lane];
    }
    f32x16 acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    auto mfma = [&](const u32x4& av, const u32x4& bv, f32x16& c) {
        if (BF) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv), c, 0, 0, 0);
        else c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, av), __builtin_bit_cast(f16x8, bv), c, 0, 0, 0);
    };
    if (!LDSF) {
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) mfma(a[i], b[(i + j) & 7], acc[j]);
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) *(u32x4*)(probe_lds + (size_t)((w * 8 + i) * 64 + lane) * 16) = a[i];
        __syncthreads();
        const unsigned lbase = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)probe_lds + (unsigned)lane * 16u;
        char* fill_dst = probe_lds + 64 * 1024 + w * 4096;
        unsigned fo = ((unsigned)blockIdx.x * 8u + (unsigned)w) * 65536u + (unsigned)lane * 16u;
        u32x4 f[4];
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f[r]) : "v"(lbase), "n"(((r * 17) & 63) * 1024));
#pragma unroll
            for (int n = 0; n < 64; ++n) {
                // fragment n has landed once at most min(3, 63 - n) younger reads are in flight
                if (63 - n >= 3) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(f[n & 3]));
                else if (63 - n == 2) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(f[n & 3]));
                else if (63 - n == 1) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(f[n & 3]));
                else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[n & 3]));
                mfma(f[n & 3], b[(n + (n >> 3)) & 7], acc[n & 7]);
                if (n + 4 < 64) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f[n & 3]) : "v"(lbase), "n"((((n + 4) * 17) & 63) * 1024));
                if ((n & 7) == 7) {
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(stream + (fo & stream_mask)),
                                                     (__attribute__((address_space(3))) void*)(fill_dst + ((n >> 3) & 3) * 1024), 16, 0, 0);
                    fo += 1024u;
                }
            }
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[j][e];
    out[(size_t)blockIdx.x * 512 + threadIdx.x] = s;
}

unsigned short to_f16(float f) { const _Float16 h = (_Float16)f; unsigned short u; memcpy(&u, &h, 2); return u; }
unsigned short to_bf16(float f) { unsigned int u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (unsigned short)(u >> 16); }

}  // namespace

extern "C" int rmu_probe_mfma_rate(int dtype, int variant, int millis, double* tflops_out) {
    RMU_ENTRY();
    if (!tflops_out || dtype < 0 || dtype > 1 || variant < 0 || variant > 1 || millis < 1 || millis > 10000)
        return pfail(RMU_E_INVALID, "rmu_probe_mfma_rate: dtype 0 (f16) | 1 (bf16), variant 0 (registers only) | 1 (LDS fragment per MFMA + LDS-DMA fill), 1 <= millis <= 10000");
    if (variant == 1 && dtype == 1) return pfail(RMU_E_INVALID, "rmu_probe_mfma_rate: the LDS / DMA skeleton is the f16 screening kernel's");
    int dev = 0, ncu = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu < 1)
        return pfail(RMU_E_HIP, "rmu_probe_mfma_rate: no device");
    // 128 fragments of 1 KiB, N(0, 3.3) -- fp16(64 x) of unit-vector elements in 384 dimensions (xorshift + Box-Muller: no libc state touched)
    const size_t nhalf = 128 * 512;
    std::vector<unsigned short> h(nhalf);
    unsigned long long st = 0x9E3779B97F4A7C15ull;
    auto u01 = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (double)((st >> 11) + 1) / 9007199254740994.0; };
    for (size_t i = 0; i < nhalf; ++i) {
        const float g = 3.3f * (float)(std::sqrt(-2.0 * std::log(u01())) * std::cos(6.283185307179586 * u01()));
        h[i] = dtype ? to_bf16(g) : to_f16(g);
    }
    hipStream_t s = nullptr;
    void *d = nullptr, *o = nullptr, *strm = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = RMU_OK;
    const unsigned sbytes = 64u << 20;
    auto cleanup = [&]() {
        if (s) (void)hipStreamSynchronize(s);
        for (void* p : {d, o, strm}) if (p) (void)hipFree(p);
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
        if (s) (void)hipStreamDestroy(s);
        (void)hipGetLastError();
    };
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess || hipMalloc(&d, nhalf * 2) != hipSuccess ||
        hipMalloc(&o, (size_t)ncu * 512 * sizeof(float)) != hipSuccess || (variant == 1 && hipMalloc(&strm, sbytes) != hipSuccess) ||
        hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess ||
        hipMemcpyAsync(d, h.data(), nhalf * 2, hipMemcpyHostToDevice, s) != hipSuccess ||
        (variant == 1 && hipMemsetAsync(strm, 1, sbytes, s) != hipSuccess) || hipStreamSynchronize(s) != hipSuccess) {
        cleanup();
        return pfail(RMU_E_HIP, "rmu_probe_mfma_rate: set-up");
    }
    const int iters = 2000;                                        // 128k MFMAs per wave and launch: ~2-3 ms
    const double flop = (double)ncu * 8 * iters * 64 * 32768.0;
    std::vector<double> rates;
    double elapsed = 0.0;
    while (elapsed < (double)millis) {
        (void)hipEventRecord(e0, s);
        if (variant == 1) hipLaunchKernelGGL((k_mfma_probe<false, true>), dim3(ncu), dim3(512), 96 * 1024, s, (const u32x4*)d, (float*)o, iters, (const char*)strm, sbytes - 1024u);
        else if (dtype) hipLaunchKernelGGL((k_mfma_probe<true, false>), dim3(ncu), dim3(512), 0, s, (const u32x4*)d, (float*)o, iters, (const char*)nullptr, 0u);
        else hipLaunchKernelGGL((k_mfma_probe<false, false>), dim3(ncu), dim3(512), 0, s, (const u32x4*)d, (float*)o, iters, (const char*)nullptr, 0u);
        float ms = 0.f;
        if (hipEventRecord(e1, s) != hipSuccess || hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess || !(ms > 0.f)) { rc = RMU_E_HIP; break; }
        elapsed += ms;
        rates.push_back(flop / ms / 1e9);
    }
    cleanup();
    if (rc || rates.empty()) return pfail(RMU_E_HIP, "rmu_probe_mfma_rate: launch / timing");
    // the power manager's steady state: mean over the last half of the launches (the first ones of a burst run 10-15 % slower)
    double sum = 0.0; size_t n = 0;
    for (size_t i = rates.size() / 2; i < rates.size(); ++i) { sum += rates[i]; ++n; }
    *tflops_out = sum / (double)n;
    return RMU_OK;
}

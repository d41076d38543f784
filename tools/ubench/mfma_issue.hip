// mfma_issue.hip -- what does ONE wave per SIMD get out of back-to-back v_mfma_f32_32x32x16_bf16, depending on where the operands
// live?  (round 3: the fused FFN kernel's bare MFMA stream -- no DMA, no fragment reads, no activation -- measured 50 cycles
// per MFMA instead of 32.)  Variants: A/B register alignment (VGPR bank = index mod 4), B operand in AGPRs, accumulators in
// VGPRs, 12 independent accumulators vs 2 alternating ones, an s_waitcnt between MFMAs, an s_nop between MFMAs.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/mfma_issue tools/ubench/mfma_issue.hip && tools/ubench/mfma_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP12(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11)

extern __shared__ __attribute__((aligned(16))) char lds_[];

template <int V>
__global__ __launch_bounds__(256) void k(unsigned long long* out, int iters, const char* w, unsigned int wbytes, unsigned int pattern) {
    unsigned long long t0 = 0, t1 = 0;
    const unsigned int lds_base = (unsigned int)(unsigned long long)(__attribute__((address_space(3))) char*)lds_ + (threadIdx.x >> 6) * 1024u;
    unsigned int voff = (threadIdx.x & 63) * 16u + (threadIdx.x >> 6) * 1024u;
    unsigned int slot = 0;
    // zero the registers the variants use (values do not matter: zeros keep the clock high and equal across variants)
    asm volatile(
        "v_mov_b32 v0, 0\n v_mov_b32 v1, 0\n v_mov_b32 v2, 0\n v_mov_b32 v3, 0\n v_mov_b32 v4, 0\n v_mov_b32 v5, 0\n v_mov_b32 v6, 0\n"
        "v_mov_b32 v7, 0\n v_mov_b32 v8, 0\n v_mov_b32 v9, 0\n v_mov_b32 v10, 0\n v_mov_b32 v11, 0\n"
        ::: "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11");
    if (V == 13 || V == 14) {   // non-zero operands: bf16 pairs from `pattern` in A / B, 1.0f in the accumulators (power, not timing, differs)
        asm volatile("v_mov_b32 v0, %0\n v_mov_b32 v1, %0\n v_mov_b32 v2, %0\n v_mov_b32 v3, %0\n v_mov_b32 v4, %1\n v_mov_b32 v5, %1\n v_mov_b32 v6, %1\n v_mov_b32 v7, %1"
                     ::"s"(pattern), "s"(pattern * 2654435761u) : "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7");
    }
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0));
    for (int i = 0; i < iters; ++i) {
        if (V == 0) {        // A v[0:3], B v[4:7] (both 4-aligned), 12 AGPR accumulators
#define M(j) asm volatile("v_mfma_f32_32x32x16_bf16 a[%c0:%c1], v[0:3], v[4:7], a[%c0:%c1]" ::"i"(16 * j), "i"(16 * j + 15));
            REP12(M)
#undef M
        } else if (V == 1) { // B offset by 2 registers
#define M(j) asm volatile("v_mfma_f32_32x32x16_bf16 a[%c0:%c1], v[0:3], v[6:9], a[%c0:%c1]" ::"i"(16 * j), "i"(16 * j + 15));
            REP12(M)
#undef M
        } else if (V == 2) { // A offset by 2, B aligned
#define M(j) asm volatile("v_mfma_f32_32x32x16_bf16 a[%c0:%c1], v[2:5], v[8:11], a[%c0:%c1]" ::"i"(16 * j), "i"(16 * j + 15));
            REP12(M)
#undef M
        } else if (V == 3) { // B operand in AGPRs
#define M(j) asm volatile("v_mfma_f32_32x32x16_bf16 a[%c0:%c1], v[0:3], a[200:203], a[%c0:%c1]" ::"i"(16 * j), "i"(16 * j + 15));
            REP12(M)
#undef M
        } else if (V == 4) { // accumulators in VGPRs (v[16..207])
#define M(j) asm volatile("v_mfma_f32_32x32x16_bf16 v[%c0:%c1], v[0:3], v[4:7], v[%c0:%c1]" ::"i"(16 + 16 * j), "i"(16 + 16 * j + 15));
            REP12(M)
#undef M
        } else if (V == 5) { // two alternating accumulators (dependent distance 2)
#define M(j) asm volatile("v_mfma_f32_32x32x16_bf16 a[%c0:%c1], v[0:3], v[4:7], a[%c0:%c1]" ::"i"(16 * (j & 1)), "i"(16 * (j & 1) + 15));
            REP12(M)
#undef M
        } else if (V == 6) { // a counted wait in front of every MFMA (nothing outstanding: it only costs an issue slot)
#define M(j) asm volatile("s_waitcnt lgkmcnt(3)\n v_mfma_f32_32x32x16_bf16 a[%c0:%c1], v[0:3], v[4:7], a[%c0:%c1]" ::"i"(16 * j), "i"(16 * j + 15));
            REP12(M)
#undef M
        } else if (V == 7) { // an s_nop 0 on both sides of a wait (what hipcc put around the asm waits)
#define M(j) asm volatile("s_nop 0\n s_waitcnt lgkmcnt(3)\n s_nop 0\n v_mfma_f32_32x32x16_bf16 a[%c0:%c1], v[0:3], v[4:7], a[%c0:%c1]" ::"i"(16 * j), "i"(16 * j + 15));
            REP12(M)
#undef M
        } else if (V == 8) { // five independent VALU fillers behind every MFMA
#define M(j) asm volatile("v_mfma_f32_32x32x16_bf16 a[%c0:%c1], v[0:3], v[4:7], a[%c0:%c1]\n v_fma_f32 v8, v8, v8, v8\n v_fma_f32 v9, v9, v9, v9\n v_fma_f32 v10, v10, v10, v10\n v_fma_f32 v11, v11, v11, v11\n v_fma_f32 v8, v8, v8, v8" ::"i"(16 * j), "i"(16 * j + 15));
            REP12(M)
#undef M
        } else if (V == 10 || V == 11 || V == 12 || V == 14) {
            // the fused FFN kernel's skeleton: wait + MFMA + ds_read_b128 per step, one 1-KiB LDS-DMA piece per wave every 4th step
            // (a 5 x 24 KiB ring, L2-resident source shared by every CU), counted vmcnt + s_barrier every 16 steps;
            // 11: + five v_fma_f32 per step; 12: no barrier; 14: as 11 with non-zero operands
            asm volatile("v_mov_b32 v12, 0" ::: "v12");
#define STEP(j) asm volatile("s_waitcnt lgkmcnt(3)\n v_mfma_f32_32x32x16_bf16 a[%c0:%c1], v[0:3], v[4:7], a[%c0:%c1]\n ds_read_b128 v[208:211], v12" ::"i"(16 * j), "i"(16 * j + 15)); \
            if (V == 11 || V == 14) asm volatile("v_fma_f32 v8, v8, v8, v8\n v_fma_f32 v9, v9, v9, v9\n v_fma_f32 v10, v10, v10, v10\n v_fma_f32 v11, v11, v11, v11\n v_fma_f32 v8, v8, v8, v8"); \
            if ((j & 3) == 0) { \
                asm volatile("s_mov_b32 m0, %1\n s_nop 0\n global_load_lds_dwordx4 %0, %2" ::"v"(voff), "s"(__builtin_amdgcn_readfirstlane((int)(lds_base + slot * 24576u))), "s"(w) : "memory"); \
                voff += 4096u; if (voff >= wbytes) voff -= wbytes; \
            }
            REP12(STEP)
#undef STEP
            slot = slot == 4 ? 0 : slot + 1;
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            if (V != 12) __builtin_amdgcn_s_barrier();
        } else if (V == 13) { // plain 12-accumulator stream, non-zero operands
#define M(j) asm volatile("v_mfma_f32_32x32x16_bf16 a[%c0:%c1], v[0:3], v[4:7], a[%c0:%c1]" ::"i"(16 * j), "i"(16 * j + 15));
            REP12(M)
#undef M
        } else if (V == 9) { // one ds_read_b128 behind every MFMA (+ the counted wait in front)
#define M(j) asm volatile("s_waitcnt lgkmcnt(3)\n v_mfma_f32_32x32x16_bf16 a[%c0:%c1], v[0:3], v[4:7], a[%c0:%c1]\n ds_read_b128 v[208:211], v12" ::"i"(16 * j), "i"(16 * j + 15));
            asm volatile("v_mov_b32 v12, 0" ::: "v12");
            REP12(M)
#undef M
        }
    }
    asm volatile("s_waitcnt vmcnt(0)\n s_nop 15\n s_nop 15\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

static char* g_w = nullptr;
static const unsigned int WBYTES = 2359296;       // the FFN weights of one layer (bf16)

#define REP6(X) X(0) X(1) X(2) X(3) X(4) X(5)
// the FFN skeleton (+ five v_fma_f32 per MFMA when VALU) with 6 accumulators: <= 128 registers, so NW = 4 (one wave per SIMD) and
// NW = 8 (two per SIMD) run the same instruction stream per wave; DMA pieces per wave scale so that a workgroup moves 3 KiB per
// 6 MFMA steps of every wave either way
template <int NW, int VALU>
__global__ __launch_bounds__(64 * NW) void k2(unsigned long long* out, int iters, const char* w, unsigned int wbytes) {
    unsigned long long t0 = 0, t1 = 0;
    const unsigned int lds_base = (unsigned int)(unsigned long long)(__attribute__((address_space(3))) char*)lds_ + (threadIdx.x >> 6) * 1024u;
    unsigned int voff = (threadIdx.x & 63) * 16u + (threadIdx.x >> 6) * 1024u;
    unsigned int slot = 0;
    asm volatile("v_mov_b32 v0, 0\n v_mov_b32 v1, 0\n v_mov_b32 v2, 0\n v_mov_b32 v3, 0\n v_mov_b32 v4, 0\n v_mov_b32 v5, 0\n v_mov_b32 v6, 0\n v_mov_b32 v7, 0\n"
                 "v_mov_b32 v8, 0\n v_mov_b32 v9, 0\n v_mov_b32 v10, 0\n v_mov_b32 v11, 0\n v_mov_b32 v12, 0"
                 ::: "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12");
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0));
    for (int i = 0; i < iters; ++i) {
#define STEP(j) asm volatile("s_waitcnt lgkmcnt(3)\n v_mfma_f32_32x32x16_bf16 a[%c0:%c1], v[0:3], v[4:7], a[%c0:%c1]\n ds_read_b128 v[16:19], v12" ::"i"(16 * j), "i"(16 * j + 15)); \
        if (VALU) asm volatile("v_fma_f32 v8, v8, v8, v8\n v_fma_f32 v9, v9, v9, v9\n v_fma_f32 v10, v10, v10, v10\n v_fma_f32 v11, v11, v11, v11\n v_fma_f32 v8, v8, v8, v8"); \
        if ((j % (NW == 8 ? 6 : 3)) == 0) { \
            asm volatile("s_mov_b32 m0, %1\n s_nop 0\n global_load_lds_dwordx4 %0, %2" ::"v"(voff), "s"(__builtin_amdgcn_readfirstlane((int)(lds_base + slot * 24576u))), "s"(w) : "memory"); \
            voff += 1024u * NW; if (voff >= wbytes) voff -= wbytes; \
        }
        REP6(STEP)
        REP6(STEP)
#undef STEP
        slot = slot == 4 ? 0 : slot + 1;
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)\n s_nop 15\n s_nop 15\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * NW + (threadIdx.x >> 6)] = t1 - t0;
}

template <int V>
static void run(const char* name, unsigned long long* d, int iters) {
    const int grid = 256;
    const int lds = 5 * 24576;
    hipFuncSetAttribute((const void*)k<V>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(k<V>, dim3(grid), dim3(256), lds, 0, d, 8, g_w, WBYTES, 0x3f9d3e4cu);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<V>, dim3(grid), dim3(256), lds, 0, d, iters, g_w, WBYTES, 0x3f9d3e4cu);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(grid * 4);
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    double s = 0;
    for (auto v : h) s += (double)v;
    const double per = s / h.size() / ((double)iters * 12);
    const double tf = (double)grid * 4 * iters * 12 * 32768.0 / (ms * 1e-3) / 1e12;
    printf("%-64s %7.1f memtime ticks / MFMA   %7.3f ms   %7.1f TFLOP/s  (=> %.1f ns / MFMA / SIMD)\n", name, per, ms, tf, ms * 1e6 / ((double)iters * 12));
}

template <int NW, int VALU>
static void run2(const char* name, unsigned long long* d, int iters) {
    const int grid = 256, lds = 5 * 24576;
    hipFuncSetAttribute((const void*)k2<NW, VALU>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL((k2<NW, VALU>), dim3(grid), dim3(64 * NW), lds, 0, d, 8, g_w, WBYTES);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k2<NW, VALU>), dim3(grid), dim3(64 * NW), lds, 0, d, iters, g_w, WBYTES);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(grid * NW);
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    double s = 0;
    for (auto v : h) s += (double)v;
    const double per_wave = s / h.size() / ((double)iters * 12);
    const double tf = (double)grid * NW * iters * 12 * 32768.0 / (ms * 1e-3) / 1e12;
    printf("%-64s %7.1f ticks / MFMA / wave = %5.1f per SIMD   %7.3f ms   %7.1f TFLOP/s\n", name, per_wave, per_wave / (NW / 4), ms, tf);
}

int main() {
    unsigned long long* d;
    hipMalloc(&d, 1024 * 8 * 8);
    hipMalloc((void**)&g_w, WBYTES + 8192);
    hipMemset(g_w, 0x3c, WBYTES + 8192);
    const int iters = 20000;
    run<0>("12 acc (AGPR), A v[0:3], B v[4:7]", d, iters);
    run<1>("12 acc (AGPR), A v[0:3], B v[6:9]  (B offset 2)", d, iters);
    run<2>("12 acc (AGPR), A v[2:5], B v[8:11] (A offset 2)", d, iters);
    run<3>("12 acc (AGPR), A v[0:3], B a[200:203]", d, iters);
    run<4>("12 acc (VGPR), A v[0:3], B v[4:7]", d, iters);
    run<5>("2 alternating acc (AGPR)", d, iters);
    run<6>("12 acc + s_waitcnt lgkmcnt(3) before each MFMA", d, iters);
    run<7>("12 acc + s_nop 0, s_waitcnt, s_nop 0 before each MFMA", d, iters);
    run<8>("12 acc + 5 independent v_fma_f32 behind each MFMA", d, iters);
    run<9>("12 acc + wait + ds_read_b128 per MFMA", d, iters);
    run<10>("FFN skeleton: + 1 KiB LDS-DMA / 4 MFMAs, vmcnt + barrier / 12", d, iters);
    run<11>("FFN skeleton + 5 v_fma_f32 per MFMA", d, iters);
    run<12>("FFN skeleton without the barrier", d, iters);
    run<13>("plain 12-acc stream, NON-ZERO operands", d, iters);
    run<14>("FFN skeleton + 5 v_fma_f32, NON-ZERO operands", d, iters);
    run2<4, 0>("6-acc skeleton, 4 waves (1 per SIMD)", d, iters);
    run2<8, 0>("6-acc skeleton, 8 waves (2 per SIMD)", d, iters);
    run2<4, 1>("6-acc skeleton + 5 v_fma_f32, 4 waves (1 per SIMD)", d, iters);
    run2<8, 1>("6-acc skeleton + 5 v_fma_f32, 8 waves (2 per SIMD)", d, iters);
    return 0;
}

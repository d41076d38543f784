// What does a dependent kernel boundary cost inside a replayed hipGraph, and what does a grid barrier cost inside ONE persistent kernel?
// (round 5: the question behind "the interactive forward as one cooperative launch" -- a 16-token query forward is 33 dependent launches
// replayed in 0.18 ms.)  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/graph_chain tools/ubench/graph_chain.hip && tools/ubench/graph_chain
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void k_empty(int* p) { if (threadIdx.x == 9999) p[0] = 1; }
// a dependent round trip per kernel: every workgroup reads what the previous kernel's workgroups wrote, adds, writes
__global__ void k_chain(const float* __restrict__ in, float* __restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[(i * 17 + 5) % n] + 1.0f;
}
// persistent kernel: `rounds` grid barriers among the active workgroups (active = blockIdx % stride == 0), each round the same dependent
// round trip as k_chain.  Barrier: agent-scope release / acquire around an atomic counter (monotonic: no reset races).
__global__ void k_persist(float* a, float* b, int n, int rounds, int stride, unsigned* counter, int active_wgs) {
    if (blockIdx.x % stride) return;
    const int wg = blockIdx.x / stride;
    float* in = a; float* out = b;
    for (int r = 0; r < rounds; ++r) {
        for (int i = wg * blockDim.x + threadIdx.x; i < n; i += active_wgs * blockDim.x) out[i] = in[(i * 17 + 5) % n] + 1.0f;
        __syncthreads();
        if (threadIdx.x == 0) {
            __atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE);                      // agent scope by default for global atomics
            const unsigned want = (unsigned)(r + 1) * (unsigned)active_wgs;
            while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
        float* t = in; in = out; out = t;
    }
}

int main() {
    const int n = 16 * 384;                       // a 16-token activation
    float *a, *b; int* p; unsigned* cnt;
    CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMalloc(&p, 4)); CK(hipMalloc(&cnt, 4));
    CK(hipMemset(a, 0, n * 4)); CK(hipMemset(b, 0, n * 4));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    auto replay_us = [&](hipGraphExec_t ge, int reps) {
        for (int i = 0; i < 20; ++i) (void)hipGraphLaunch(ge, s);
        (void)hipStreamSynchronize(s);
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < reps; ++i) { (void)hipGraphLaunch(ge, s); (void)hipStreamSynchronize(s); }
        return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
    };
    for (int kind = 0; kind < 3; ++kind)
        for (int nk : {1, 8, 34, 68}) {
            hipGraph_t g; hipGraphExec_t ge;
            CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
            for (int i = 0; i < nk; ++i) {
                if (kind == 0) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s, p);
                else if (kind == 1) hipLaunchKernelGGL(k_empty, dim3(36), dim3(256), 0, s, p);
                else hipLaunchKernelGGL(k_chain, dim3(n / 256), dim3(256), 0, s, (i & 1) ? b : a, (i & 1) ? a : b, n);
            }
            CK(hipStreamEndCapture(s, &g));
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            const double us = replay_us(ge, 300);
            printf("graph of %2d %s kernels: %.1f us per replay (+ sync) = %.2f us per kernel\n", nk,
                   kind == 0 ? "empty 1-wave      " : kind == 1 ? "empty 36-workgroup" : "dependent-chain   ", us, us / nk);
            (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g);
        }
    for (int stride : {8, 1})
        for (int wgs : {12, 32, 256}) {
            if (stride == 8 && wgs > 32) continue;
            const int grid = wgs * stride;
            for (int rounds : {1, 34, 136}) {
                CK(hipMemsetAsync(cnt, 0, 4, s));
                hipLaunchKernelGGL(k_persist, dim3(grid), dim3(256), 0, s, a, b, n, rounds, stride, cnt, wgs);
                CK(hipStreamSynchronize(s));
                const auto t0 = std::chrono::steady_clock::now();
                const int reps = 100;
                for (int i = 0; i < reps; ++i) {
                    (void)hipMemsetAsync(cnt, 0, 4, s);
                    hipLaunchKernelGGL(k_persist, dim3(grid), dim3(256), 0, s, a, b, n, rounds, stride, cnt, wgs);
                    (void)hipStreamSynchronize(s);
                }
                const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
                printf("persistent kernel, %3d active workgroups (%s), %3d rounds of [dependent round trip + grid barrier]: %.1f us per launch (+ memset + sync)\n",
                       wgs, stride == 8 ? "all on ONE XCD: blockIdx % 8 == 0" : "spread over the 8 XCDs         ", rounds, us);
            }
        }
    return 0;
}

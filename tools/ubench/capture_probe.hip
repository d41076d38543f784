// capture_probe.hip -- which HIP calls of a SECOND thread disturb a hipGraph capture running on a first thread (ROCm 7 / gfx950)?
// Thread A: begin capture (mode M_A) on its stream, launch a kernel, wait for B, launch another, end capture, instantiate, replay, check.
// Thread B (while A is inside the capture): one call under test, with its thread's capture-interaction mode left at the default (global) or
// exchanged to relaxed (hipThreadExchangeStreamCaptureMode) -- what PyTorch's allocator does around cudaMalloc.
//   hipcc --offload-arch=gfx950 -O2 -o tools/ubench/capture_probe tools/ubench/capture_probe.hip -lpthread
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdio>
#include <thread>
#include <vector>

__global__ void k_inc(int* p) { atomicAdd(p, 1); }

static const char* mode_name(hipStreamCaptureMode m) {
    return m == hipStreamCaptureModeGlobal ? "global" : (m == hipStreamCaptureModeThreadLocal ? "thread-local" : "relaxed");
}

struct Call { const char* name; int id; };

int main() {
    hipSetDevice(0);
    int* counter; hipMalloc(&counter, 4);
    void* scratch; hipMalloc(&scratch, 1 << 20);
    char* hsrc = (char*)malloc(4096);
    hipStream_t sB; hipStreamCreateWithFlags(&sB, hipStreamNonBlocking);
    hipEvent_t evB; hipEventCreateWithFlags(&evB, hipEventDisableTiming);
    const Call calls[] = {{"hipMalloc", 0}, {"hipFree", 1}, {"hipDeviceSynchronize", 2}, {"hipStreamSynchronize(other stream)", 3},
                          {"hipEventSynchronize(event on other stream)", 4}, {"hipMemcpy (synchronous, pageable H2D)", 5},
                          {"hipMemcpyAsync + hipStreamSynchronize", 6}, {"hipMallocAsync + hipFreeAsync", 7}, {"hipEventQuery", 8},
                          {"hipHostMalloc + hipHostFree", 9}, {"hipMemsetAsync(other stream)", 10}, {"hipStreamCreate + Destroy", 11}};
    const hipStreamCaptureMode modesA[] = {hipStreamCaptureModeGlobal, hipStreamCaptureModeThreadLocal, hipStreamCaptureModeRelaxed};
    printf("%-46s %-13s %-8s | %-28s | capture\n", "call on thread B", "A's mode", "B's mode", "B's result");
    for (hipStreamCaptureMode mA : modesA)
        for (int relaxedB = 0; relaxedB < 2; ++relaxedB)
            for (const Call& c : calls) {
                hipStream_t sA; hipStreamCreateWithFlags(&sA, hipStreamNonBlocking);
                std::atomic<int> stage{0};
                hipError_t eB = hipSuccess, eEnd = hipSuccess, eInst = hipSuccess, eRun = hipSuccess, eBegin = hipSuccess;
                int got = -1;
                std::thread A([&] {
                    hipSetDevice(0);
                    hipMemsetAsync(counter, 0, 4, sA); hipStreamSynchronize(sA);
                    eBegin = hipStreamBeginCapture(sA, mA);
                    hipLaunchKernelGGL(k_inc, dim3(1), dim3(1), 0, sA, counter);
                    stage = 1;
                    while (stage != 2) std::this_thread::yield();
                    hipLaunchKernelGGL(k_inc, dim3(1), dim3(1), 0, sA, counter);
                    hipGraph_t g = nullptr; hipGraphExec_t ge = nullptr;
                    eEnd = hipStreamEndCapture(sA, &g);
                    if (eEnd == hipSuccess && g) {
                        eInst = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
                        if (eInst == hipSuccess) {
                            eRun = hipGraphLaunch(ge, sA);
                            if (eRun == hipSuccess) eRun = hipStreamSynchronize(sA);
                            hipMemcpy(&got, counter, 4, hipMemcpyDeviceToHost);
                            hipGraphExecDestroy(ge);
                        }
                        hipGraphDestroy(g);
                    }
                    (void)hipGetLastError();
                });
                std::thread B([&] {
                    hipSetDevice(0);
                    while (stage != 1) std::this_thread::yield();
                    hipStreamCaptureMode m = hipStreamCaptureModeRelaxed;
                    if (relaxedB) hipThreadExchangeStreamCaptureMode(&m);
                    void* p = nullptr;
                    switch (c.id) {
                        case 0: eB = hipMalloc(&p, 1 << 20); break;
                        case 1: { hipStreamCaptureMode r = hipStreamCaptureModeRelaxed; if (!relaxedB) hipThreadExchangeStreamCaptureMode(&r);
                                  hipMalloc(&p, 1 << 20); if (!relaxedB) hipThreadExchangeStreamCaptureMode(&r); eB = hipFree(p); p = nullptr; break; }
                        case 2: eB = hipDeviceSynchronize(); break;
                        case 3: eB = hipStreamSynchronize(sB); break;
                        case 4: hipEventRecord(evB, sB); eB = hipEventSynchronize(evB); break;
                        case 5: eB = hipMemcpy(scratch, hsrc, 4096, hipMemcpyHostToDevice); break;
                        case 6: eB = hipMemcpyAsync(scratch, hsrc, 4096, hipMemcpyHostToDevice, sB); if (eB == hipSuccess) eB = hipStreamSynchronize(sB); break;
                        case 7: eB = hipMallocAsync(&p, 1 << 20, sB); if (eB == hipSuccess) { eB = hipFreeAsync(p, sB); p = nullptr; hipStreamSynchronize(sB); } break;
                        case 8: hipEventRecord(evB, sB); eB = hipEventQuery(evB); if (eB == hipErrorNotReady) eB = hipSuccess; break;
                        case 9: eB = hipHostMalloc(&p, 4096); if (eB == hipSuccess) { eB = hipHostFree(p); p = nullptr; } break;
                        case 10: eB = hipMemsetAsync(scratch, 0, 4096, sB); break;
                        case 11: { hipStream_t t; eB = hipStreamCreateWithFlags(&t, hipStreamNonBlocking); if (eB == hipSuccess) eB = hipStreamDestroy(t); break; }
                    }
                    (void)hipGetLastError();
                    if (relaxedB) hipThreadExchangeStreamCaptureMode(&m);
                    if (p) { hipStreamCaptureMode r = hipStreamCaptureModeRelaxed; hipThreadExchangeStreamCaptureMode(&r); hipFree(p); hipThreadExchangeStreamCaptureMode(&r); }
                    stage = 2;
                });
                A.join(); B.join();
                const bool ok = eBegin == hipSuccess && eEnd == hipSuccess && eInst == hipSuccess && eRun == hipSuccess && got == 2;
                printf("%-46s %-13s %-8s | %-28s | %s%s%s\n", c.name, mode_name(mA), relaxedB ? "relaxed" : "default", hipGetErrorName(eB),
                       ok ? "OK" : "BROKEN: ", ok ? "" : hipGetErrorName(eEnd != hipSuccess ? eEnd : (eInst != hipSuccess ? eInst : eRun)),
                       (!ok && got != 2 && eEnd == hipSuccess && eInst == hipSuccess && eRun == hipSuccess) ? " (wrong count)" : "");
                fflush(stdout);
                // a stream left in capture mode cannot be destroyed cleanly: end it if needed
                hipStreamCaptureStatus st; if (hipStreamIsCapturing(sA, &st) == hipSuccess && st != hipStreamCaptureStatusNone) { hipGraph_t g; hipStreamEndCapture(sA, &g); }
                (void)hipGetLastError();
                hipStreamDestroy(sA);
            }
    return 0;
}

// cross_stream_probe.hip — what does a dependency between two HIP streams cost on MI355X, and does the producer's queue pay?
// Pattern of the space-charge kick: stream A runs P (producer), then C (continues, independent of B); stream B runs W, which
// needs P's result. Two ways to express "W after P":
//   event : hipEventRecord(e, A) after P; hipStreamWaitEvent(B, e)
//   value : P's last lane stores an epoch to a flag word; hipStreamWaitValue32(B, flag, epoch, >=)   (no packet on A)
// Run under `rocprofv3 --kernel-trace` and read the timeline (profiles/summarize is not needed: the program prints the
// host-side wall time per iteration; the trace gives P.end -> W.start and P.end -> C.start).
// Build: hipcc --offload-arch=gfx950 -O3 -o cross_stream_probe benchmarks/cross_stream_probe.hip
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void busy_P(float* buf, int n, unsigned* flag, unsigned* counter, unsigned epoch) {   // ~10 us of streaming work, then the flag
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    for (int k = i; k < n; k += gridDim.x * blockDim.x) buf[k] = buf[k] * 1.0001f + 1.0f;
    if (flag) {
        __threadfence();
        __shared__ unsigned done;
        // last workgroup to arrive publishes the epoch
        if (threadIdx.x == 0) done = atomicAdd(counter, 1u);    // ordinary device memory: the flag word is uncached
        __syncthreads();
        if (threadIdx.x == 0 && done == gridDim.x - 1) {
            *counter = 0;
            __hip_atomic_store(flag, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
__global__ void busy_C(float* buf, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    for (int k = i; k < n; k += gridDim.x * blockDim.x) buf[k] = buf[k] * 0.9999f + 2.0f;
}
__global__ void busy_W(const float* src, float* dst, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    for (int k = i; k < n; k += gridDim.x * blockDim.x) dst[k] = src[k] + 3.0f;
}

int main(int argc, char** argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 0;   // 0: events, 1: wait-value
    const int n = 1 << 24, iters = 200;
    float *a, *b, *c;
    CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMalloc(&c, n * 4));
    CK(hipMemset(a, 0, n * 4)); CK(hipMemset(b, 0, n * 4)); CK(hipMemset(c, 0, n * 4));
    unsigned* flag = nullptr;
    unsigned* counter = nullptr;
    CK(hipMalloc(&counter, 4)); CK(hipMemset(counter, 0, 4));
    if (mode == 1) {
        int can = 0;
        CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
        printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
        if (!can) return 0;
        if (hipExtMallocWithFlags((void**)&flag, 8, hipMallocSignalMemory) != hipSuccess) {
            printf("signal memory not available, using plain device memory\n");
            CK(hipMalloc(&flag, 8));
        }
        CK(hipMemset(flag, 0, 8));
    }
    hipStream_t A, B;
    CK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking));
    hipEvent_t fork, join;
    CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
    auto t0 = std::chrono::steady_clock::now();
    double wait_call_us = 0.0;
    for (int it = 1; it <= iters + 20; ++it) {
        if (it == 21) { CK(hipDeviceSynchronize()); t0 = std::chrono::steady_clock::now(); }
        hipLaunchKernelGGL(busy_P, dim3(1024), dim3(256), 0, A, a, n, flag, counter, (unsigned)it);
        if (mode == 0) {
            CK(hipEventRecord(fork, A));
            CK(hipStreamWaitEvent(B, fork, 0));
        } else {
            const auto w0 = std::chrono::steady_clock::now();
            CK(hipStreamWaitValue32(B, flag, (unsigned)it, hipStreamWaitValueGte, 0xffffffffu));
            wait_call_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - w0).count();
        }
        hipLaunchKernelGGL(busy_W, dim3(1024), dim3(256), 0, B, a, b, n);
        hipLaunchKernelGGL(busy_C, dim3(1024), dim3(256), 0, A, c, n);
        hipLaunchKernelGGL(busy_C, dim3(1024), dim3(256), 0, A, c, n);
        // B's result is needed by A at the end of the iteration (the kick's join): always an event here
        CK(hipEventRecord(join, B));
        CK(hipStreamWaitEvent(A, join, 0));
    }
    CK(hipDeviceSynchronize());
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / iters;
    if (mode == 1) printf("hipStreamWaitValue32 itself: %.1f us of host time per call\n", wait_call_us / (iters + 20));
    printf("mode %s: %.1f us per iteration (P, then W on the other stream next to C, C; join)\n", mode ? "wait-value" : "events", us);
    return 0;
}

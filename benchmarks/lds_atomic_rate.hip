// lds_atomic_rate.hip — LDS atomic throughput per CU on gfx950 by data type (random cells of a 1024-cell tile).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
__device__ __forceinline__ unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

template <typename T, int CELLS>
__global__ __launch_bounds__(1024) void k(T* out, int iters) {
    __shared__ T tile[CELLS];
    for (int i = threadIdx.x; i < CELLS; i += blockDim.x) tile[i] = (T)0;
    __syncthreads();
    unsigned h = hash(blockIdx.x * 1024 + threadIdx.x);
    for (int i = 0; i < iters; ++i) {
        h = h * 1664525u + 1013904223u;
        if constexpr (sizeof(T) == 4 && !__is_floating_point(T)) atomicAdd(&tile[(h >> 8) % CELLS], (T)1);
        else if constexpr (!__is_floating_point(T)) atomicAdd((unsigned long long*)&tile[(h >> 8) % CELLS], 1ull);
        else unsafeAtomicAdd(&tile[(h >> 8) % CELLS], (T)1);
    }
    __syncthreads();
    T s = 0;
    for (int i = threadIdx.x; i < CELLS; i += blockDim.x) s += tile[i];
    if (s == (T)12345) out[blockIdx.x] = s;
}

template <typename T, int CELLS>
void run(const char* name, int threads) {
    T* out; CK(hipMalloc(&out, 4096 * sizeof(T)));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 2000, blocks = 256;
    hipLaunchKernelGGL((k<T, CELLS>), dim3(blocks), dim3(threads), 0, 0, out, 10);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<T, CELLS>), dim3(blocks), dim3(threads), 0, 0, out, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double per_cu = (double)iters * threads / (ms * 1e-3);  // one block per CU
    printf("%-10s cells=%5d threads=%4d  %8.1f us  %7.2f G lane-atomics/s per CU  (%.2f cycles per lane-atomic @2.4GHz)\n",
           name, CELLS, threads, ms * 1e3, per_cu / 1e9, 2.4e9 / per_cu);
    CK(hipFree(out));
}

int main() {
    for (int threads : {256, 1024}) {
        run<float, 1024>("f32", threads);
        run<float, 512>("f32", threads);
        run<double, 1024>("f64", threads);
        run<unsigned, 1024>("u32", threads);
        run<unsigned long long, 1024>("u64", threads);
    }
    return 0;
}

// atomic_scope.hip — do workgroup-scope float atomics execute in the XCD-local L2 on gfx950, and how much
// faster are they than agent-scope atomics (which go to the memory side)?  Used to design the
// XCD-privatised cloud-in-cell deposit (one grid copy per XCD, picked by HW_REG_XCC_ID).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics benchmarks/atomic_scope.hip -o build/atomic_scope
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}

__device__ __forceinline__ unsigned hash(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

// MODE 0: agent scope into one grid; MODE 1: workgroup scope into grid copy [xcc]; MODE 2: agent scope into copy [xcc]
template <int MODE>
__global__ __launch_bounds__(256) void k_atomics(float* __restrict__ grid, unsigned cells, unsigned long n,
                                                 unsigned* __restrict__ xcc_hist) {
    const unsigned xcc = xcc_id();
    if (threadIdx.x == 0) atomicAdd(&xcc_hist[xcc * 64 + (blockIdx.x % 8)], 1u);
    float* g = (MODE == 0) ? grid : grid + (size_t)xcc * cells;
    for (unsigned long i = (unsigned long)blockIdx.x * 256 + threadIdx.x; i < n; i += (unsigned long)gridDim.x * 256) {
        const unsigned c = hash((unsigned)i) % cells;
        if (MODE == 1) __hip_atomic_fetch_add(g + c, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else __hip_atomic_fetch_add(g + c, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

__global__ void k_reduce(const float* __restrict__ copies, float* __restrict__ out, unsigned cells) {
    for (unsigned c = blockIdx.x * 256 + threadIdx.x; c < cells; c += gridDim.x * 256) {
        float s = 0.f;
        for (int k = 0; k < 8; ++k) s += copies[(size_t)k * cells + c];
        out[c] = s;
    }
}

int main() {
    const unsigned long n = 8000000;  // atomics per launch (like 1e6 particles x 8 corners)
    for (unsigned cells : {32u * 32 * 32, 128u * 128 * 128, 2448u * 2040}) {
        float *grid, *copies, *out;
        unsigned* hist;
        CK(hipMalloc(&grid, (size_t)cells * 4)); CK(hipMalloc(&copies, (size_t)cells * 4 * 8)); CK(hipMalloc(&out, (size_t)cells * 4));
        CK(hipMalloc(&hist, 16 * 64 * 4));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        auto run = [&](int mode, const char* name) {
            float best = 1e9;
            double total = 0;
            for (int rep = 0; rep < 5; ++rep) {
                CK(hipMemset(grid, 0, (size_t)cells * 4)); CK(hipMemset(copies, 0, (size_t)cells * 4 * 8)); CK(hipMemset(hist, 0, 16 * 64 * 4));
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0));
                if (mode == 0) hipLaunchKernelGGL(k_atomics<0>, dim3(2048), dim3(256), 0, 0, grid, cells, n, hist);
                if (mode == 1) hipLaunchKernelGGL(k_atomics<1>, dim3(2048), dim3(256), 0, 0, copies, cells, n, hist);
                if (mode == 2) hipLaunchKernelGGL(k_atomics<2>, dim3(2048), dim3(256), 0, 0, copies, cells, n, hist);
                if (mode != 0) hipLaunchKernelGGL(k_reduce, dim3(2048), dim3(256), 0, 0, copies, out, cells);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            std::vector<float> h(cells);
            CK(hipMemcpy(h.data(), mode == 0 ? grid : out, (size_t)cells * 4, hipMemcpyDeviceToHost));
            for (unsigned c = 0; c < cells; ++c) total += h[c];
            printf("cells=%9u %-34s %8.1f us  %6.1f G atomics/s  sum=%.0f (expect %lu) %s\n", cells, name, best * 1e3,
                   n / (best * 1e-3) / 1e9, total, n, total == (double)n ? "OK" : "MISMATCH");
        };
        run(0, "agent scope, one grid");
        run(2, "agent scope, per-XCD copies+reduce");
        run(1, "workgroup scope, per-XCD copies+red");
        std::vector<unsigned> hh(16 * 64);
        CK(hipMemcpy(hh.data(), hist, 16 * 64 * 4, hipMemcpyDeviceToHost));
        if (cells == 32u * 32 * 32) {
            printf("xcc_id -> blockIdx%%8 histogram:\n");
            for (int x = 0; x < 8; ++x) { printf("  xcc %d:", x); for (int j = 0; j < 8; ++j) printf(" %4u", hh[x * 64 + j]); printf("\n"); }
        }
        CK(hipFree(grid)); CK(hipFree(copies)); CK(hipFree(out)); CK(hipFree(hist));
    }
    return 0;
}

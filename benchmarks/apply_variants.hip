// apply_variants.hip — micro-benchmark harness for the affine-7 apply kernel (fp32), used to pick the
// production tiling. Build & run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off benchmarks/apply_variants.hip -o /tmp/av && /tmp/av
// Every variant computes y = R x for N particles (AoS rows of 7 floats) and is checked against variant 0.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ void apply7(const float* __restrict__ R, const float (&x)[7], float (&y)[7]) {
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        float acc = R[i * 7] * x[0];
#pragma unroll
        for (int j = 1; j < 7; ++j) acc = fmaf(R[i * 7 + j], x[j], acc);
        y[i] = acc;
    }
}

// ---- V_tile<PPT>: one tile per block through LDS (production structure) -------------------------
template <int PPT>
__global__ __launch_bounds__(256) void k_tile(const float* __restrict__ x, const float* __restrict__ R,
                                              float* __restrict__ y, long N) {
    constexpr int TP = PPT * 256;
    __shared__ __attribute__((aligned(16))) float lds[TP * 7];
    const long n0 = (long)blockIdx.x * TP;
    const int np = (int)((N - n0 < TP) ? (N - n0) : TP);
    const float4* gv = reinterpret_cast<const float4*>(x + n0 * 7);
    float4* lv = reinterpret_cast<float4*>(lds);
    const int nvec = np * 7 / 4;
    for (int v = threadIdx.x; v < nvec; v += 256) lv[v] = gv[v];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const int p = threadIdx.x + k * 256;
        if (p < np) {
            float a[7], b[7];
#pragma unroll
            for (int j = 0; j < 7; ++j) a[j] = lds[p * 7 + j];
            apply7(R, a, b);
#pragma unroll
            for (int j = 0; j < 7; ++j) lds[p * 7 + j] = b[j];
        }
    }
    __syncthreads();
    float4* ov = reinterpret_cast<float4*>(y + n0 * 7);
    for (int v = threadIdx.x; v < nvec; v += 256) ov[v] = lv[v];
}

// ---- V_pipe<PPT>: persistent blocks, register prefetch of the next tile ------------------------------
template <int PPT>
__global__ __launch_bounds__(256) void k_pipe(const float* __restrict__ x, const float* __restrict__ R,
                                              float* __restrict__ y, long N) {
    constexpr int TP = PPT * 256;
    constexpr int VPT = (TP * 7 / 4 + 255) / 256;  // float4 per thread per tile
    __shared__ __attribute__((aligned(16))) float lds[TP * 7];
    const long tiles = (N + TP - 1) / TP;
    float4 pre[VPT];
    long t = blockIdx.x;
    auto prefetch = [&](long tt) {
        const long n0 = tt * TP;
        const int np = (int)((N - n0 < TP) ? (N - n0) : TP);
        const int nvec = np * 7 / 4;
        const float4* gv = reinterpret_cast<const float4*>(x + n0 * 7);
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int v = threadIdx.x + i * 256;
            if (v < nvec) pre[i] = gv[v];
        }
    };
    if (t < tiles) prefetch(t);
    float4* lv = reinterpret_cast<float4*>(lds);
    for (; t < tiles; t += gridDim.x) {
        const long n0 = t * TP;
        const int np = (int)((N - n0 < TP) ? (N - n0) : TP);
        const int nvec = np * 7 / 4;
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int v = threadIdx.x + i * 256;
            if (v < nvec) lv[v] = pre[i];
        }
        __syncthreads();
        if (t + gridDim.x < tiles) prefetch(t + gridDim.x);
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int p = threadIdx.x + k * 256;
            if (p < np) {
                float a[7], b[7];
#pragma unroll
                for (int j = 0; j < 7; ++j) a[j] = lds[p * 7 + j];
                apply7(R, a, b);
#pragma unroll
                for (int j = 0; j < 7; ++j) lds[p * 7 + j] = b[j];
            }
        }
        __syncthreads();
        float4* ov = reinterpret_cast<float4*>(y + n0 * 7);
        for (int v = threadIdx.x; v < nvec; v += 256) ov[v] = lv[v];
        __syncthreads();
    }
}

// ---- V_direct: no LDS, every lane moves its own 28-byte row with dword accesses -----------------------
__global__ __launch_bounds__(256) void k_direct(const float* __restrict__ x, const float* __restrict__ R,
                                                float* __restrict__ y, long N) {
    const long p = (long)blockIdx.x * 256 + threadIdx.x;
    if (p >= N) return;
    float a[7], b[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) a[j] = x[p * 7 + j];
    apply7(R, a, b);
#pragma unroll
    for (int j = 0; j < 7; ++j) y[p * 7 + j] = b[j];
}

// ---- V_quad: a lane handles 4 consecutive particles = 7 float4 (112 contiguous bytes), no LDS ---------
__global__ __launch_bounds__(256) void k_quad(const float* __restrict__ x, const float* __restrict__ R,
                                              float* __restrict__ y, long N) {
    const long q = (long)blockIdx.x * 256 + threadIdx.x;  // group of 4 particles
    if (q * 4 + 3 >= N) {
        for (long p = q * 4; p < N; ++p) {
            float a[7], b[7];
            for (int j = 0; j < 7; ++j) a[j] = x[p * 7 + j];
            apply7(R, a, b);
            for (int j = 0; j < 7; ++j) y[p * 7 + j] = b[j];
        }
        return;
    }
    const float4* gv = reinterpret_cast<const float4*>(x + q * 28);
    float v[28];
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        const float4 t = gv[i];
        v[4 * i] = t.x; v[4 * i + 1] = t.y; v[4 * i + 2] = t.z; v[4 * i + 3] = t.w;
    }
    float o[28];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float a[7], b[7];
#pragma unroll
        for (int j = 0; j < 7; ++j) a[j] = v[k * 7 + j];
        apply7(R, a, b);
#pragma unroll
        for (int j = 0; j < 7; ++j) o[k * 7 + j] = b[j];
    }
    float4* ov = reinterpret_cast<float4*>(y + q * 28);
#pragma unroll
    for (int i = 0; i < 7; ++i) ov[i] = make_float4(o[4 * i], o[4 * i + 1], o[4 * i + 2], o[4 * i + 3]);
}

typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 nt_load(const float4* p) {
    const v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void nt_store(float4 v, float4* p) {
    v4f w = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(w, reinterpret_cast<v4f*>(p));
}

// ---- V_wave<PPL>: one WAVE per block (64 lanes), tile = 64 * PPL particles: the barrier only ever waits for the
// wave's own loads, so waves of a CU drift apart and keep loads, FMAs and stores of different tiles in flight together
template <int PPL, bool NT>
__global__ __launch_bounds__(64) void k_wave(const float* __restrict__ x, const float* __restrict__ R,
                                             float* __restrict__ y, long N) {
    constexpr int TP = PPL * 64;
    __shared__ __attribute__((aligned(16))) float lds[TP * 7];
    const long n0 = (long)blockIdx.x * TP;
    const int np = (int)((N - n0 < TP) ? (N - n0) : TP);
    const float4* gv = reinterpret_cast<const float4*>(x + n0 * 7);
    float4* lv = reinterpret_cast<float4*>(lds);
    const int nvec = np * 7 / 4;
    for (int v = threadIdx.x; v < nvec; v += 64) lv[v] = NT ? nt_load(gv + v) : gv[v];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PPL; ++k) {
        const int p = threadIdx.x + k * 64;
        if (p < np) {
            float a[7], b[7];
#pragma unroll
            for (int j = 0; j < 7; ++j) a[j] = lds[p * 7 + j];
            apply7(R, a, b);
#pragma unroll
            for (int j = 0; j < 7; ++j) lds[p * 7 + j] = b[j];
        }
    }
    __syncthreads();
    float4* ov = reinterpret_cast<float4*>(y + n0 * 7);
    for (int v = threadIdx.x; v < nvec; v += 64) {
        if (NT) nt_store(lv[v], ov + v);
        else ov[v] = lv[v];
    }
}

// production tile with non-temporal loads / stores
template <int PPT, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void k_tile_nt(const float* __restrict__ x, const float* __restrict__ R,
                                                 float* __restrict__ y, long N) {
    constexpr int TP = PPT * 256;
    __shared__ __attribute__((aligned(16))) float lds[TP * 7];
    const long n0 = (long)blockIdx.x * TP;
    const int np = (int)((N - n0 < TP) ? (N - n0) : TP);
    const float4* gv = reinterpret_cast<const float4*>(x + n0 * 7);
    float4* lv = reinterpret_cast<float4*>(lds);
    const int nvec = np * 7 / 4;
    for (int v = threadIdx.x; v < nvec; v += 256) lv[v] = NTL ? nt_load(gv + v) : gv[v];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const int p = threadIdx.x + k * 256;
        if (p < np) {
            float a[7], b[7];
#pragma unroll
            for (int j = 0; j < 7; ++j) a[j] = lds[p * 7 + j];
            apply7(R, a, b);
#pragma unroll
            for (int j = 0; j < 7; ++j) lds[p * 7 + j] = b[j];
        }
    }
    __syncthreads();
    float4* ov = reinterpret_cast<float4*>(y + n0 * 7);
    for (int v = threadIdx.x; v < nvec; v += 256) {
        if (NTS) nt_store(lv[v], ov + v);
        else ov[v] = lv[v];
    }
}

__global__ __launch_bounds__(256) void k_copy_nt(const float4* __restrict__ x, float4* __restrict__ y, long nvec) {
    for (long v = (long)blockIdx.x * 256 + threadIdx.x; v < nvec; v += (long)gridDim.x * 256) nt_store(nt_load(x + v), y + v);
}

// plain float4 copy of the same bytes: the practical ceiling
__global__ __launch_bounds__(256) void k_copy(const float4* __restrict__ x, float4* __restrict__ y, long nvec) {
    for (long v = (long)blockIdx.x * 256 + threadIdx.x; v < nvec; v += (long)gridDim.x * 256) y[v] = x[v];
}

template <typename F>
float time_it(F launch, int iters) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) launch();
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / iters;
}

int main(int argc, char** argv) {
    for (long N : {1000000L, 16000000L}) {
        std::vector<float> hx(N * 7), hR(49, 0.f);
        for (long i = 0; i < N * 7; ++i) hx[i] = (float)((i * 2654435761u) % 1000) * 1e-3f - 0.5f;
        for (int i = 0; i < 7; ++i) for (int j = 0; j < 7; ++j) hR[i * 7 + j] = (i == j) ? 1.f : 0.01f * (i + j);
        float *x, *y, *y0, *R;
        CK(hipMalloc(&x, N * 28)); CK(hipMalloc(&y, N * 28)); CK(hipMalloc(&y0, N * 28)); CK(hipMalloc(&R, 196));
        CK(hipMemcpy(x, hx.data(), N * 28, hipMemcpyHostToDevice));
        CK(hipMemcpy(R, hR.data(), 196, hipMemcpyHostToDevice));
        const int iters = N > 2000000 ? 50 : 300;
        auto report = [&](const char* name, float ms, bool check) {
            int bad = 0;
            if (check) {
                std::vector<float> a(7000), b(7000);
                CK(hipMemcpy(a.data(), y0 + (N - 1000) * 7, 28000, hipMemcpyDeviceToHost));
                CK(hipMemcpy(b.data(), y + (N - 1000) * 7, 28000, hipMemcpyDeviceToHost));
                for (int i = 0; i < 7000; ++i) if (a[i] != b[i]) ++bad;
            }
            printf("N=%ld %-28s %8.3f us  %7.1f GB/s  mismatches=%d\n", N, name, ms * 1e3, 56.0 * N / (ms * 1e-3) / 1e9, bad);
        };
        float ms;
        ms = time_it([&] { hipLaunchKernelGGL(k_tile<2>, dim3((N + 511) / 512), dim3(256), 0, 0, x, R, y0, N); }, iters);
        report("tile PPT=2 (production)", ms, false);
        ms = time_it([&] { hipLaunchKernelGGL(k_tile<1>, dim3((N + 255) / 256), dim3(256), 0, 0, x, R, y, N); }, iters);
        report("tile PPT=1", ms, true);
        ms = time_it([&] { hipLaunchKernelGGL(k_tile<4>, dim3((N + 1023) / 1024), dim3(256), 0, 0, x, R, y, N); }, iters);
        report("tile PPT=4", ms, true);
        for (int g : {256, 512, 1024, 2048}) {
            char name[64];
            snprintf(name, 64, "pipe PPT=2 grid=%d", g);
            ms = time_it([&] { hipLaunchKernelGGL(k_pipe<2>, dim3(g), dim3(256), 0, 0, x, R, y, N); }, iters);
            report(name, ms, true);
            snprintf(name, 64, "pipe PPT=4 grid=%d", g);
            ms = time_it([&] { hipLaunchKernelGGL(k_pipe<4>, dim3(g), dim3(256), 0, 0, x, R, y, N); }, iters);
            report(name, ms, true);
        }
        ms = time_it([&] { hipLaunchKernelGGL((k_wave<2, false>), dim3((N + 127) / 128), dim3(64), 0, 0, x, R, y, N); }, iters);
        report("wave tile PPL=2", ms, true);
        ms = time_it([&] { hipLaunchKernelGGL((k_wave<4, false>), dim3((N + 255) / 256), dim3(64), 0, 0, x, R, y, N); }, iters);
        report("wave tile PPL=4", ms, true);
        ms = time_it([&] { hipLaunchKernelGGL((k_wave<8, false>), dim3((N + 511) / 512), dim3(64), 0, 0, x, R, y, N); }, iters);
        report("wave tile PPL=8", ms, true);
        ms = time_it([&] { hipLaunchKernelGGL((k_wave<4, true>), dim3((N + 255) / 256), dim3(64), 0, 0, x, R, y, N); }, iters);
        report("wave tile PPL=4 nontemporal", ms, true);
        ms = time_it([&] { hipLaunchKernelGGL((k_tile_nt<2, true, true>), dim3((N + 511) / 512), dim3(256), 0, 0, x, R, y, N); }, iters);
        report("tile PPT=2 nt load+store", ms, true);
        ms = time_it([&] { hipLaunchKernelGGL((k_tile_nt<2, true, false>), dim3((N + 511) / 512), dim3(256), 0, 0, x, R, y, N); }, iters);
        report("tile PPT=2 nt load", ms, true);
        ms = time_it([&] { hipLaunchKernelGGL((k_tile_nt<2, false, true>), dim3((N + 511) / 512), dim3(256), 0, 0, x, R, y, N); }, iters);
        report("tile PPT=2 nt store", ms, true);
        ms = time_it([&] { hipLaunchKernelGGL((k_tile_nt<1, true, true>), dim3((N + 255) / 256), dim3(256), 0, 0, x, R, y, N); }, iters);
        report("tile PPT=1 nt load+store", ms, true);
        ms = time_it([&] { hipLaunchKernelGGL((k_wave<2, true>), dim3((N + 127) / 128), dim3(64), 0, 0, x, R, y, N); }, iters);
        report("wave tile PPL=2 nontemporal", ms, true);
        {   // the way a lattice is tracked: the output of one pass is the input of the next (two buffers)
            float* a = x; float* b = y;
            auto pp = [&](auto kernel, dim3 grid, dim3 block) {
                return time_it([&] { hipLaunchKernelGGL(kernel, grid, block, 0, 0, a, R, b, N); std::swap(a, b); }, iters % 2 ? iters + 1 : iters);
            };
            CK(hipMemcpy(y, x, N * 28, hipMemcpyDeviceToDevice));
            ms = pp(k_tile<2>, dim3((N + 511) / 512), dim3(256)); report("PING-PONG tile PPT=2", ms, false);
            ms = pp(k_tile_nt<2, true, true>, dim3((N + 511) / 512), dim3(256)); report("PING-PONG tile nt load+store", ms, false);
            ms = pp(k_tile_nt<2, true, false>, dim3((N + 511) / 512), dim3(256)); report("PING-PONG tile nt load", ms, false);
            ms = pp(k_tile_nt<2, false, true>, dim3((N + 511) / 512), dim3(256)); report("PING-PONG tile nt store", ms, false);
            ms = pp(k_wave<2, false>, dim3((N + 127) / 128), dim3(64)); report("PING-PONG wave PPL=2", ms, false);
            ms = pp(k_wave<2, true>, dim3((N + 127) / 128), dim3(64)); report("PING-PONG wave PPL=2 nt", ms, false);
            CK(hipMemcpy(x, hx.data(), N * 28, hipMemcpyHostToDevice));
        }
        ms = time_it([&] { hipLaunchKernelGGL(k_direct, dim3((N + 255) / 256), dim3(256), 0, 0, x, R, y, N); }, iters);
        report("direct dword", ms, true);
        ms = time_it([&] { hipLaunchKernelGGL(k_quad, dim3((N / 4 + 256) / 256), dim3(256), 0, 0, x, R, y, N); }, iters);
        report("quad (4 rows/lane, float4)", ms, true);
        for (int g : {1024, 2048, 4096, 8192}) {
            char name[64];
            snprintf(name, 64, "float4 copy NT grid=%d", g);
            ms = time_it([&] { hipLaunchKernelGGL(k_copy_nt, dim3(g), dim3(256), 0, 0, (const float4*)x, (float4*)y, N * 7 / 4); }, iters);
            report(name, ms, false);
        }
        ms = time_it([&] { hipLaunchKernelGGL((k_tile_nt<4, true, true>), dim3((N + 1023) / 1024), dim3(256), 0, 0, x, R, y, N); }, iters);
        report("tile PPT=4 nt load+store", ms, true);
        for (int g : {1024, 2048, 4096, 8192}) {
            char name[64];
            snprintf(name, 64, "float4 copy grid=%d", g);
            ms = time_it([&] { hipLaunchKernelGGL(k_copy, dim3(g), dim3(256), 0, 0, (const float4*)x, (float4*)y, N * 7 / 4); }, iters);
            report(name, ms, false);
        }
        CK(hipFree(x)); CK(hipFree(y)); CK(hipFree(y0)); CK(hipFree(R));
    }
    return 0;
}

// mfma_f64_probe.hip — does the fp64 matrix pipe of MI355X (gfx950) beat its fp64 vector pipe? Question behind it: the
// per-setting covariance accumulation of chx_track_moments (sum_n w y y^T, a (7 x N)(N x 7) product per batch row) is
// fp64-VALU bound; would v_mfma_f64_16x16x4_f64 pay?
//   build: hipcc --offload-arch=gfx950 -O3 -o mfma_f64_probe benchmarks/mfma_f64_probe.hip
// Prints sustained TFLOP/s of (a) v_mfma_f64_16x16x4_f64 (4 independent accumulator tiles per wave), (b) v_fma_f64
// (8 independent chains per lane), (c) v_pk_fma_f32 (8 independent packed chains per lane).
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>

typedef double v4d __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void k_mfma(double* out, int iters) {
    v4d acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
    }
    double s = 0;
    for (int t = 0; t < 4; ++t) s += acc[t].x + acc[t].y + acc[t].z + acc[t].w;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_fma64(double* out, int iters) {
    double acc[8];
    for (int t = 0; t < 8; ++t) acc[t] = t;
    const double a = 1.0 + threadIdx.x * 1e-12, b = 1e-9;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int t = 0; t < 8; ++t) acc[t] = fma(acc[t], a, b);
    }
    double s = 0;
    for (int t = 0; t < 8; ++t) s += acc[t];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_pk32(float* out, int iters) {
    v2f acc[8];
    for (int t = 0; t < 8; ++t) acc[t] = v2f{(float)t, (float)t + 0.5f};
    const v2f a = {1.0f + threadIdx.x * 1e-7f, 1.0f - threadIdx.x * 1e-7f}, b = {1e-6f, 2e-6f};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int t = 0; t < 8; ++t) acc[t] = __builtin_elementwise_fma(acc[t], a, b);
    }
    float s = 0;
    for (int t = 0; t < 8; ++t) s += acc[t].x + acc[t].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <typename F>
double time_ms(F launch) {
    launch();
    hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < 5; ++r) launch();
    hipDeviceSynchronize();
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / 5;
}

int main() {
    const int blocks = 256 * 16, iters = 4096;
    double* d;
    hipMalloc(&d, sizeof(double) * blocks * 256);
    const double waves = blocks * 4.0;
    double ms = time_ms([&] { hipLaunchKernelGGL(k_mfma, dim3(blocks), dim3(256), 0, 0, d, iters); });
    printf("v_mfma_f64_16x16x4_f64 : %8.1f TFLOP/s\n", waves * iters * 4.0 * (16.0 * 16 * 4 * 2) / (ms * 1e-3) / 1e12);
    ms = time_ms([&] { hipLaunchKernelGGL(k_fma64, dim3(blocks), dim3(256), 0, 0, d, iters); });
    printf("v_fma_f64              : %8.1f TFLOP/s\n", blocks * 256.0 * iters * 8.0 * 2 / (ms * 1e-3) / 1e12);
    ms = time_ms([&] { hipLaunchKernelGGL(k_pk32, dim3(blocks), dim3(256), 0, 0, (float*)d, iters); });
    printf("v_pk_fma_f32           : %8.1f TFLOP/s\n", blocks * 256.0 * iters * 8.0 * 4 / (ms * 1e-3) / 1e12);
    return 0;
}

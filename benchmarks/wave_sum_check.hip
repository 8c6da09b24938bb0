// chx_wave_sum (all lanes) against chx_wave_sum_lane63 (DPP row broadcasts, lane 63 only): the same bits on drawn doubles.
// hipcc --offload-arch=gfx950 -O2 -I cheetah_amd/csrc benchmarks/wave_sum_check.hip -o /tmp/wave_sum_check && /tmp/wave_sum_check
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "chx_common.h"

__global__ void both(const double* in, double* a, double* b) {
    const double v = in[blockIdx.x * 64 + threadIdx.x];
    const double s0 = chx_wave_sum(v);
    const double s1 = chx_wave_sum_lane63(v);
    if (threadIdx.x == 63) { a[blockIdx.x] = s0; b[blockIdx.x] = s1; }
}

int main() {
    const int W = 4096;
    double* h = (double*)malloc(W * 64 * sizeof(double));
    srand(7);
    for (int i = 0; i < W * 64; ++i) h[i] = ((double)rand() / RAND_MAX - 0.5) * ((i % 7) ? 1e-3 : 3.0) + 1e-9 * (i % 13);
    double *d, *a, *b;
    hipMalloc(&d, W * 64 * 8); hipMalloc(&a, W * 8); hipMalloc(&b, W * 8);
    hipMemcpy(d, h, W * 64 * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(both, dim3(W), dim3(64), 0, 0, d, a, b);
    double* ha = (double*)malloc(W * 8); double* hb = (double*)malloc(W * 8);
    hipMemcpy(ha, a, W * 8, hipMemcpyDeviceToHost); hipMemcpy(hb, b, W * 8, hipMemcpyDeviceToHost);
    int bad = 0; double worst = 0;
    for (int w = 0; w < W; ++w) {
        if (memcmp(&ha[w], &hb[w], 8) != 0) ++bad;
        double ref = 0; for (int l = 0; l < 64; ++l) ref += h[w * 64 + l];
        double e = ha[w] - ref; if (e < 0) e = -e; if (e > worst) worst = e;
    }
    printf("waves %d, differing bits in %d, worst |wave sum - serial sum| %.3e\n", W, bad, worst);
    return bad != 0;
}

// Times chx_apply_affine7 / chx_track_elementwise through the C-ABI with plain hipMalloc buffers: the same kernel
// launched on one (input, output) pair over and over, and ping-ponged between two buffers like a tracked lattice.
//   hipcc --offload-arch=gfx950 -O2 -Iinclude benchmarks/apply_abi_pingpong.hip -Lcheetah_amd -lchx -Wl,-rpath,$PWD/cheetah_amd -o /tmp/pp && /tmp/pp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "chx.h"

int main() {
    const long N = 1000000;
    float *x, *a, *b, *R;
    hipMalloc(&x, N * 28); hipMalloc(&a, N * 28); hipMalloc(&b, N * 28); hipMalloc(&R, 1000 * 196);
    std::vector<float> hx(N * 7, 0.5f), hR(1000 * 49, 0.f);
    for (int e = 0; e < 1000; ++e) for (int i = 0; i < 7; ++i) hR[e * 49 + i * 8] = 1.f;
    hipMemcpy(x, hx.data(), N * 28, hipMemcpyHostToDevice);
    hipMemcpy(a, hx.data(), N * 28, hipMemcpyHostToDevice);
    hipMemcpy(R, hR.data(), 1000 * 196, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto timeit = [&](const char* name, auto fn, int n) {
        fn(); hipDeviceSynchronize();
        hipEventRecord(e0, 0); fn(); hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-44s %7.3f us per kernel\n", name, ms * 1e3 / n);
    };
    timeit("repeated x -> a", [&] { for (int i = 0; i < 1000; ++i) chx_apply_affine7(x, R, a, 1, 1, 1, N, CHX_F32, 0); }, 1000);
    timeit("ping-pong a <-> b (apply loop)", [&] { float *p = a, *q = b; for (int i = 0; i < 1000; ++i) { chx_apply_affine7(p, R, q, 1, 1, 1, N, CHX_F32, 0); std::swap(p, q); } }, 1000);
    timeit("chx_track_elementwise E=1000", [&] { chx_track_elementwise(x, R, a, b, 1000, 1, 1, 1, N, CHX_F32, 0); }, 1000);
    timeit("in place a -> a", [&] { for (int i = 0; i < 1000; ++i) chx_apply_affine7(a, R, a, 1, 1, 1, N, CHX_F32, 0); }, 1000);
    return 0;
}

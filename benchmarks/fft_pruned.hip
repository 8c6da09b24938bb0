// Micro-benchmark: Hockney convolution FFTs at (2g)^3 with hipFFT, full 3-D plans vs "x-pruned" plans
// (2-D R2C on the g non-zero x-planes + strided 1-D C2C along x), fp32, in place.
//   hipcc --offload-arch=gfx950 -O3 benchmarks/fft_pruned.hip -o build/fft_pruned -lhipfft && build/fft_pruned 128
#include <hip/hip_runtime.h>
#include <hipfft/hipfft.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                             \
    do {                                                                  \
        auto e__ = (x);                                                   \
        if (e__ != 0) {                                                   \
            printf("error %d at %s:%d\n", (int)e__, __FILE__, __LINE__); \
            exit(1);                                                      \
        }                                                                 \
    } while (0)

template <typename F>
static float timeit(F f, hipStream_t s, int iters = 20) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) f();
    CK(hipEventRecord(a, s));
    for (int i = 0; i < iters; ++i) f();
    CK(hipEventRecord(b, s));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms / iters * 1e3f;
}

int main(int argc, char** argv) {
    const int g = argc > 1 ? atoi(argv[1]) : 128;
    const int n = 2 * g, nc = g + 1;
    hipStream_t s;
    CK(hipStreamCreate(&s));
    const size_t spec_elems = (size_t)n * n * nc;  // complex
    float2* spec;
    float* real;
    CK(hipMalloc(&spec, spec_elems * sizeof(float2)));
    CK(hipMalloc(&real, (size_t)n * n * n * sizeof(float)));
    CK(hipMemset(spec, 0, spec_elems * sizeof(float2)));
    CK(hipMemset(real, 0, (size_t)n * n * n * sizeof(float)));

    // (a) full 3-D, out of place (what torch.fft.rfftn / irfftn run)
    hipfftHandle p3f, p3b;
    CK(hipfftPlan3d(&p3f, n, n, n, HIPFFT_R2C));
    CK(hipfftPlan3d(&p3b, n, n, n, HIPFFT_C2R));
    CK(hipfftSetStream(p3f, s));
    CK(hipfftSetStream(p3b, s));
    printf("g = %d\n", g);
    printf("3-D R2C out-of-place      %8.1f us\n", timeit([&] { CK(hipfftExecR2C(p3f, real, (hipfftComplex*)spec)); }, s));
    printf("3-D C2R out-of-place      %8.1f us\n", timeit([&] { CK(hipfftExecC2R(p3b, (hipfftComplex*)spec, real)); }, s));

    // (a') full 3-D in place (padded last dim)
    hipfftHandle p3fi, p3bi;
    CK(hipfftPlan3d(&p3fi, n, n, n, HIPFFT_R2C));
    CK(hipfftPlan3d(&p3bi, n, n, n, HIPFFT_C2R));
    CK(hipfftSetStream(p3fi, s));
    CK(hipfftSetStream(p3bi, s));
    printf("3-D R2C in-place          %8.1f us\n", timeit([&] { CK(hipfftExecR2C(p3fi, (float*)spec, (hipfftComplex*)spec)); }, s));
    printf("3-D C2R in-place          %8.1f us\n", timeit([&] { CK(hipfftExecC2R(p3bi, (hipfftComplex*)spec, (float*)spec)); }, s));

    // (b) x-pruned: 2-D R2C (y,z) on the first `planes` x-planes, in place in the padded spectrum array,
    //     then 1-D C2C along x for all (ky,kz): stride n*nc, dist 1, batch n*nc
    for (int planes : {g, g + 1}) {
        hipfftHandle p2f, p2b, p1;
        int n2[2] = {n, n};
        int inembed[2] = {n, 2 * nc}, onembed[2] = {n, nc};
        CK(hipfftPlanMany(&p2f, 2, n2, inembed, 1, n * 2 * nc, onembed, 1, n * nc, HIPFFT_R2C, planes));
        CK(hipfftPlanMany(&p2b, 2, n2, onembed, 1, n * nc, inembed, 1, n * 2 * nc, HIPFFT_C2R, planes));
        int n1[1] = {n};
        int embed1[1] = {n};
        CK(hipfftPlanMany(&p1, 1, n1, embed1, n * nc, 1, embed1, n * nc, 1, HIPFFT_C2C, n * nc));
        CK(hipfftSetStream(p2f, s));
        CK(hipfftSetStream(p2b, s));
        CK(hipfftSetStream(p1, s));
        const float t2f = timeit([&] { CK(hipfftExecR2C(p2f, (float*)spec, (hipfftComplex*)spec)); }, s);
        const float t2b = timeit([&] { CK(hipfftExecC2R(p2b, (hipfftComplex*)spec, (float*)spec)); }, s);
        const float t1f = timeit([&] { CK(hipfftExecC2C(p1, (hipfftComplex*)spec, (hipfftComplex*)spec, HIPFFT_FORWARD)); }, s);
        const float t1b = timeit([&] { CK(hipfftExecC2C(p1, (hipfftComplex*)spec, (hipfftComplex*)spec, HIPFFT_BACKWARD)); }, s);
        printf("planes=%3d: 2-D R2C %7.1f us, 2-D C2R %7.1f us, 1-D x fwd %7.1f us, 1-D x bwd %7.1f us  -> fwd %7.1f, bwd %7.1f\n",
               planes, t2f, t2b, t1f, t1b, t2f + t1f, t2b + t1b);
        hipfftDestroy(p2f);
        hipfftDestroy(p2b);
        hipfftDestroy(p1);
    }
    // memset of the spectrum array (needed once per kick for the pruned layout)
    printf("memset spectrum           %8.1f us\n", timeit([&] { CK(hipMemsetAsync(spec, 0, spec_elems * sizeof(float2), s)); }, s));
    return 0;
}

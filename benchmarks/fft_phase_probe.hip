// fft_phase_probe.hip — where does a strided line-FFT pass of the 128^3 Poisson solve spend its 20 us?
// The y-forward pass (A[gx][gy][g+1] complex, 128 valid points -> Bf[gx][2gy][g+1], 256 points; 17 MB in, 34 MB out) as
//   full      the product kernel's structure (loads -> 16-point FFTs -> LDS exchange -> 16-point FFTs -> stores)
//   memory    the same loads and stores, no butterflies, no LDS
//   compute   the same butterflies and LDS exchange, no global traffic
// If memory + compute ~ full, the phases do not overlap (all workgroups of the single resident round run in lock step).
#include "../cheetah_amd/csrc/chx_fft.hip"

#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

namespace {
template <int MODE>
__global__ __launch_bounds__(CHX_BLOCK) void probe_kernel(const float* __restrict__ in, float* __restrict__ out, int n_valid,
                                                         int64_t L, int64_t inner_count, LineLayout li, LineLayout lo,
                                                         int inner_live) {
    using T = float;
    constexpr int M = 16, n = 256;
    constexpr int LP = kTL + 1, KP = M * LP + 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cplx<T>* xch = reinterpret_cast<cplx<T>*>(smem_raw);
    cplx<T>* tw = xch + 16 * KP;
    __shared__ int64_t in_base[kTL], out_base[kTL];
    __shared__ int dead[kTL];
    const int tid = threadIdx.x;
    const int64_t l0 = (int64_t)blockIdx.x * kTL;
    const int nl = (int)((L - l0 < kTL) ? (L - l0) : kTL);
    if (tid < kTL) {
        const int64_t l = l0 + (tid < nl ? tid : 0);
        const int64_t outer = l / inner_count, inner = l - outer * inner_count;
        in_base[tid] = outer * li.outer_stride + inner * li.inner_stride;
        out_base[tid] = outer * lo.outer_stride + inner * lo.inner_stride;
        dead[tid] = inner >= inner_live;
    }
    for (int k = tid; k < n; k += CHX_BLOCK) { T s, c; sincos_2pi<T>(k, n, s, c); tw[k].re = c; tw[k].im = -s; }
    __syncthreads();
    const int line = tid & 15, c = tid >> 4;
    cplx<T> x[16];
    const bool live = line < nl && !dead[line];
    const int64_t base = in_base[line];
#pragma unroll
    for (int j1 = 0; j1 < 16; ++j1) {
        const int p = c + M * j1;
        x[j1].re = (T)(MODE == 2 ? tid * 1e-3f + j1 : 0);
        x[j1].im = (T)0;
        if (MODE != 2 && live && p < n_valid) x[j1] = reinterpret_cast<const cplx<T>*>(in)[base + (int64_t)p * li.point_stride];
    }
    cplx<T> y[M];
    if (MODE != 1) {
        fft_reg<T, 16>(x, 0);
#pragma unroll
        for (int k1 = 0; k1 < 16; ++k1) {
            const cplx<T> w = tw[(c * k1) & (n - 1)];
            cplx<T> v;
            v.re = x[k1].re * w.re - x[k1].im * w.im;
            v.im = x[k1].re * w.im + x[k1].im * w.re;
            xch[k1 * KP + c * LP + line] = v;
        }
        __syncthreads();
#pragma unroll
        for (int j2 = 0; j2 < M; ++j2) y[j2] = xch[c * KP + j2 * LP + line];
        fft_reg<T, M>(y, 0);
    } else {
#pragma unroll
        for (int j2 = 0; j2 < M; ++j2) y[j2] = x[j2];
    }
    if (MODE == 2) {
        T acc = 0;
#pragma unroll
        for (int k2 = 0; k2 < M; ++k2) acc += y[k2].re + y[k2].im;
        if (acc == (T)123456.789f) out[tid] = acc;
        return;
    }
    if (live) {
        const int64_t ob = out_base[line];
#pragma unroll
        for (int k2 = 0; k2 < M; ++k2) {
            const int p = c + 16 * k2;
            reinterpret_cast<cplx<T>*>(out)[ob + (int64_t)p * lo.point_stride] = y[k2];
        }
    }
}
}  // namespace

int main() {
    const int g = 128, nzc = g + 1, ny = 2 * g;
    const int64_t nA = (int64_t)g * g * nzc, nB = (int64_t)g * ny * nzc;
    float *A, *Bf;
    CK(hipMalloc(&A, nA * 8)); CK(hipMalloc(&Bf, nB * 8));
    CK(hipMemset(A, 0, nA * 8));
    LineLayout ay{nzc, 1, (int64_t)g * nzc, nA}, by{nzc, 1, (int64_t)ny * nzc, nB};
    const int64_t L = (int64_t)g * nzc;
    const dim3 grid((unsigned)((L + kTL - 1) / kTL));
    const size_t shmem = ((size_t)16 * (16 * (kTL + 1) + 1) + 256) * sizeof(cplx<float>);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // padded pitch: g + 1 = 129 half-spectrum planes stored with a pitch of 144 (a multiple of 16 complex = 128 bytes), lines
    // enumerated over the padded pitch so that every tile of 16 lines starts on a 128-byte boundary; the 15 pad lines are dead
    const int pz = 144;
    const int64_t nAp = (int64_t)g * g * pz, nBp = (int64_t)g * ny * pz;
    float *Ap, *Bp;
    CK(hipMalloc(&Ap, nAp * 8)); CK(hipMalloc(&Bp, nBp * 8));
    CK(hipMemset(Ap, 0, nAp * 8));
    LineLayout ayp{pz, 1, (int64_t)g * pz, nAp}, byp{pz, 1, (int64_t)ny * pz, nBp};
    const int64_t Lp = (int64_t)g * pz;
    const dim3 gridp((unsigned)((Lp + kTL - 1) / kTL));
    const char* names[5] = {"full", "memory only", "compute only", "full, pitch 144 aligned", "memory only, pitch 144 aligned"};
    for (int mode = 0; mode < 5; ++mode) {
        auto launch = [&] {
            if (mode == 0) probe_kernel<0><<<grid, CHX_BLOCK, shmem>>>(A, Bf, g, L, nzc, ay, by, nzc);
            else if (mode == 1) probe_kernel<1><<<grid, CHX_BLOCK, shmem>>>(A, Bf, g, L, nzc, ay, by, nzc);
            else if (mode == 2) probe_kernel<2><<<grid, CHX_BLOCK, shmem>>>(A, Bf, g, L, nzc, ay, by, nzc);
            else if (mode == 3) probe_kernel<0><<<gridp, CHX_BLOCK, shmem>>>(Ap, Bp, g, Lp, pz, ayp, byp, nzc);
            else probe_kernel<1><<<gridp, CHX_BLOCK, shmem>>>(Ap, Bp, g, Lp, pz, ayp, byp, nzc);
        };
        launch(); launch(); CK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int it = 0; it < 20; ++it) {
            CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best;
        }
        CK(hipGetLastError());
        printf("%-32s %7.1f us  (%d tiles of %d lines, %.0f MB in + %.0f MB out useful -> %.2f TB/s)\n", names[mode], best * 1e3,
               (int)(mode >= 3 ? gridp.x : grid.x), kTL, nA * 8 / 1e6, nB * 8 / 1e6, (nA + nB) * 8 / (best * 1e-3) / 1e12);
    }
    return 0;
}

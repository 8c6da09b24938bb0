// fft_phase_probe.hip — where does a strided line-FFT pass of the 128^3 Poisson solve spend its time?
// The y-forward pass (A[g+1][g][g] complex, 128 valid points -> Bf[g+1][2g][g], 256 points; 17 MB in, 34 MB out) as
//   full      the product kernel's structure (loads -> 16-point FFTs -> LDS exchange -> 16-point FFTs -> stores)
//   memory    the same loads and stores, no butterflies, no LDS
//   compute   the same butterflies and LDS exchange, no global traffic
// If memory + compute ~ full, the phases do not overlap (all workgroups of the single resident round run in lock step).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude -o build/fft_phase_probe benchmarks/fft_phase_probe.hip \
//        -Lcheetah_amd -l:libchx.so -Wl,-rpath,$PWD/cheetah_amd     (chx_fft.hip calls into chx_spacecharge.hip)
#include "../cheetah_amd/csrc/chx_fft.hip"

#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

namespace {
template <int MODE>
__global__ __launch_bounds__(CHX_BLOCK) void probe_kernel(const float* __restrict__ in, float* __restrict__ out, int64_t L,
                                                         int64_t inner_count, LineLayout li, LineLayout lo) {
    using T = float;
    constexpr int M = 16;
    using RT = RegTile<T, M>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    vec2<T>* xch = reinterpret_cast<vec2<T>*>(smem_raw);
    vec2<T>* tw = xch + 16 * RT::KP;
    __shared__ int64_t in_base[kTL], out_base[kTL];
    const int tid = threadIdx.x;
    const int64_t l0 = (int64_t)blockIdx.x * kTL;
    const int nl = (int)((L - l0 < kTL) ? (L - l0) : kTL);
    if (tid < kTL) {
        const int64_t l = l0 + (tid < nl ? tid : 0);
        const int64_t outer = l / inner_count, inner = l - outer * inner_count;
        in_base[tid] = outer * li.outer_stride + inner * li.inner_stride;
        out_base[tid] = outer * lo.outer_stride + inner * lo.inner_stride;
    }
    fill_twiddles<T, M>(tw, false);
    __syncthreads();
    const int line = tid & 15, c = tid >> 4;
    vec2<T> x[16];
    const vec2<T>* q = reinterpret_cast<const vec2<T>*>(in) + in_base[line];
#pragma unroll
    for (int j1 = 0; j1 < 16; ++j1) {
        x[j1] = vec2<T>{(T)(MODE == 2 ? tid * 1e-3f + j1 : 0), (T)0};
        if (MODE != 2 && j1 < 8) x[j1] = q[(int64_t)(c + M * j1) * li.point_stride];
    }
    vec2<T> y[M];
    if (MODE != 1) {
        pass1_to_lds<T, M, false, true>(x, xch, tw, c, line);
        __syncthreads();
#pragma unroll
        for (int j2 = 0; j2 < M; ++j2) y[j2] = xch[c * RT::KP + j2 * RT::LP + line];
        fft_small<T, M, false>(y);
    } else {
#pragma unroll
        for (int j2 = 0; j2 < M; ++j2) y[j2] = x[j2 & 7];
    }
    if (MODE == 2) {
        T acc = 0;
#pragma unroll
        for (int k2 = 0; k2 < M; ++k2) acc += y[k2].x + y[k2].y;
        if (acc == (T)123456.789f) out[tid] = acc;
        return;
    }
    if (line < nl) {
        vec2<T>* o = reinterpret_cast<vec2<T>*>(out) + out_base[line];
#pragma unroll
        for (int k2 = 0; k2 < M; ++k2) o[(int64_t)(c + 16 * k2) * lo.point_stride] = y[k2];
    }
}
}  // namespace

int main() {
    const int g = 128, nxc = g + 1, ny = 2 * g;
    const int64_t nA = (int64_t)nxc * g * g, nB = (int64_t)nxc * ny * g;
    float *A, *Bf;
    CK(hipMalloc(&A, nA * 8)); CK(hipMalloc(&Bf, nB * 8));
    CK(hipMemset(A, 0, nA * 8));
    LineLayout ay{g, 1, (int64_t)g * g, nA}, by{g, 1, (int64_t)ny * g, nB};   // lines (kx, z), point stride g
    const int64_t L = (int64_t)nxc * g;
    const dim3 grid((unsigned)((L + kTL - 1) / kTL));
    const size_t shmem = RegTile<float, 16>::shmem;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char* names[3] = {"full", "memory only", "compute only"};
    for (int mode = 0; mode < 3; ++mode) {
        auto launch = [&] {
            if (mode == 0) probe_kernel<0><<<grid, CHX_BLOCK, shmem>>>(A, Bf, L, g, ay, by);
            else if (mode == 1) probe_kernel<1><<<grid, CHX_BLOCK, shmem>>>(A, Bf, L, g, ay, by);
            else probe_kernel<2><<<grid, CHX_BLOCK, shmem>>>(A, Bf, L, g, ay, by);
        };
        launch(); launch(); CK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int it = 0; it < 20; ++it) {
            CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best;
        }
        CK(hipGetLastError());
        printf("%-16s %7.1f us  (%d tiles of %d lines, %.0f MB in + %.0f MB out -> %.2f TB/s)\n", names[mode], best * 1e3, (int)grid.x,
               kTL, nA * 8 / 1e6, nB * 8 / 1e6, (nA + nB) * 8 / (best * 1e-3) / 1e12);
    }
    return 0;
}

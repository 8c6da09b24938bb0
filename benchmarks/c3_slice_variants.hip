// c3_slice_variants.hip — C3's write stream (one shared beam x 4096 maps, 11.47 GB of result) with the store pattern of the fastest
// fill of profiles/r04_c3_write_ceiling.md: consecutive workgroups own ADJACENT slices of 4 KB (8 / 16 KB) of the contiguous
// (B, N, 7) output; a slice holds 146.3 particle rows, rows straddling a slice edge are computed by both neighbours, every lane
// stores 16-byte chunks of its workgroup's slice from an LDS image of it. The arithmetic is that of apply_shared_wave_kernel (the
// fma chain of chx_apply_affine7), checked against a host evaluation of three batch rows.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off benchmarks/c3_slice_variants.hip -o /tmp/c3s && /tmp/c3s
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));       \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));

// SLICE floats per workgroup (a multiple of 4 * THREADS / k); NT: nontemporal stores
template <int THREADS, int SLICE, bool NT>
__global__ __launch_bounds__(THREADS) void k_slice(const float* __restrict__ x, const float* __restrict__ R, float* __restrict__ out, long N,
                                                   long total) {
    __shared__ __attribute__((aligned(16))) float img[SLICE + 8];
    const long e0 = (long)blockIdx.x * SLICE;
    const long e1 = (e0 + SLICE < total) ? e0 + SLICE : total;
    const long g0 = e0 / 7, g1 = (e1 - 1) / 7;                  // global particle rows (b * N + n) this slice touches
    const int shift = (int)(e0 - g0 * 7);                       // floats of row g0 that belong to the previous slice
    for (int l = threadIdx.x; l <= (int)(g1 - g0); l += THREADS) {
        const long g = g0 + l;
        const long b = g / N, n = g - b * N;
        const float* __restrict__ Rb = R + b * 49;
        float xi[7], yi[7];
#pragma unroll
        for (int j = 0; j < 7; ++j) xi[j] = x[n * 7 + j];
#pragma unroll
        for (int a = 0; a < 7; ++a) {
            float acc = Rb[a * 7] * xi[0];
#pragma unroll
            for (int j = 1; j < 7; ++j) acc = fmaf(Rb[a * 7 + j], xi[j], acc);
            yi[a] = acc;
        }
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            const int pos = l * 7 + j - shift;
            if (pos >= 0 && pos < SLICE) img[pos] = yi[j];
        }
    }
    __syncthreads();
    const int nvec = (int)((e1 - e0) >> 2);
    v4f* o = reinterpret_cast<v4f*>(out + e0);
    const v4f* s = reinterpret_cast<const v4f*>(img);
    for (int i = threadIdx.x; i < nvec; i += THREADS) {
        if (NT) __builtin_nontemporal_store(s[i], o + i);
        else o[i] = s[i];
    }
}

// the same slices with the inputs brought in like the production kernel does: the slice's ~147 beam rows as ONE contiguous chunk
// through LDS (coalesced dword loads; the chunk starts at a multiple of 28 B, not of 16), the map through wave-uniform scalar loads
// when the slice lies inside one batch row (all but 1 in 683 slices)
template <int THREADS, int SLICE, bool NT>
__global__ __launch_bounds__(THREADS) void k_slice_staged(const float* __restrict__ x, const float* __restrict__ R, float* __restrict__ out,
                                                          long N, long total) {
    constexpr int ROWS = SLICE / 7 + 3;
    __shared__ __attribute__((aligned(16))) float img[SLICE + 8];
    __shared__ float xin[ROWS * 7];
    const long e0 = (long)blockIdx.x * SLICE;
    const long e1 = (e0 + SLICE < total) ? e0 + SLICE : total;
    const long g0 = e0 / 7, g1 = (e1 - 1) / 7;
    const int shift = (int)(e0 - g0 * 7);
    const int nrows = (int)(g1 - g0) + 1;
    const long b0 = g0 / N, b1 = g1 / N;
    const long n0 = g0 - b0 * N;
    if (b0 == b1) {
        const float* src = x + n0 * 7;
        for (int i = threadIdx.x; i < nrows * 7; i += THREADS) xin[i] = src[i];
    } else {
        for (int i = threadIdx.x; i < nrows * 7; i += THREADS) {
            const long g = g0 + i / 7;
            xin[i] = x[(g % N) * 7 + (i % 7)];
        }
    }
    __syncthreads();
    const float* __restrict__ Ru = R + b0 * 49;                 // wave-uniform: scalar loads
    for (int l = threadIdx.x; l < nrows; l += THREADS) {
        float xi[7], yi[7];
#pragma unroll
        for (int j = 0; j < 7; ++j) xi[j] = xin[l * 7 + j];
        if (b0 == b1) {
#pragma unroll
            for (int a = 0; a < 7; ++a) {
                float acc = Ru[a * 7] * xi[0];
#pragma unroll
                for (int j = 1; j < 7; ++j) acc = fmaf(Ru[a * 7 + j], xi[j], acc);
                yi[a] = acc;
            }
        } else {
            const float* __restrict__ Rb = R + ((g0 + l) / N) * 49;
#pragma unroll
            for (int a = 0; a < 7; ++a) {
                float acc = Rb[a * 7] * xi[0];
#pragma unroll
                for (int j = 1; j < 7; ++j) acc = fmaf(Rb[a * 7 + j], xi[j], acc);
                yi[a] = acc;
            }
        }
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            const int pos = l * 7 + j - shift;
            if (pos >= 0 && pos < SLICE) img[pos] = yi[j];
        }
    }
    __syncthreads();
    const int nvec = (int)((e1 - e0) >> 2);
    v4f* o = reinterpret_cast<v4f*>(out + e0);
    const v4f* sv = reinterpret_cast<const v4f*>(img);
    for (int i = threadIdx.x; i < nvec; i += THREADS) {
        if (NT) __builtin_nontemporal_store(sv[i], o + i);
        else o[i] = sv[i];
    }
}

template <int STORE>
__global__ __launch_bounds__(256) void k_fill_span(float* __restrict__ out, long nvec, long span, float val) {
    v4f* o = reinterpret_cast<v4f*>(out);
    const v4f v = {val, val, val, val};
    const long i0 = (long)blockIdx.x * span, i1 = (i0 + span < nvec) ? i0 + span : nvec;
    for (long i = i0 + threadIdx.x; i < i1; i += 256) {
        if (STORE == 0) __builtin_nontemporal_store(v, o + i);
        else o[i] = v;
    }
}

template <typename F>
float time_ms(F&& launch, int reps) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    launch(); launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

template <int THREADS, int SLICE, bool NT>
void run_staged(const float* x, const float* R, float* out, long B, long N, const std::vector<float>& ref, const long* ref_b) {
    const long total = B * N * 7;
    const long wgs = (total + SLICE - 1) / SLICE;
    auto launch = [&] { hipLaunchKernelGGL((k_slice_staged<THREADS, SLICE, NT>), dim3((unsigned)wgs), dim3(THREADS), 0, 0, x, R, out, N, total); };
    const float ms = time_ms(launch, 5);
    double maxerr = 0;
    std::vector<float> got(N * 7);
    for (int r = 0; r < 3; ++r) {
        CK(hipMemcpy(got.data(), out + ref_b[r] * N * 7, N * 28, hipMemcpyDeviceToHost));
        for (long i = 0; i < N * 7; ++i) maxerr = fmax(maxerr, fabs((double)got[i] - ref[r * N * 7 + i]));
    }
    printf("staged slice %5d B, %3d threads, %s stores: wgs %8ld : %7.3f ms  %5.2f TB/s  err %.1e\n", SLICE * 4, THREADS, NT ? "nt   " : "plain",
           wgs, ms, (double)total * 4 / ms / 1e9, maxerr);
    fflush(stdout);
}

template <int THREADS, int SLICE, bool NT>
void run(const float* x, const float* R, float* out, long B, long N, const std::vector<float>& ref, const long* ref_b) {
    const long total = B * N * 7;
    const long wgs = (total + SLICE - 1) / SLICE;
    auto launch = [&] { hipLaunchKernelGGL((k_slice<THREADS, SLICE, NT>), dim3((unsigned)wgs), dim3(THREADS), 0, 0, x, R, out, N, total); };
    const float ms = time_ms(launch, 5);
    double maxerr = 0;
    std::vector<float> got(N * 7);
    for (int r = 0; r < 3; ++r) {
        CK(hipMemcpy(got.data(), out + ref_b[r] * N * 7, N * 28, hipMemcpyDeviceToHost));
        for (long i = 0; i < N * 7; ++i) maxerr = fmax(maxerr, fabs((double)got[i] - ref[r * N * 7 + i]));
    }
    printf("slice %5d B, %3d threads, %s stores: wgs %8ld : %7.3f ms  %5.2f TB/s  err %.1e\n", SLICE * 4, THREADS, NT ? "nt   " : "plain", wgs, ms,
           (double)total * 4 / ms / 1e9, maxerr);
    fflush(stdout);
}

int main() {
    const long B = 4096, N = 100000;
    float *x, *R, *out;
    CK(hipMalloc(&x, N * 28));
    CK(hipMalloc(&R, B * 49 * 4));
    CK(hipMalloc(&out, B * N * 28));
    std::vector<float> hx(N * 7), hR(B * 49);
    srand(1);
    for (auto& v : hx) v = (rand() / (float)RAND_MAX - 0.5f) * 1e-3f;
    for (long i = 0; i < N; ++i) hx[i * 7 + 6] = 1.f;
    for (auto& v : hR) v = (rand() / (float)RAND_MAX - 0.5f);
    CK(hipMemcpy(x, hx.data(), N * 28, hipMemcpyHostToDevice));
    CK(hipMemcpy(R, hR.data(), B * 49 * 4, hipMemcpyHostToDevice));
    const long ref_b[3] = {0, 2049, 4095};
    std::vector<float> ref(3 * N * 7);
    for (int r = 0; r < 3; ++r)
        for (long i = 0; i < N; ++i) {
            const float* Rb = hR.data() + ref_b[r] * 49;
            for (int a = 0; a < 7; ++a) {
                float acc = Rb[a * 7] * hx[i * 7];
                for (int j = 1; j < 7; ++j) acc = fmaf(Rb[a * 7 + j], hx[i * 7 + j], acc);
                ref[(r * N + i) * 7 + a] = acc;
            }
        }
    const long nvec = B * N * 7 / 4;
    for (int rep = 0; rep < 2; ++rep) {
        for (long span : {256L, 512L, 1024L}) {
            const long g = (nvec + span - 1) / span;
            float ms = time_ms([&] { hipLaunchKernelGGL(k_fill_span<0>, dim3((unsigned)g), dim3(256), 0, 0, out, nvec, span, 1.5f); }, 5);
            printf("fill nt, %5ld B per workgroup (%ld wgs) : %7.3f ms %5.2f TB/s\n", span * 16, g, ms, (double)nvec * 16 / ms / 1e9);
        }
        run_staged<64, 1024, true>(x, R, out, B, N, ref, ref_b);
        run_staged<128, 1024, true>(x, R, out, B, N, ref, ref_b);
        run_staged<256, 1024, true>(x, R, out, B, N, ref, ref_b);
        run_staged<256, 2048, true>(x, R, out, B, N, ref, ref_b);
        run_staged<256, 4096, true>(x, R, out, B, N, ref, ref_b);
        run<64, 1024, true>(x, R, out, B, N, ref, ref_b);
        run<64, 1024, false>(x, R, out, B, N, ref, ref_b);
        run<128, 1024, true>(x, R, out, B, N, ref, ref_b);
        run<256, 1024, true>(x, R, out, B, N, ref, ref_b);
        run<256, 1024, false>(x, R, out, B, N, ref, ref_b);
        run<64, 2048, true>(x, R, out, B, N, ref, ref_b);
        run<128, 2048, true>(x, R, out, B, N, ref, ref_b);
        run<256, 2048, true>(x, R, out, B, N, ref, ref_b);
        run<256, 4096, true>(x, R, out, B, N, ref, ref_b);
        run<256, 7168, true>(x, R, out, B, N, ref, ref_b);
    }
    return 0;
}

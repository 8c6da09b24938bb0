// shared_wave_variants.hip — why does C3's write stream (one beam x 4096 maps, 11.47 GB written) sit at 5.7 TB/s when a
// linear fill of the same bytes reaches 6.9? Variants of apply_shared_wave_kernel (chx_apply.hip) and fills that keep its
// ADDRESS PATTERN but drop the arithmetic, timed with HIP events. Build & run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off benchmarks/shared_wave_variants.hip -o /tmp/swv && /tmp/swv
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cmath>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void apply7(const float* __restrict__ R, const float (&x)[7], float (&y)[7]) {
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        float acc = R[i * 7] * x[0];
#pragma unroll
        for (int j = 1; j < 7; ++j) acc = fmaf(R[i * 7 + j], x[j], acc);
        y[i] = acc;
    }
}

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// STORE: 0 = nt, 1 = plain
template <int STORE>
__device__ __forceinline__ void st16(v4f v, v4f* p) {
    if (STORE == 0) __builtin_nontemporal_store(v, p);
    else *p = v;
}

// ---- production structure. THREADS per workgroup, PPT rows per lane, SYNC: workgroup barrier per batch row (keeps the waves of
// a workgroup in step: their chunks of a batch row leave together), TRANSPOSE: blockIdx.x runs over the batch chunks
template <int THREADS, int PPT, int STORE, bool SYNC, bool TRANSPOSE, bool COMPUTE>
__global__ __launch_bounds__(THREADS) void k_shared(const float* __restrict__ x_in, const float* __restrict__ R,
                                                    float* __restrict__ x_out, long B, long N, long rows_per_chunk) {
    constexpr int TP = PPT * THREADS, WP = PPT * 64, WE = WP * 7, WV = WE / 4;
    __shared__ __attribute__((aligned(16))) float lds[TP * 7];
    const long tile = TRANSPOSE ? blockIdx.y : blockIdx.x, chunk = TRANSPOSE ? blockIdx.x : blockIdx.y;
    const long n0 = tile * TP;
    const int np = (int)((N - n0 < TP) ? (N - n0) : TP);
    const long b0 = chunk * rows_per_chunk;
    const long b1 = (b0 + rows_per_chunk < B) ? b0 + rows_per_chunk : B;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int e = threadIdx.x; e < np * 7; e += THREADS) lds[e] = x_in[n0 * 7 + e];
    __syncthreads();
    float x[PPT][7];
    float* wl = lds + wave * WE;
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const int p = wave * WP + k * 64 + lane;
#pragma unroll
        for (int j = 0; j < 7; ++j) x[k][j] = (p < np) ? wl[(k * 64 + lane) * 7 + j] : 0.f;
    }
    const int valid = (np - wave * WP < 0) ? 0 : ((np - wave * WP < WP) ? (np - wave * WP) : WP);
    const int vchunks = valid * 7 / 4;
    if (SYNC) __syncthreads();
    for (long b = b0; b < b1; ++b) {
        if (COMPUTE) {
            const float* __restrict__ Rb = R + b * 49;
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                float y[7];
                apply7(Rb, x[k], y);
#pragma unroll
                for (int j = 0; j < 7; ++j) wl[(k * 64 + lane) * 7 + j] = y[j];
            }
            wave_sync();
        }
        if (SYNC) __builtin_amdgcn_s_barrier();
        float* __restrict__ gout = x_out + (b * N + n0 + wave * WP) * 7;
        v4f* __restrict__ gv = reinterpret_cast<v4f*>(gout);
        const v4f* lv = reinterpret_cast<const v4f*>(wl);
#pragma unroll
        for (int c = 0; c < (WV + 63) / 64; ++c) {
            const int v = c * 64 + lane;
            if (v < vchunks) st16<STORE>(COMPUTE ? lv[v] : v4f{x[0][0], x[0][1], x[0][2], (float)b}, gv + v);
        }
        if (vchunks < WV)
            for (int e = vchunks * 4 + lane; e < valid * 7; e += 64) gout[e] = wl[e];
        if (COMPUTE) wave_sync();
    }
}

// ---- STRIDED: workgroup (tile, c of C) writes batch rows c, c + C, c + 2 C, ... — with tiles * C workgroups all resident the
// chip as a whole writes C consecutive batch rows at a time and marches through the output like a linear fill
template <int THREADS, int PPT, int STORE, bool TRANSPOSE>
__global__ __launch_bounds__(THREADS) void k_strided(const float* __restrict__ x_in, const float* __restrict__ R,
                                                     float* __restrict__ x_out, long B, long N, long C) {
    constexpr int TP = PPT * THREADS, WP = PPT * 64, WE = WP * 7, WV = WE / 4;
    __shared__ __attribute__((aligned(16))) float lds[TP * 7];
    const long tile = TRANSPOSE ? blockIdx.y : blockIdx.x, chunk = TRANSPOSE ? blockIdx.x : blockIdx.y;
    const long n0 = tile * TP;
    const int np = (int)((N - n0 < TP) ? (N - n0) : TP);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int e = threadIdx.x; e < np * 7; e += THREADS) lds[e] = x_in[n0 * 7 + e];
    __syncthreads();
    float x[PPT][7];
    float* wl = lds + wave * WE;
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const int p = wave * WP + k * 64 + lane;
#pragma unroll
        for (int j = 0; j < 7; ++j) x[k][j] = (p < np) ? wl[(k * 64 + lane) * 7 + j] : 0.f;
    }
    const int valid = (np - wave * WP < 0) ? 0 : ((np - wave * WP < WP) ? (np - wave * WP) : WP);
    const int vchunks = valid * 7 / 4;
    wave_sync();
    for (long b = chunk; b < B; b += C) {
        const float* __restrict__ Rb = R + b * 49;
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            float y[7];
            apply7(Rb, x[k], y);
#pragma unroll
            for (int j = 0; j < 7; ++j) wl[(k * 64 + lane) * 7 + j] = y[j];
        }
        wave_sync();
        float* __restrict__ gout = x_out + (b * N + n0 + wave * WP) * 7;
        v4f* __restrict__ gv = reinterpret_cast<v4f*>(gout);
        const v4f* lv = reinterpret_cast<const v4f*>(wl);
#pragma unroll
        for (int c = 0; c < (WV + 63) / 64; ++c) {
            const int v = c * 64 + lane;
            if (v < vchunks) st16<STORE>(lv[v], gv + v);
        }
        if (vchunks < WV)
            for (int e = vchunks * 4 + lane; e < valid * 7; e += 64) gout[e] = wl[e];
        wave_sync();
    }
}

// ---- registers only: a lane keeps PPT rows and computes y for them; the 7 * PPT outputs of a lane are written as dwords
// straight from registers (strided 28 B: seven 4-byte stores per row — the uncoalesced baseline)
// ---- linear fill, 16 bytes per lane, grid-stride
template <int STORE>
__global__ __launch_bounds__(256) void k_fill(float* __restrict__ out, long nvec, float val) {
    v4f* o = reinterpret_cast<v4f*>(out);
    const v4f v = {val, val, val, val};
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long)gridDim.x * 256) st16<STORE>(v, o + i);
}
// linear fill, one contiguous span per workgroup
template <int STORE>
__global__ __launch_bounds__(256) void k_fill_span(float* __restrict__ out, long nvec, long span, float val) {
    v4f* o = reinterpret_cast<v4f*>(out);
    const v4f v = {val, val, val, val};
    const long i0 = (long)blockIdx.x * span, i1 = (i0 + span < nvec) ? i0 + span : nvec;
    for (long i = i0 + threadIdx.x; i < i1; i += 256) st16<STORE>(v, o + i);
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
    CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
    return ms / reps;
}

template <int THREADS, int PPT, int STORE, bool SYNC, bool TRANSPOSE, bool COMPUTE>
void run_shared(const char* name, const float* x, const float* R, float* out, long B, long N, long target_wgs, long min_rows,
                const std::vector<float>& ref_rows, const long* ref_b, int nref) {
    constexpr int TP = PPT * THREADS;
    const long tiles = (N + TP - 1) / TP;
    long chunks = (target_wgs + tiles - 1) / tiles;
    if (chunks > B) chunks = B;
    long rows = (B + chunks - 1) / chunks;
    if (rows < min_rows) rows = min_rows;
    chunks = (B + rows - 1) / rows;
    dim3 grid = TRANSPOSE ? dim3((unsigned)chunks, (unsigned)tiles) : dim3((unsigned)tiles, (unsigned)chunks);
    auto launch = [&] {
        hipLaunchKernelGGL((k_shared<THREADS, PPT, STORE, SYNC, TRANSPOSE, COMPUTE>), grid, dim3(THREADS), 0, 0, x, R, out, B, N, rows);
    };
    const float ms = time_ms(launch, 5);
    const double bytes = (double)B * N * 28.0;
    double maxerr = -1;
    if (COMPUTE && nref) {
        maxerr = 0;
        std::vector<float> got(N * 7);
        for (int r = 0; r < nref; ++r) {
            CK(hipMemcpy(got.data(), out + ref_b[r] * N * 7, N * 28, hipMemcpyDeviceToHost));
            for (long i = 0; i < N * 7; ++i) maxerr = fmax(maxerr, fabs((double)got[i] - ref_rows[r * N * 7 + i]));
        }
    }
    printf("%-58s wgs %6ld rows/chunk %4ld : %7.3f ms  %6.2f TB/s  err %.1e\n", name, tiles * chunks, rows, ms, bytes / ms / 1e9, maxerr);
    fflush(stdout);
}

template <int THREADS, int PPT, int STORE, bool TRANSPOSE>
void run_strided(const char* name, const float* x, const float* R, float* out, long B, long N, long C,
                 const std::vector<float>& ref_rows, const long* ref_b, int nref) {
    constexpr int TP = PPT * THREADS;
    const long tiles = (N + TP - 1) / TP;
    dim3 grid = TRANSPOSE ? dim3((unsigned)C, (unsigned)tiles) : dim3((unsigned)tiles, (unsigned)C);
    auto launch = [&] { hipLaunchKernelGGL((k_strided<THREADS, PPT, STORE, TRANSPOSE>), grid, dim3(THREADS), 0, 0, x, R, out, B, N, C); };
    const float ms = time_ms(launch, 5);
    const double bytes = (double)B * N * 28.0;
    double maxerr = 0;
    std::vector<float> got(N * 7);
    for (int r = 0; r < nref; ++r) {
        CK(hipMemcpy(got.data(), out + ref_b[r] * N * 7, N * 28, hipMemcpyDeviceToHost));
        for (long i = 0; i < N * 7; ++i) maxerr = fmax(maxerr, fabs((double)got[i] - ref_rows[r * N * 7 + i]));
    }
    printf("%-44s wgs %6ld C %4ld : %7.3f ms  %6.2f TB/s  err %.1e\n", name, tiles * C, C, ms, bytes / ms / 1e9, maxerr);
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
    // reference rows on the host
    const long ref_b[3] = {0, 2049, 4095};
    std::vector<float> ref(3 * N * 7);
    for (int r = 0; r < 3; ++r)
        for (long i = 0; i < N; ++i) {
            float xi[7], yi[7];
            for (int j = 0; j < 7; ++j) xi[j] = hx[i * 7 + j];
            const float* Rb = hR.data() + ref_b[r] * 49;
            for (int a = 0; a < 7; ++a) {
                float acc = Rb[a * 7] * xi[0];
                for (int j = 1; j < 7; ++j) acc = fmaf(Rb[a * 7 + j], xi[j], acc);
                yi[a] = acc;
            }
            for (int j = 0; j < 7; ++j) ref[(r * N + i) * 7 + j] = yi[j];
        }
    const long nvec = B * N * 7 / 4;
    const double bytes = (double)B * N * 28.0;
    {
        float ms = time_ms([&] { CK(hipMemsetAsync(out, 0, (size_t)B * N * 28, 0)); }, 5);
        printf("hipMemsetAsync : %7.3f ms %6.2f TB/s\n", ms, bytes / ms / 1e9);
        ms = time_ms([&] { CK(hipMemsetD32Async((hipDeviceptr_t)out, 0x3fc00000, (size_t)B * N * 7, 0)); }, 5);
        printf("hipMemsetD32Async : %7.3f ms %6.2f TB/s\n", ms, bytes / ms / 1e9);
    }
    for (int grid : {2048, 8192, 65536}) {
        float ms = time_ms([&] { hipLaunchKernelGGL(k_fill<0>, dim3(grid), dim3(256), 0, 0, out, nvec, 1.5f); }, 5);
        printf("fill nt grid-stride grid %6d : %7.3f ms %6.2f TB/s\n", grid, ms, bytes / ms / 1e9);
        ms = time_ms([&] { hipLaunchKernelGGL(k_fill<1>, dim3(grid), dim3(256), 0, 0, out, nvec, 1.5f); }, 5);
        printf("fill plain grid-stride grid %6d : %7.3f ms %6.2f TB/s\n", grid, ms, bytes / ms / 1e9);
    }
    for (long span : {256L, 512L, 1024L, 1792L, 7168L, 28672L}) {   // 16-byte chunks per workgroup: 4 KB ... 448 KB
        const long g = (nvec + span - 1) / span;
        float ms = time_ms([&] { hipLaunchKernelGGL(k_fill_span<0>, dim3((unsigned)g), dim3(256), 0, 0, out, nvec, span, 1.5f); }, 5);
        printf("fill nt span %6ld chunks (%ld wgs) : %7.3f ms %6.2f TB/s\n", span, g, ms, bytes / ms / 1e9);
        ms = time_ms([&] { hipLaunchKernelGGL(k_fill_span<1>, dim3((unsigned)g), dim3(256), 0, 0, out, nvec, span, 1.5f); }, 5);
        printf("fill plain span %6ld chunks (%ld wgs) : %7.3f ms %6.2f TB/s\n", span, g, ms, bytes / ms / 1e9);
    }
    fflush(stdout);
#define RUN(T, P, S, SY, TR, C, wgs, mr) run_shared<T, P, S, SY, TR, C>(#T "thr ppt" #P " store" #S " sync" #SY " transp" #TR " compute" #C, x, R, out, B, N, wgs, mr, ref, ref_b, 3)
    for (int rep = 0; rep < 2; ++rep) {
    RUN(256, 4, 0, false, false, true, 8192, 8);     // production
    RUN(256, 1, 0, false, false, true, 4000000, 2);
    RUN(256, 1, 0, false, false, true, 4000000, 3);
    RUN(256, 1, 0, false, false, true, 4000000, 4);
    RUN(256, 1, 0, false, false, true, 4000000, 6);
    RUN(256, 1, 0, false, false, true, 4000000, 8);
    RUN(256, 1, 0, false, false, true, 4000000, 16);
    RUN(256, 2, 0, false, false, true, 4000000, 2);
    RUN(256, 2, 0, false, false, true, 4000000, 4);
    RUN(256, 2, 0, false, false, true, 4000000, 8);
    RUN(128, 2, 0, false, false, true, 4000000, 2);
    RUN(128, 2, 0, false, false, true, 4000000, 4);
    RUN(128, 1, 0, false, false, true, 4000000, 4);
    RUN(64, 2, 0, false, false, true, 4000000, 4);
    RUN(64, 4, 0, false, false, true, 4000000, 2);
    RUN(64, 4, 0, false, false, true, 4000000, 4);
    RUN(512, 1, 0, false, false, true, 4000000, 2);
    RUN(512, 1, 0, false, false, true, 4000000, 4);
    }
    return 0;
}

// sort_probe.hip — cost model for the tile-ordered space-charge pipeline (round 2): how expensive are the memory patterns a
// counting sort by 8^3-cell tile needs on MI355X, measured in isolation on a Gaussian beam (1e6 particles, 128^3 grid,
// +-3 sigma extent)?
//   count      read x,y,tau of every 28-byte row, LDS histogram per workgroup (two grid shapes)
//   scat_*     scatter pass with LDS cursors: 4-byte particle index / 16-byte / 28-byte / 32-byte record per particle
//   scat_lds   the same 28-byte record, first sorted by tile inside the workgroup's LDS, then written as runs
//   gath_rows  tile-ordered read of whole 28-byte rows through the index array (what a tile-ordered kick does)
//   scatrows   tile-ordered write of 28-byte rows back to original order
//   nodes      the current gather: 8 random 16-byte node fetches per particle from a 32 MB field grid, particle order
//   nodes_lds  tile-ordered: 9^3 x 16 B node block in LDS per tile, rows through the index array, rows written back
// Build: hipcc --offload-arch=gfx950 -O3 -o sort_probe benchmarks/sort_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int G = 128, TD = 8, NTA = G / TD, NT = NTA * NTA * NTA;  // 4096 tiles

__device__ __forceinline__ void locate(const float* __restrict__ x, int64_t n, int (&i)[3], float (&f)[3]) {
    const int cols[3] = {0, 2, 4};
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const float v = x[n * 7 + cols[d]];
        const float p = (v + 3.0f) / (6.0f / G) - 0.5f;
        const float fl = floorf(p);
        i[d] = (int)fl;
        f[d] = p - fl;
    }
}
__device__ __forceinline__ int home_tile(const int (&i)[3]) {
    int t[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) { int c = i[d] < 0 ? 0 : (i[d] > G - 1 ? G - 1 : i[d]); t[d] = c / TD; }
    return (t[0] * NTA + t[1]) * NTA + t[2];
}

// ---- count: one LDS histogram per workgroup, written wg-major
template <int THREADS>
__global__ __launch_bounds__(THREADS) void count_kernel(const float* __restrict__ x, int64_t N, int* __restrict__ counts) {
    __shared__ int hist[NT];
    for (int t = threadIdx.x; t < NT; t += THREADS) hist[t] = 0;
    __syncthreads();
    const int64_t per = (N + gridDim.x - 1) / gridDim.x;
    const int64_t n0 = blockIdx.x * per, n1 = n0 + per < N ? n0 + per : N;
    for (int64_t n = n0 + threadIdx.x; n < n1; n += THREADS) {
        int i[3]; float f[3];
        locate(x, n, i, f);
        atomicAdd(&hist[home_tile(i)], 1);
    }
    __syncthreads();
    for (int t = threadIdx.x; t < NT; t += THREADS) counts[(int64_t)blockIdx.x * NT + t] = hist[t];
}

struct Rec16 { int idx; float a, b, c; };
struct Rec28 { int i[3]; float f[3]; float c; };
struct __attribute__((aligned(16))) Rec32 { int i[3]; float f[3]; float c; int pad; };

// ---- scatter with LDS cursors (absolute positions), MODE 0: index, 1: Rec16, 2: Rec28, 3: Rec32
template <int THREADS, int MODE>
__global__ __launch_bounds__(THREADS) void scatter_kernel(const float* __restrict__ x, int64_t N,
                                                        const int* __restrict__ cursors, void* __restrict__ out) {
    __shared__ int cur[NT];
    for (int t = threadIdx.x; t < NT; t += THREADS) cur[t] = cursors[(int64_t)blockIdx.x * NT + t];
    __syncthreads();
    const int64_t per = (N + gridDim.x - 1) / gridDim.x;
    const int64_t n0 = blockIdx.x * per, n1 = n0 + per < N ? n0 + per : N;
    for (int64_t n = n0 + threadIdx.x; n < n1; n += THREADS) {
        int i[3]; float f[3];
        locate(x, n, i, f);
        const int pos = atomicAdd(&cur[home_tile(i)], 1);
        if (MODE == 0) ((int*)out)[pos] = (int)n;
        if (MODE == 1) { Rec16 r{(int)n, f[0], f[1], f[2]}; ((Rec16*)out)[pos] = r; }
        if (MODE == 2) { Rec28 r{{i[0], i[1], i[2]}, {f[0], f[1], f[2]}, 1.0f}; ((Rec28*)out)[pos] = r; }
        if (MODE == 3) { Rec32 r{{i[0], i[1], i[2]}, {f[0], f[1], f[2]}, 1.0f, 0}; ((Rec32*)out)[pos] = r; }
    }
}

// ---- scatter, records sorted by tile inside LDS first: the workgroup's CH particles are ranked with a small LDS
// histogram over the tiles THEY touch (open-addressed: CH particles touch <= CH tiles), written out run by run.
// Simplification for the probe: rank = order of arrival at an LDS cursor per tile (hist over all NT tiles, 16 KB),
// local offsets by a workgroup scan over NT, then each lane copies record j to global cursor[tile_j] + (j - lstart[tile_j]).
template <int THREADS, int CH>
__global__ __launch_bounds__(THREADS) void scatter_lds_kernel(const float* __restrict__ x, int64_t N,
                                                            const int* __restrict__ cursors, Rec28* __restrict__ out) {
    __shared__ int hist[NT];        // local counts -> local exclusive starts
    __shared__ int lcur[NT];        // local cursors
    __shared__ Rec28 recs[CH];
    __shared__ short tile_of[CH];
    __shared__ int wsum[THREADS / 64];
    const int64_t per = (N + gridDim.x - 1) / gridDim.x;
    const int64_t w0 = blockIdx.x * per, w1 = w0 + per < N ? w0 + per : N;
    const int* gcur = cursors + (int64_t)blockIdx.x * NT;
    int done_before = 0;  // records of earlier chunks of this workgroup per tile are tracked in gofs
    __shared__ int gofs[NT];
    for (int t = threadIdx.x; t < NT; t += THREADS) gofs[t] = gcur[t];
    for (int64_t c0 = w0; c0 < w1; c0 += CH) {
        const int cn = (int)(c0 + CH < w1 ? CH : w1 - c0);
        for (int t = threadIdx.x; t < NT; t += THREADS) hist[t] = 0;
        __syncthreads();
        int my_tile[(CH + THREADS - 1) / THREADS], my_rank[(CH + THREADS - 1) / THREADS];
        Rec28 my_rec[(CH + THREADS - 1) / THREADS];
        int k = 0;
        for (int j = threadIdx.x; j < cn; j += THREADS, ++k) {
            int i[3]; float f[3];
            locate(x, c0 + j, i, f);
            my_tile[k] = home_tile(i);
            my_rank[k] = atomicAdd(&hist[my_tile[k]], 1);
            my_rec[k] = Rec28{{i[0], i[1], i[2]}, {f[0], f[1], f[2]}, 1.0f};
        }
        __syncthreads();
        // exclusive scan of hist over NT (THREADS lanes, NT / THREADS each)
        {
            constexpr int PER = NT / THREADS;
            int v[PER], s = 0;
#pragma unroll
            for (int q = 0; q < PER; ++q) { v[q] = hist[threadIdx.x * PER + q]; s += v[q]; }
            int incl = s;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d, 64); if ((threadIdx.x & 63) >= d) incl += o; }
            if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
            __syncthreads();
            int base = 0;
            for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) base += wsum[w];
            int run = base + incl - s;
#pragma unroll
            for (int q = 0; q < PER; ++q) { lcur[threadIdx.x * PER + q] = run; run += v[q]; }
        }
        __syncthreads();
        k = 0;
        for (int j = threadIdx.x; j < cn; j += THREADS, ++k) {
            const int slot = lcur[my_tile[k]] + my_rank[k];
            recs[slot] = my_rec[k];
            tile_of[slot] = (short)my_tile[k];
        }
        __syncthreads();
        for (int j = threadIdx.x; j < cn; j += THREADS) {
            const int t = tile_of[j];
            out[gofs[t] + (j - lcur[t])] = recs[j];
        }
        __syncthreads();
        for (int t = threadIdx.x; t < NT; t += THREADS) gofs[t] += hist[t];
        (void)done_before;
    }
}

// ---- tile-ordered row gather: perm sorted by tile; sum of the row written in tile order (coalesced)
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ x, const int* __restrict__ perm, int64_t N,
                                                         float* __restrict__ out) {
    for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < N; j += (int64_t)gridDim.x * 256) {
        const int n = perm[j];
        float s = 0;
#pragma unroll
        for (int c = 0; c < 7; ++c) s += x[(int64_t)n * 7 + c];
        out[j] = s;
    }
}
// ---- tile-ordered row read + scatter back to original order
__global__ __launch_bounds__(256) void scatter_rows_kernel(const float* __restrict__ x, const int* __restrict__ perm, int64_t N,
                                                          float* __restrict__ out) {
    for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < N; j += (int64_t)gridDim.x * 256) {
        const int n = perm[j];
        float r[7];
#pragma unroll
        for (int c = 0; c < 7; ++c) r[c] = x[(int64_t)n * 7 + c];
#pragma unroll
        for (int c = 0; c < 7; ++c) out[(int64_t)n * 7 + c] = r[c] * 1.0001f;
    }
}

// ---- particle-order gather of 8 field nodes (current sc_particle_kernel pattern), rows through an LDS-free path
__global__ __launch_bounds__(256) void nodes_kernel(const float* __restrict__ x, const float4* __restrict__ F, int64_t N,
                                                   float* __restrict__ out) {
    for (int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x; n < N; n += (int64_t)gridDim.x * 256) {
        int i[3]; float f[3];
        locate(x, n, i, f);
        float4 v[8];
#pragma unroll
        for (int o = 0; o < 8; ++o) {
            int a = i[0] + (o >> 2), b = i[1] + ((o >> 1) & 1), c = i[2] + (o & 1);
            a = a < 0 ? 0 : (a > G - 1 ? G - 1 : a); b = b < 0 ? 0 : (b > G - 1 ? G - 1 : b); c = c < 0 ? 0 : (c > G - 1 ? G - 1 : c);
            v[o] = F[((int64_t)a * G + b) * G + c];
        }
        float sx = 0, sy = 0, sz = 0;
#pragma unroll
        for (int o = 0; o < 8; ++o) {
            const float w = ((o >> 2) ? f[0] : 1 - f[0]) * (((o >> 1) & 1) ? f[1] : 1 - f[1]) * ((o & 1) ? f[2] : 1 - f[2]);
            sx += w * v[o].x; sy += w * v[o].y; sz += w * v[o].z;
        }
        float r[7];
#pragma unroll
        for (int c = 0; c < 7; ++c) r[c] = x[n * 7 + c];
        r[1] += sx; r[3] += sy; r[5] += sz;
#pragma unroll
        for (int c = 0; c < 7; ++c) out[n * 7 + c] = r[c];
    }
}

// ---- tile-ordered: node block (TD+2)^3 x 16 B in LDS, one workgroup per tile, rows through perm, rows written back
template <int THREADS>
__global__ __launch_bounds__(THREADS) void nodes_lds_kernel(const float* __restrict__ x, const float4* __restrict__ F,
                                                          const int* __restrict__ perm, const int* __restrict__ tile_start,
                                                          float* __restrict__ out) {
    constexpr int LD = TD + 2;
    __shared__ float4 blk[LD * LD * LD];
    const int t = blockIdx.x;
    const int beg = tile_start[t], end = tile_start[t + 1];
    if (beg >= end) return;
    const int tz = t % NTA, ty = (t / NTA) % NTA, tx = t / (NTA * NTA);
    for (int c = threadIdx.x; c < LD * LD * LD; c += THREADS) {
        const int lz = c % LD, ly = (c / LD) % LD, lx = c / (LD * LD);
        int a = tx * TD - 1 + lx, b = ty * TD - 1 + ly, cc = tz * TD - 1 + lz;
        a = a < 0 ? 0 : (a > G - 1 ? G - 1 : a); b = b < 0 ? 0 : (b > G - 1 ? G - 1 : b); cc = cc < 0 ? 0 : (cc > G - 1 ? G - 1 : cc);
        blk[c] = F[((int64_t)a * G + b) * G + cc];
    }
    __syncthreads();
    for (int j = beg + threadIdx.x; j < end; j += THREADS) {
        const int n = perm[j];
        float r[7];
#pragma unroll
        for (int c = 0; c < 7; ++c) r[c] = x[(int64_t)n * 7 + c];
        int i[3]; float f[3];
        {
            const int cols[3] = {0, 2, 4};
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const float p = (r[cols[d]] + 3.0f) / (6.0f / G) - 0.5f;
                const float fl = floorf(p);
                i[d] = (int)fl; f[d] = p - fl;
            }
        }
        const int lx = i[0] - (tx * TD - 1), ly = i[1] - (ty * TD - 1), lz = i[2] - (tz * TD - 1);
        float sx = 0, sy = 0, sz = 0;
#pragma unroll
        for (int o = 0; o < 8; ++o) {
            int a = lx + (o >> 2), b = ly + ((o >> 1) & 1), c = lz + (o & 1);
            a = a < 0 ? 0 : (a > LD - 1 ? LD - 1 : a); b = b < 0 ? 0 : (b > LD - 1 ? LD - 1 : b); c = c < 0 ? 0 : (c > LD - 1 ? LD - 1 : c);
            const float4 v = blk[(a * LD + b) * LD + c];
            const float w = ((o >> 2) ? f[0] : 1 - f[0]) * (((o >> 1) & 1) ? f[1] : 1 - f[1]) * ((o & 1) ? f[2] : 1 - f[2]);
            sx += w * v.x; sy += w * v.y; sz += w * v.z;
        }
        r[1] += sx; r[3] += sy; r[5] += sz;
#pragma unroll
        for (int c = 0; c < 7; ++c) out[(int64_t)n * 7 + c] = r[c];
    }
}

// ---- tile-ordered deposit from the index array: LDS tile (TD+1)^3 fp64 (low-side halo), halo to a scratch block array
template <int THREADS>
__global__ __launch_bounds__(THREADS) void deposit_perm_kernel(const float* __restrict__ x, const int* __restrict__ perm,
                                                             const int* __restrict__ tile_start, float* __restrict__ blocks) {
    constexpr int LD = TD + 1;
    __shared__ double tile[LD * LD * LD];
    const int t = blockIdx.x;
    const int beg = tile_start[t], end = tile_start[t + 1];
    const int tz = t % NTA, ty = (t / NTA) % NTA, tx = t / (NTA * NTA);
    for (int c = threadIdx.x; c < LD * LD * LD; c += THREADS) tile[c] = 0.0;
    __syncthreads();
    for (int j = beg + threadIdx.x; j < end; j += THREADS) {
        const int n = perm[j];
        int i[3]; float f[3];
        locate(x, n, i, f);
        // home tile = clamp(i) / TD; cells i, i+1 -> local 0..TD (high-side halo in this probe)
        const int lx = i[0] - tx * TD, ly = i[1] - ty * TD, lz = i[2] - tz * TD;
#pragma unroll
        for (int o = 0; o < 8; ++o) {
            const int a = lx + (o >> 2), b = ly + ((o >> 1) & 1), c = lz + (o & 1);
            const float w = ((o >> 2) ? f[0] : 1 - f[0]) * (((o >> 1) & 1) ? f[1] : 1 - f[1]) * ((o & 1) ? f[2] : 1 - f[2]);
            if (a >= 0 && a < LD && b >= 0 && b < LD && c >= 0 && c < LD) unsafeAtomicAdd(&tile[(a * LD + b) * LD + c], (double)w);
        }
    }
    __syncthreads();
    float* ob = blocks + (int64_t)t * (LD * LD * LD);
    for (int c = threadIdx.x; c < LD * LD * LD; c += THREADS) ob[c] = (float)tile[c];
}
// assemble: grid cell = sum of the <= 8 blocks that hold it
__global__ __launch_bounds__(256) void assemble_kernel(const float* __restrict__ blocks, float* __restrict__ grid) {
    constexpr int LD = TD + 1;
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= (int64_t)G * G * G) return;
    const int z = c % G, y = (c / G) % G, xx = c / (G * G);
    float s = 0;
#pragma unroll
    for (int o = 0; o < 8; ++o) {
        // block (tx, ty, tz) holds cell at local l = cell - t * TD, l in [0, TD]
        int t3[3], l3[3];
        const int cell[3] = {xx, y, z};
        bool ok = true;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const int use_prev = (o >> (2 - d)) & 1;
            int t = cell[d] / TD, l = cell[d] - t * TD;
            if (use_prev) { if (l != 0 || t == 0) ok = false; t -= 1; l = TD; }
            t3[d] = t; l3[d] = l;
        }
        if (ok) s += blocks[(int64_t)((t3[0] * NTA + t3[1]) * NTA + t3[2]) * (LD * LD * LD) + (l3[0] * LD + l3[1]) * LD + l3[2]];
    }
    grid[c] = s;
}

template <typename F>
float time_it(const char* name, F&& launch, int iters = 20, double bytes = 0) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch(); launch();
    CK(hipDeviceSynchronize());
    float best = 1e30f, sum = 0;
    for (int i = 0; i < iters; ++i) {
        CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms); sum += ms;
    }
    CK(hipGetLastError());
    printf("%-34s min %8.1f us  avg %8.1f us", name, best * 1e3, sum / iters * 1e3);
    if (bytes > 0) printf("  (%.2f TB/s of %.0f MB)", bytes / (best * 1e-3) / 1e12, bytes / 1e6);
    printf("\n");
    return best;
}

int main() {
    const int64_t N = 1000000;
    std::vector<float> hx(N * 7);
    std::mt19937 rng(1234);
    std::normal_distribution<float> nd(0.f, 1.f);
    for (int64_t n = 0; n < N; ++n) {
        for (int c = 0; c < 6; ++c) { float v; do { v = nd(rng); } while (fabsf(v) >= 2.999f); hx[n * 7 + c] = v; }
        hx[n * 7 + 6] = 1.f;
    }
    float *x, *out, *blocks, *grid; float4* F; int *counts, *cursors, *perm, *tile_start; void* recs;
    CK(hipMalloc(&x, N * 28)); CK(hipMalloc(&out, N * 28)); CK(hipMalloc(&F, (size_t)G * G * G * 16));
    CK(hipMalloc(&counts, (size_t)2048 * NT * 4)); CK(hipMalloc(&cursors, (size_t)2048 * NT * 4));
    CK(hipMalloc(&perm, N * 4)); CK(hipMalloc(&tile_start, (NT + 1) * 4)); CK(hipMalloc(&recs, N * 32));
    CK(hipMalloc(&blocks, (size_t)NT * 729 * 4)); CK(hipMalloc(&grid, (size_t)G * G * G * 4));
    CK(hipMemcpy(x, hx.data(), N * 28, hipMemcpyHostToDevice));
    CK(hipMemset(F, 0, (size_t)G * G * G * 16));

    for (int wgs : {256, 512, 1024}) {
        char nm[64];
        snprintf(nm, 64, "count %d x 1024", wgs);
        time_it(nm, [&] { count_kernel<1024><<<wgs, 1024>>>(x, N, counts); }, 20, N * 28.0);
        snprintf(nm, 64, "count %d x 256", wgs);
        time_it(nm, [&] { count_kernel<256><<<wgs, 256>>>(x, N, counts); }, 20, N * 28.0);
    }
    // cursors for a given number of workgroups (host scan: tile-major, wg-minor)
    std::vector<int> hts(NT + 1);
    auto make_cursors = [&](int wgs, int threads) {
        if (threads == 1024) count_kernel<1024><<<wgs, 1024>>>(x, N, counts); else count_kernel<256><<<wgs, 256>>>(x, N, counts);
        std::vector<int> hc((size_t)wgs * NT), cur((size_t)wgs * NT);
        CK(hipMemcpy(hc.data(), counts, hc.size() * 4, hipMemcpyDeviceToHost));
        int run = 0;
        for (int t = 0; t < NT; ++t) { hts[t] = run; for (int w = 0; w < wgs; ++w) { cur[(size_t)w * NT + t] = run; run += hc[(size_t)w * NT + t]; } }
        hts[NT] = run;
        CK(hipMemcpy(cursors, cur.data(), cur.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(tile_start, hts.data(), (NT + 1) * 4, hipMemcpyHostToDevice));
    };
    for (int wgs : {256, 1024}) {
        make_cursors(wgs, 1024);
        char nm[64];
        snprintf(nm, 64, "scat idx4  %d x 1024", wgs);
        time_it(nm, [&] { scatter_kernel<1024, 0><<<wgs, 1024>>>(x, N, cursors, perm); });
        snprintf(nm, 64, "scat rec16 %d x 1024", wgs);
        time_it(nm, [&] { scatter_kernel<1024, 1><<<wgs, 1024>>>(x, N, cursors, recs); });
        snprintf(nm, 64, "scat rec28 %d x 1024", wgs);
        time_it(nm, [&] { scatter_kernel<1024, 2><<<wgs, 1024>>>(x, N, cursors, recs); });
        snprintf(nm, 64, "scat rec32 %d x 1024", wgs);
        time_it(nm, [&] { scatter_kernel<1024, 3><<<wgs, 1024>>>(x, N, cursors, recs); });
    }
    make_cursors(256, 1024);
    time_it("scat rec28 via LDS sort 256x1024", [&] { scatter_lds_kernel<1024, 2048><<<256, 1024>>>(x, N, cursors, (Rec28*)recs); });
    make_cursors(256, 256);
    time_it("scat idx4  256 x 256", [&] { scatter_kernel<256, 0><<<256, 256>>>(x, N, cursors, perm); });
    // perm for the tile-ordered kernels
    make_cursors(256, 1024);
    scatter_kernel<1024, 0><<<256, 1024>>>(x, N, cursors, perm);
    CK(hipDeviceSynchronize());
    int hot = 0; for (int t = 0; t < NT; ++t) hot = std::max(hot, hts[t + 1] - hts[t]);
    printf("fullest tile: %d particles of %lld (mean %.0f)\n", hot, (long long)N, (double)N / NT);
    time_it("gather rows by perm (2048 wg)", [&] { gather_rows_kernel<<<2048, 256>>>(x, perm, N, out); });
    time_it("rows by perm -> rows scattered", [&] { scatter_rows_kernel<<<2048, 256>>>(x, perm, N, out); });
    time_it("nodes: 8 x 16 B, particle order", [&] { nodes_kernel<<<2048, 256>>>(x, F, N, out); });
    time_it("nodes in LDS, tile order, 256 thr", [&] { nodes_lds_kernel<256><<<NT, 256>>>(x, F, perm, tile_start, out); });
    time_it("nodes in LDS, tile order, 128 thr", [&] { nodes_lds_kernel<128><<<NT, 128>>>(x, F, perm, tile_start, out); });
    time_it("deposit by perm -> blocks, 256 thr", [&] { deposit_perm_kernel<256><<<NT, 256>>>(x, perm, tile_start, blocks); });
    time_it("deposit by perm -> blocks, 128 thr", [&] { deposit_perm_kernel<128><<<NT, 128>>>(x, perm, tile_start, blocks); });
    time_it("assemble blocks -> grid", [&] { assemble_kernel<<<(G * G * G + 255) / 256, 256>>>(blocks, grid); });
    // checksum: total deposited weight must equal N
    std::vector<float> hg((size_t)G * G * G);
    CK(hipMemcpy(hg.data(), grid, hg.size() * 4, hipMemcpyDeviceToHost));
    double tot = 0; for (float v : hg) tot += v;
    printf("assembled grid total %.3f (expected %lld)\n", tot, (long long)N);
    return 0;
}

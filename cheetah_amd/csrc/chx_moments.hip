// chx_moments.hip — weighted beam moments and the reductions of the backward pass.
//
// Replaces the per-property reductions of ParticleBeam (cheetah/particles/particle_beam.py:1672-1943)
// built on utils/statistics.py:4-62: every `mu_*`, `sigma_*`, `cov_*` of the reference re-reads a
// stride-7 column 3-4 times; here ONE call streams the particle array twice (mean pass + centred
// pass, the reference's two-pass algorithm, statistics.py:41-46) and returns all 6 means and the
// 21 covariances. Accumulation is fp64 in registers -> wavefront shuffle reduction -> LDS across
// the 4 waves -> per-workgroup partials in the caller's workspace -> a second tiny kernel sums the
// partials in a fixed order (deterministic, no float atomics).
// The same machinery gives dR = sum_n dY^T X for the apply backward.
#include <cstdlib>

#include "chx_common.h"
#include "chx_moments_dev.h"
#include "chx_sc_math.h"

namespace {

template <typename T> struct red_cfg;
template <> struct red_cfg<float> { static constexpr int PPT = 2; };
template <> struct red_cfg<double> { static constexpr int PPT = 1; };

__host__ __device__ inline int64_t red_nblk(int64_t B, int64_t N, int tile_rows) {
    int64_t tiles = (N + tile_rows - 1) / tile_rows;
    // one workgroup per CU in total: measured on MI355X (1e6 particles, 29 accumulators) 256 workgroups 18 us,
    // 512: 22 us, 1024: 33 us, 128: 23 us — the per-workgroup tree reduction and the partials pass dominate beyond that
    int64_t cap = 256 / B;
    if (cap < 1) cap = 1;
    return tiles < cap ? tiles : cap;
}

// Workgroups of the one-pass moments kernel (per batch row). With the DPP block reduction the per-workgroup epilogue is
// ~350 VALU instructions, so the grid is sized for latency hiding (4 workgroups per CU), not to minimise the number of
// reductions. CHX_TUNE_MOMENTS_WGS overrides the total (benchmarks only; read once).
inline int64_t onepass_total_wgs() {
    static const int64_t v = [] {
        const char* e = getenv("CHX_TUNE_MOMENTS_WGS");
        const long n = e ? atol(e) : 0;
        return (int64_t)(n >= 1 && n <= 1024 ? n : 512);
    }();
    return v;
}
inline int64_t onepass_nblk(int64_t B, int64_t N, int tile_rows) {
    const int64_t tiles = (N + tile_rows - 1) / tile_rows;
    int64_t cap = onepass_total_wgs() / B;
    if (cap < 1) cap = 1;
    return tiles < cap ? tiles : cap;
}

// Generic reduction over the particles of batch row b = blockIdx.y. Each lane streams its own 28-/56-byte
// rows straight from global memory (measured on MI355X: dword-strided row reads reach the same bandwidth as
// LDS-staged float4 tiles, benchmarks/apply_variants.hip "direct dword"), 2 rows in flight per lane and
// iteration, no barrier inside the loop; one block reduction at the end.
// F::accumulate(x[7], w, n, acc[K]) is called once per particle.
// TRANSPOSED: the K block sums are written by K lanes to partial_out[k * out_stride] (partials laid out [k][block], so
// that the finalize kernel reads them coalesced) after a DPP row reduction + one LDS exchange of 16 row sums.
template <typename T, int K, typename F, bool TRANSPOSED = false>
__device__ __forceinline__ void tiled_reduce(const T* __restrict__ x, const T* __restrict__ w,
                                             int64_t Bx, int64_t Bw, int64_t N, F& f,
                                             double* __restrict__ partial_out /*[K]*/, int64_t out_stride = 1) {
    __shared__ double red[(TRANSPOSED ? 16 : 4) * K];
    const int64_t b = blockIdx.y;
    const int64_t xrow = (Bx == 1) ? 0 : b, wrow = (Bw == 1) ? 0 : b;
    const T* __restrict__ xb = x + xrow * N * 7;
    const T* __restrict__ wb = w ? w + wrow * N : nullptr;
    double acc[K];
#pragma unroll
    for (int k = 0; k < K; ++k) acc[k] = 0.0;
    // every workgroup owns a contiguous range of rows (a multiple of the block size); four rows in flight per lane and
    // iteration, rows past the end of the range predicated off
    const int64_t per = (((N + gridDim.x - 1) / gridDim.x + CHX_BLOCK - 1) / CHX_BLOCK) * CHX_BLOCK;
    const int64_t n0 = (int64_t)blockIdx.x * per;
    const int64_t n1 = (n0 + per < N) ? n0 + per : N;
    for (int64_t n = n0 + threadIdx.x; n < n1; n += 4 * CHX_BLOCK) {
        T r[4][7];
        double wv[4];
        bool ok[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t nn = n + u * CHX_BLOCK;
            ok[u] = nn < n1;
            const int64_t src = ok[u] ? nn : n;
#pragma unroll
            for (int j = 0; j < 7; ++j) r[u][j] = xb[src * 7 + j];
            wv[u] = wb ? (double)wb[src] : 1.0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (!ok[u]) continue;
            double xv[7];
#pragma unroll
            for (int j = 0; j < 7; ++j) xv[j] = (double)r[u][j];
            f.accumulate(xv, wv[u], n + u * CHX_BLOCK, acc);
        }
    }
    if (TRANSPOSED) {
        const int lane = threadIdx.x & 63, row = (threadIdx.x >> 6) * 4 + (lane >> 4);
#pragma unroll
        for (int k = 0; k < K; ++k) acc[k] = chx_row16_sum(acc[k]);
        if ((lane & 15) == 0) {
#pragma unroll
            for (int k = 0; k < K; ++k) red[row * K + k] = acc[k];
        }
        __syncthreads();
        if (threadIdx.x < K) {
            double t = 0.0;
#pragma unroll
            for (int r = 0; r < 16; ++r) t += red[r * K + threadIdx.x];
            partial_out[threadIdx.x * out_stride] = t;
        }
        return;
    }
    chx_block_sum<K>(acc, red);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) partial_out[k] = acc[k];
    }
}

struct SumsFn {
    __device__ __forceinline__ void accumulate(const double (&x)[7], double w, int64_t, double (&a)[8]) {
        a[0] += w;
        a[1] += w * w;
#pragma unroll
        for (int j = 0; j < 6; ++j) a[2 + j] += w * x[j];
    }
};

struct CentredFn {
    double mu[6];
    __device__ __forceinline__ void accumulate(const double (&x)[7], double w, int64_t, double (&a)[21]) {
        double d[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) d[j] = x[j] - mu[j];
        int k = 0;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const double wd = w * d[i];
#pragma unroll
            for (int j = i; j < 6; ++j) a[k++] += wd * d[j];
        }
    }
};

template <typename T>
__global__ __launch_bounds__(CHX_BLOCK) void moment_sums_kernel(const T* __restrict__ x,
                                                               const T* __restrict__ w, int64_t Bx,
                                                               int64_t Bw, int64_t N,
                                                               double* __restrict__ partials) {
    SumsFn f;
    tiled_reduce<T, 8, SumsFn>(x, w, Bx, Bw, N, f,
                               partials + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 8);
}

template <typename T>
__global__ __launch_bounds__(CHX_BLOCK) void moment_centred_kernel(const T* __restrict__ x,
                                                                  const T* __restrict__ w,
                                                                  const double* __restrict__ sums,
                                                                  int64_t Bx, int64_t Bw, int64_t N,
                                                                  double* __restrict__ partials) {
    CentredFn f;
    const double* s = sums + (int64_t)blockIdx.y * CHX_MOM_NSUMS;
    const double W = s[0];
#pragma unroll
    for (int j = 0; j < 6; ++j) f.mu[j] = s[2 + j] / W;
    tiled_reduce<T, 21, CentredFn>(x, w, Bx, Bw, N, f,
                                   partials + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 21);
}

// out[b][k] = sum over blk of partials[b][blk][k]. One wavefront per (k, b): lane-strided partial
// sums in a fixed order, then the wave tree -> deterministic for a given launch geometry.
__global__ __launch_bounds__(64) void reduce_partials_kernel(const double* __restrict__ partials,
                                                            int nblk, int K,
                                                            double* __restrict__ out) {
    const int64_t b = blockIdx.y;
    const int k = blockIdx.x;
    const double* p = partials + b * nblk * K + k;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;  // 4 independent chains hide the load latency
    int i = threadIdx.x;
    for (; i + 192 < nblk; i += 256) {
        s0 += p[(int64_t)i * K];
        s1 += p[(int64_t)(i + 64) * K];
        s2 += p[(int64_t)(i + 128) * K];
        s3 += p[(int64_t)(i + 192) * K];
    }
    for (; i < nblk; i += 64) s0 += p[(int64_t)i * K];
    const double s = chx_wave_sum((s0 + s1) + (s2 + s3));
    if (threadIdx.x == 0) out[b * K + k] = s;
}

__global__ void moment_finalize_kernel(const double* __restrict__ sums, const double* __restrict__ m2,
                                       int64_t B, double* __restrict__ out) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const double* s = sums + b * CHX_MOM_NSUMS;
    const double W = s[0], W2 = s[1];
    double* o = out + b * CHX_MOM_NOUT;
    o[0] = W;
    o[1] = W2;
    for (int j = 0; j < 6; ++j) o[2 + j] = s[2 + j] / W;
    const double cf = W - W2 / W;  // statistics.py:42
    for (int k = 0; k < 21; ++k) o[8 + k] = m2[b * 21 + k] / cf;
}

// dX[n][a] = w_n ( dmu_a / W + (1/cf) sum_b Gsym[a][b] (x_b - mu_b) ), Gsym = g + g^T on the
// upper-triangular cotangent g of the covariances. Column 6 gets 0.
template <typename T>
__global__ __launch_bounds__(CHX_BLOCK) void moments_bwd_kernel(const T* __restrict__ x,
                                                               const T* __restrict__ w,
                                                               const double* __restrict__ out,
                                                               const double* __restrict__ d_out,
                                                               int64_t Bx, int64_t Bw, int64_t N,
                                                               T* __restrict__ dX, T* __restrict__ dWt) {
    constexpr int PPT = red_cfg<T>::PPT;
    constexpr int TP = PPT * CHX_BLOCK;
    __shared__ __attribute__((aligned(16))) T lds[TP * 7];
    const int64_t b = blockIdx.y;
    const int64_t xrow = (Bx == 1) ? 0 : b, wrow = (Bw == 1) ? 0 : b;
    const double* o = out + b * CHX_MOM_NOUT;
    const double* g = d_out + b * CHX_MOM_NOUT;
    const double W = o[0], W2 = o[1], icf = 1.0 / (W - W2 / W);
    double G[6][6];
    {
        int k = 0;
        for (int i = 0; i < 6; ++i)
            for (int j = i; j < 6; ++j) {
                const double v = g[8 + k++];
                if (i == j) G[i][i] = 2.0 * v;
                else { G[i][j] = v; G[j][i] = v; }
            }
    }
    // weights: with S = sum_{a<=b} g_ab cov_ab and k = 1 + W2 / W^2,
    //   dW[n] = g_W + 2 w_n g_W2 + (g_mu . d_n) / W + (1/cf) ( d_n^T Gsym d_n / 2 - S (k - 2 w_n / W) )
    // (dM_ab / dw_n = d_na d_nb: the shift of the mean drops out because sum_m w_m d_m = 0; d cf / d w_n = k - 2 w_n / W)
    double S = 0.0;
    for (int k = 0; k < 21; ++k) S += g[8 + k] * o[8 + k];
    const double kcf = 1.0 + W2 / (W * W);
    const int64_t n0 = (int64_t)blockIdx.x * TP;
    const int np = (int)((N - n0 < TP) ? (N - n0) : TP);
    const bool vin = chx_aligned16(x) && (((xrow * N * 7 * (int64_t)sizeof(T)) & 15) == 0);
    const bool vout = dX && chx_aligned16(dX) && (((b * N * 7 * (int64_t)sizeof(T)) & 15) == 0);
    tile_load<T, TP>(x + (xrow * N + n0) * 7, lds, np * 7, vin);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const int p = threadIdx.x + k * CHX_BLOCK;
        if (p < np) {
            double d[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) d[j] = (double)lds[p * 7 + j] - o[2 + j];
            const double wv = w ? (double)w[wrow * N + n0 + p] : 1.0;
            double quad = 0.0, lin = 0.0;
#pragma unroll
            for (int a = 0; a < 6; ++a) {
                double s = 0.0;
#pragma unroll
                for (int c = 0; c < 6; ++c) s += G[a][c] * d[c];
                quad += d[a] * s;
                lin += g[2 + a] * d[a];
                lds[p * 7 + a] = (T)(wv * (g[2 + a] / W + icf * s));
            }
            lds[p * 7 + 6] = (T)0;
            if (dWt) dWt[b * N + n0 + p] = (T)(g[0] + 2.0 * wv * g[1] + lin / W + icf * (0.5 * quad - S * (kcf - 2.0 * wv / W)));
        }
    }
    __syncthreads();
    if (dX) tile_store<T, TP>(dX + (b * N + n0) * 7, lds, np * 7, vout);
}

// dR[i][j] = sum_n dY[n][i] X[n][j]   (49 fp64 accumulators per lane)
// Same shape as the one-pass moments kernel: every workgroup owns a contiguous range of rows, each lane streams its rows of
// BOTH arrays straight from global memory (two rows of each in flight, no barrier in the loop), then a DPP sum over each row
// of 16 lanes, one LDS exchange of the 16 row sums, and 49 lanes write the workgroup's sums to partials[b][k][block]
// (coalesced for reduce_partials_t_kernel). The LDS-staged, 256-workgroup form it replaces took 37.9 us for 1e6 rows.
template <typename T>
__global__ __launch_bounds__(CHX_BLOCK) void apply_bwd_dR_kernel(const T* __restrict__ dY,
                                                                const T* __restrict__ X, int64_t Bx,
                                                                int64_t N,
                                                                double* __restrict__ partials) {
    __shared__ double red[16 * 49];
    const int64_t b = blockIdx.y;
    const T* __restrict__ xb = X + ((Bx == 1) ? 0 : b) * N * 7;
    const T* __restrict__ yb = dY + b * N * 7;
    double acc[49];
#pragma unroll
    for (int k = 0; k < 49; ++k) acc[k] = 0.0;
    const int64_t per = (((N + gridDim.x - 1) / gridDim.x + CHX_BLOCK - 1) / CHX_BLOCK) * CHX_BLOCK;
    const int64_t n0 = (int64_t)blockIdx.x * per;
    const int64_t n1 = (n0 + per < N) ? n0 + per : N;
    for (int64_t n = n0 + threadIdx.x; n < n1; n += 2 * CHX_BLOCK) {
        T rx[2][7], ry[2][7];
        bool ok[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int64_t nn = n + u * CHX_BLOCK;
            ok[u] = nn < n1;
            const int64_t src = ok[u] ? nn : n;
#pragma unroll
            for (int j = 0; j < 7; ++j) { rx[u][j] = xb[src * 7 + j]; ry[u][j] = yb[src * 7 + j]; }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (!ok[u]) continue;
#pragma unroll
            for (int i = 0; i < 7; ++i) {
                const double yv = (double)ry[u][i];
#pragma unroll
                for (int j = 0; j < 7; ++j) acc[i * 7 + j] = fma(yv, (double)rx[u][j], acc[i * 7 + j]);
            }
        }
    }
    const int lane = threadIdx.x & 63, row = (threadIdx.x >> 6) * 4 + (lane >> 4);
#pragma unroll
    for (int k = 0; k < 49; ++k) acc[k] = chx_row16_sum(acc[k]);
    if ((lane & 15) == 0) {
#pragma unroll
        for (int k = 0; k < 49; ++k) red[row * 49 + k] = acc[k];
    }
    __syncthreads();
    if (threadIdx.x < 49) {
        double t = 0.0;
#pragma unroll
        for (int r = 0; r < 16; ++r) t += red[r * 49 + threadIdx.x];
        partials[((int64_t)b * 49 + threadIdx.x) * gridDim.x + blockIdx.x] = t;
    }
}

// out[b][k] = sum over blk of partials[b][k][blk] (the transposed layout: contiguous reads). One wavefront per (k, b).
__global__ __launch_bounds__(64) void reduce_partials_t_kernel(const double* __restrict__ partials, int nblk, int K,
                                                              double* __restrict__ out) {
    const int64_t b = blockIdx.y;
    const int k = blockIdx.x;
    const double* p = partials + (b * K + k) * (int64_t)nblk;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int i = threadIdx.x;
    for (; i + 192 < nblk; i += 256) {
        s0 += p[i];
        s1 += p[i + 64];
        s2 += p[i + 128];
        s3 += p[i + 192];
    }
    for (; i < nblk; i += 64) s0 += p[i];
    const double s = chx_wave_sum((s0 + s1) + (s2 + s3));
    if (threadIdx.x == 0) out[b * K + k] = s;
}

template <typename T>
__global__ void transpose_maps_kernel(const T* __restrict__ R, int64_t B, T* __restrict__ Rt) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * 49) return;
    const int64_t b = idx / 49;
    const int k = (int)(idx - b * 49), i = k / 7, j = k - 7 * i;
    Rt[b * 49 + j * 7 + i] = R[idx];
}

// ---- fused observables: moments of R_b x without writing the tracked particles --------------------------------
// (element.py:180-191 followed by particle_beam.py:1672-1943, e.g. a k1 scan that only
// reads sigma_x / sigma_y of the outgoing beams.) Shifted single pass: with d = R_b x - c_b,
//   acc = { sum w, sum w^2, sum w d_a (6), sum w d_a d_b (21, a<=b) }
// where c_b = R_b centre is the image of a point near the incoming mean, so |mean(d)| << sigma and the one-pass
// second moments do not cancel. The apply runs in the storage dtype with the same fma chain as chx_apply_affine7
// (bit-identical y), the accumulation in fp64.
constexpr int kTM = 29;

template <typename T> __device__ __forceinline__ T tm_fma(T a, T b, T c);
template <> __device__ __forceinline__ float tm_fma<float>(float a, float b, float c) { return fmaf(a, b, c); }
template <> __device__ __forceinline__ double tm_fma<double>(double a, double b, double c) { return fma(a, b, c); }

template <typename T>
__device__ __forceinline__ void tm_accumulate(const T (&R)[42], const T (&x)[7], double w, const double (&c)[6],
                                              double (&a)[kTM]) {
    double d[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        T y = R[i * 7] * x[0];
#pragma unroll
        for (int j = 1; j < 7; ++j) y = tm_fma<T>(R[i * 7 + j], x[j], y);
        d[i] = (double)y - c[i];
    }
    a[0] += w;
    a[1] += w * w;
    int k = 8;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const double wd = w * d[i];
        a[2 + i] += wd;
#pragma unroll
        for (int j = i; j < 6; ++j) a[k++] = fma(wd, d[j], a[k]);
    }
}

// c = R centre (fp64; centre[6] with the affine 1 appended); NULL centre -> no shift
template <typename T>
__device__ __forceinline__ void tm_centre(const T (&R)[42], const double* __restrict__ centre, double (&c)[6]) {
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        double s = 0.0;
        if (centre) {
            s = (double)R[i * 7 + 6];
#pragma unroll
            for (int j = 0; j < 6; ++j) s += (double)R[i * 7 + j] * centre[j];
        }
        // float32 beams: a float32 shift point (any point near the mean serves), so that kernels which subtract it in float32
        // and the finalize kernel, which adds it back in float64, mean the same number
        c[i] = sizeof(T) == 4 ? (double)(float)s : s;
    }
}

// Shared input beam, many maps: ONE LANE PER BATCH ROW. The lane keeps its map (42 values) and its 29 fp64
// accumulators in registers and walks over a chunk of particles staged in LDS; every lane reads the same LDS
// address (broadcast, conflict-free), and there is no cross-lane reduction at all.
template <typename T> struct tm_cfg { static constexpr int chunk = sizeof(T) == 4 ? 1024 : 512; };  // 28 KiB of LDS

template <typename T>
__global__ __launch_bounds__(CHX_BLOCK) void track_moments_rows_kernel(const T* __restrict__ x, const T* __restrict__ w,
                                                                      const T* __restrict__ Rm,
                                                                      const double* __restrict__ centre, int64_t B,
                                                                      int64_t N, int64_t per_chunk,
                                                                      double* __restrict__ partials) {
    constexpr int kTMChunk = tm_cfg<T>::chunk;
    __shared__ __attribute__((aligned(16))) T xs[kTMChunk * 7];
    __shared__ T ws[kTMChunk];
    const int64_t b = (int64_t)blockIdx.x * CHX_BLOCK + threadIdx.x;
    const int64_t bb = b < B ? b : B - 1;
    T R[42];
#pragma unroll
    for (int k = 0; k < 42; ++k) R[k] = Rm[bb * 49 + k];
    double c[6], acc[kTM];
    tm_centre<T>(R, centre, c);
#pragma unroll
    for (int k = 0; k < kTM; ++k) acc[k] = 0.0;
    const int64_t n_begin = (int64_t)blockIdx.y * per_chunk;
    const int64_t n_end = (n_begin + per_chunk < N) ? n_begin + per_chunk : N;
    for (int64_t n0 = n_begin; n0 < n_end; n0 += kTMChunk) {
        const int np = (int)((n_end - n0 < kTMChunk) ? (n_end - n0) : kTMChunk);
        __syncthreads();
        tile_load<T, kTMChunk>(x + n0 * 7, xs, np * 7, chx_aligned16(x) && ((n0 * 7 * (int64_t)sizeof(T)) & 15) == 0);
        for (int i = threadIdx.x; i < np; i += CHX_BLOCK) ws[i] = w ? w[n0 + i] : (T)1;
        __syncthreads();
        for (int i = 0; i < np; ++i) {
            T xv[7];
#pragma unroll
            for (int j = 0; j < 7; ++j) xv[j] = xs[i * 7 + j];
            tm_accumulate<T>(R, xv, (double)ws[i], c, acc);
        }
    }
    if (b < B) {
        double* p = partials + ((int64_t)blockIdx.y * B + b) * kTM;
#pragma unroll
        for (int k = 0; k < kTM; ++k) p[k] = acc[k];
    }
}

// The same for float32 beams with the products in PACKED float32 and float64 only for the running sums: a lane handles two
// particles at a time (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 on {particle 2i, particle 2i + 1}), keeps 29 packed float32
// partial sums over kTMFlush pairs and adds them to its float64 accumulators once per flush. The all-float64 kernel above is
// fp64-VALU bound (41 float64 operations per (setting, particle) behind the 42 float32 FMAs of the map: 1.0 ms for
// 4096 x 1e5); here a pair costs ~83 packed operations and 3.6 float64 operations per particle.
// Rounding: y is the float32 the apply kernels store; d = y - c in float32 (c rounded to float32, see tm_centre) and
// w d_i d_j in float32 carry ~6e-8 relative error each, independent from particle to particle; 32 of them are summed in
// float32 before the float64 add (kTMFlush pairs) — the moments agree with the float64 accumulation to ~1e-7 relative
// (tests/test_gpu_track_moments.py, profiles/r03_track_moments_rows.md).
constexpr int kTMFlush = 64;   // pairs between two float64 updates

__global__ __launch_bounds__(CHX_BLOCK) void track_moments_rows_f32_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                          const float* __restrict__ Rm,
                                                                          const double* __restrict__ centre, int64_t B, int64_t N,
                                                                          int64_t per_chunk, double* __restrict__ partials) {
    constexpr int kChunk = 1024, kPairs = kChunk / 2;
    __shared__ __attribute__((aligned(16))) chx_v2f xs[kPairs * 7];   // [pair][coordinate] = {x of the even, x of the odd particle}
    __shared__ __attribute__((aligned(8))) chx_v2f ws[kPairs];
    const int64_t b = (int64_t)blockIdx.x * CHX_BLOCK + threadIdx.x;
    const int64_t bb = b < B ? b : B - 1;
    float R[42];
#pragma unroll
    for (int k = 0; k < 42; ++k) R[k] = Rm[bb * 49 + k];
    double c[6], acc[kTM];
    tm_centre<float>(R, centre, c);
    float cf[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) cf[i] = (float)c[i];               // exact: tm_centre<float> returns float32 values
#pragma unroll
    for (int k = 0; k < kTM; ++k) acc[k] = 0.0;
    const int64_t n_begin = (int64_t)blockIdx.y * per_chunk;
    const int64_t n_end = (n_begin + per_chunk < N) ? n_begin + per_chunk : N;
    for (int64_t n0 = n_begin; n0 < n_end; n0 += kChunk) {
        const int np = (int)((n_end - n0 < kChunk) ? (n_end - n0) : kChunk);
        const int npairs = (np + 1) / 2;
        __syncthreads();
        // stage the chunk pairwise; a missing odd partner gets weight 0 (and finite coordinates)
        float* xf = reinterpret_cast<float*>(xs);
        for (int e = threadIdx.x; e < npairs * 14; e += CHX_BLOCK) {
            const int pr = e / 14, r = e - pr * 14, j = r >> 1, odd = r & 1;
            const int i = 2 * pr + odd;
            xf[e] = i < np ? x[(n0 + i) * 7 + j] : 0.0f;
        }
        float* wf = reinterpret_cast<float*>(ws);
        for (int i = threadIdx.x; i < npairs * 2; i += CHX_BLOCK) wf[i] = i < np ? (w ? w[n0 + i] : 1.0f) : 0.0f;
        __syncthreads();
        for (int p0 = 0; p0 < npairs; p0 += kTMFlush) {
            const int p1 = p0 + kTMFlush < npairs ? p0 + kTMFlush : npairs;
            chx_v2f a[kTM];
#pragma unroll
            for (int k = 0; k < kTM; ++k) a[k] = chx_v2f{0.0f, 0.0f};
            for (int p = p0; p < p1; ++p) {
                chx_v2f xv[7];
#pragma unroll
                for (int j = 0; j < 7; ++j) xv[j] = xs[p * 7 + j];
                const chx_v2f wv = ws[p];
                chx_v2f d[6];
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    chx_v2f y = xv[0] * R[i * 7];                  // the fma chain of chx_apply_affine7, two particles wide
#pragma unroll
                    for (int j = 1; j < 7; ++j) {
                        const chx_v2f r = {R[i * 7 + j], R[i * 7 + j]};
                        y = __builtin_elementwise_fma(r, xv[j], y);
                    }
                    d[i] = y - chx_v2f{cf[i], cf[i]};
                }
                a[0] = a[0] + wv;
                a[1] = __builtin_elementwise_fma(wv, wv, a[1]);
                int k = 8;
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    const chx_v2f wd = wv * d[i];
                    a[2 + i] = a[2 + i] + wd;
#pragma unroll
                    for (int j = i; j < 6; ++j, ++k) a[k] = __builtin_elementwise_fma(wd, d[j], a[k]);
                }
            }
#pragma unroll
            for (int k = 0; k < kTM; ++k) acc[k] += (double)a[k].x + (double)a[k].y;
        }
    }
    if (b < B) {
        double* p = partials + ((int64_t)blockIdx.y * B + b) * kTM;
#pragma unroll
        for (int k = 0; k < kTM; ++k) p[k] = acc[k];
    }
}

// General case (per-row inputs or few rows): lane per particle, block reduction per workgroup (tiled_reduce).
template <typename T>
struct TrackMomFn {
    T R[42];
    double c[6];
    __device__ __forceinline__ void accumulate(const double (&xd)[7], double w, int64_t, double (&a)[kTM]) {
        T xv[7];
#pragma unroll
        for (int j = 0; j < 7; ++j) xv[j] = (T)xd[j];  // exact: xd came from T
        tm_accumulate<T>(R, xv, w, c, a);
    }
};

template <typename T>
__global__ __launch_bounds__(CHX_BLOCK) void track_moments_particles_kernel(const T* __restrict__ x, const T* __restrict__ w,
                                                                           const T* __restrict__ Rm,
                                                                           const double* __restrict__ centre, int64_t Bx,
                                                                           int64_t BR, int64_t Bw, int64_t N,
                                                                           double* __restrict__ partials) {
    TrackMomFn<T> f;
    const int64_t b = blockIdx.y;
    const T* Rb = Rm + ((BR == 1) ? 0 : b) * 49;
#pragma unroll
    for (int k = 0; k < 42; ++k) f.R[k] = Rb[k];
    tm_centre<T>(f.R, centre ? centre + ((Bx == 1) ? 0 : b) * 6 : nullptr, f.c);
    tiled_reduce<T, kTM, TrackMomFn<T>>(x, w, Bx, Bw, N, f, partials + ((int64_t)blockIdx.x * gridDim.y + b) * kTM);
}

// out[b] from partials[chunk][b][29]: sum the chunks in order, undo the shift, normalise (statistics.py:41-46)
template <typename T>
__global__ void track_moments_finalize_kernel(const double* __restrict__ partials, int nchunk, const T* __restrict__ Rm,
                                              const double* __restrict__ centre, int64_t B, int64_t Bx, int64_t BR,
                                              double* __restrict__ out) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double a[kTM];
    for (int k = 0; k < kTM; ++k) a[k] = 0.0;
    for (int ch = 0; ch < nchunk; ++ch)
        for (int k = 0; k < kTM; ++k) a[k] += partials[((int64_t)ch * B + b) * kTM + k];
    T R[42];
    for (int k = 0; k < 42; ++k) R[k] = Rm[((BR == 1) ? 0 : b) * 49 + k];
    double c[6];
    tm_centre<T>(R, centre ? centre + ((Bx == 1) ? 0 : b) * 6 : nullptr, c);
    const double W = a[0], W2 = a[1];
    double* o = out + b * CHX_MOM_NOUT;
    o[0] = W;
    o[1] = W2;
    double m[6];
    for (int j = 0; j < 6; ++j) { m[j] = a[2 + j] / W; o[2 + j] = c[j] + m[j]; }
    const double cf = W - W2 / W;
    int k = 8;
    for (int i = 0; i < 6; ++i)
        for (int j = i; j < 6; ++j, ++k) o[k] = (a[k] - W * m[i] * m[j]) / cf;
}

// ---- one-pass moments (chx_moments): same statistics as the two-pass pair above from ONE sweep over the particles.
// Second moments are accumulated about c = the first particle of the row (a point of the distribution, so
// |mean - c| is a few sigma at most and sum w d d^T does not cancel), then re-centred exactly:
//   mu = c + s / W,  M = m - s s^T / W,  cov = M / (W - W2 / W).
struct OnePassFn {
    double c[6];
    __device__ __forceinline__ void accumulate(const double (&x)[7], double w, int64_t, double (&a)[kTM]) {
        double d[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) d[j] = x[j] - c[j];
        a[0] += w;
        a[1] += w * w;
        int k = 8;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const double wd = w * d[i];
            a[2 + i] += wd;
#pragma unroll
            for (int j = i; j < 6; ++j) { a[k] = fma(wd, d[j], a[k]); ++k; }
        }
    }
};

// The provisional centre of the one-pass sums: the weighted mean of the row's first 16 particles, not particle 0 alone — one
// far outlier in slot 0 (|x0 - mu| = r sigma costs r^2 of the fp64 headroom of sum w d d^T) no longer matters, and a zero-weight
// one does not enter the centre at all. Every wave forms the same c from the same 16 rows (448 bytes: cache hits) under its own
// row requests: 16 lanes load, four DPP steps sum the row of lanes, one reciprocal, v_readfirstlane hands the result to the wave
// (the 64-row / ds_bpermute / six-division form of this cost 2.5 us of the 9 us sweep). No weight in those rows (or a non-finite
// mean, e.g. 0 * inf of a dead outlier): particle 0 — any finite centre is exact algebra, only the headroom differs.
template <typename T>
__device__ __forceinline__ void first_wave_centre(const T* __restrict__ xb, const T* __restrict__ wb, int64_t N, double (&c)[6]) {
    const int lane = threadIdx.x & 63;
    const bool ok = lane < 16 && lane < N;
    double wv = 0.0, xv[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) xv[j] = ok ? (double)xb[(int64_t)lane * 7 + j] : 0.0;
    if (ok) wv = wb ? (double)wb[lane] : 1.0;
    const double W = chx_row16_sum(wv);
    const double inv = 1.0 / W;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const double m = chx_row16_sum(wv * xv[j]) * inv;
        const double pick = (W > 0.0 && isfinite(m)) ? m : xv[j];       // lane 0: row 0
        c[j] = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(pick)), __builtin_amdgcn_readfirstlane(__double2loint(pick)));
    }
}

// The sweep: every lane streams its own rows straight from global memory (dword-strided row reads), kMomUnroll rows in flight per
// lane and iteration; with 512 workgroups a lane of the 1e6-particle case holds all its 8 rows in ONE batch of requests, i.e. the
// kernel is a single memory round trip long. (Measured and dropped, round 4: per-wave LDS staging with 14 KB of 16-byte loads in
// flight per wave — 12.0 us against 9.3 for this form at 1e6 particles; the sweep is a latency chain, not a bandwidth problem,
// and the LDS detour lengthens the chain.) partials[b][k][blk]; centre_out[b][6] (workgroup 0).
constexpr int kMomUnroll = 8;
template <typename T>
__global__ __launch_bounds__(CHX_BLOCK) void moments_onepass_kernel(const T* __restrict__ x, const T* __restrict__ w,
                                                                   int64_t Bx, int64_t Bw, int64_t N,
                                                                   double* __restrict__ partials, double* __restrict__ centre_out) {
    constexpr int U = kMomUnroll;
    __shared__ double red[16 * kTM];
    const int64_t b = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const T* __restrict__ xb = x + ((Bx == 1) ? 0 : b) * N * 7;
    const T* __restrict__ wb = w ? w + ((Bw == 1) ? 0 : b) * N : nullptr;
    const int64_t per = (((N + gridDim.x - 1) / gridDim.x + CHX_BLOCK - 1) / CHX_BLOCK) * CHX_BLOCK;
    const int64_t n0 = (int64_t)blockIdx.x * per;
    const int64_t n1 = (n0 + per < N) ? n0 + per : N;
    OnePassFn f;
    double acc[kTM];
#pragma unroll
    for (int k = 0; k < kTM; ++k) acc[k] = 0.0;
    bool have_c = false;
    // (the loop bounds are the same for every lane of the workgroup: the DPP sums of first_wave_centre need whole rows of lanes)
    for (int64_t nb = n0; nb < n1; nb += U * CHX_BLOCK) {
        const int64_t n = nb + threadIdx.x;
        T r[U][7];
        double wv[U];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t nn = n + u * CHX_BLOCK;
            ok[u] = nn < n1;
            const int64_t src = ok[u] ? nn : nb;
#pragma unroll
            for (int j = 0; j < 7; ++j) r[u][j] = xb[src * 7 + j];
            wv[u] = wb ? (double)wb[src] : 1.0;
        }
        if (!have_c) {              // (formed under the row requests above)
            first_wave_centre<T>(xb, wb, N, f.c);
            have_c = true;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!ok[u]) continue;
            double xv[7];
#pragma unroll
            for (int j = 0; j < 7; ++j) xv[j] = (double)r[u][j];
            f.accumulate(xv, wv[u], n + u * CHX_BLOCK, acc);
        }
    }
    if (!have_c) first_wave_centre<T>(xb, wb, N, f.c);      // (a lane without rows still takes part in the reduction)
    if (blockIdx.x == 0 && threadIdx.x == 0) {
#pragma unroll
        for (int j = 0; j < 6; ++j) centre_out[b * 6 + j] = f.c[j];
    }
    const int row = wave * 4 + (lane >> 4);
#pragma unroll
    for (int k = 0; k < kTM; ++k) acc[k] = chx_row16_sum(acc[k]);
    if ((lane & 15) == 0) {
#pragma unroll
        for (int k = 0; k < kTM; ++k) red[row * kTM + k] = acc[k];
    }
    __syncthreads();
    if (threadIdx.x < kTM) {
        double t = 0.0;
#pragma unroll
        for (int r2 = 0; r2 < 16; ++r2) t += red[r2 * kTM + threadIdx.x];
        partials[(b * kTM + threadIdx.x) * (int64_t)gridDim.x + blockIdx.x] = t;
    }
}

// Many SHORT rows (a vectorised beam: 4096 beams of 1000 particles): one WAVE per batch row does the whole of chx_moments — the
// sweep (rows lane, lane + 64, ...; four in flight per lane), the wave-wide sums and the finalize arithmetic of
// moments_reduce_finalize_kernel on lanes 0..28 — in one launch without partial sums. The workgroup form above holds 29
// accumulators and eight rows per lane (~200 VGPRs: two workgroups per CU) and needs eight rounds of workgroups plus a second
// launch for 4096 rows: 47 + 6 us against ~10 us here. Same statistics, the fp64 sums in another order.
constexpr int kRowWaveUnroll = 4;
template <typename T>
__global__ __launch_bounds__(CHX_BLOCK) void moments_rows_wave_kernel(const T* __restrict__ x, const T* __restrict__ w, int64_t B,
                                                                     int64_t Bx, int64_t Bw, int64_t N, double* __restrict__ out,
                                                                     int entry, int entry_sqrt, T* __restrict__ entry_out) {
    constexpr int U = kRowWaveUnroll;
    const int lane = threadIdx.x & 63;
    const int64_t b = (int64_t)blockIdx.x * (CHX_BLOCK / 64) + (threadIdx.x >> 6);
    if (b >= B) return;                                    // (whole waves: nothing below synchronises the workgroup)
    const T* __restrict__ xb = x + ((Bx == 1) ? 0 : b) * N * 7;
    const T* __restrict__ wb = w ? w + ((Bw == 1) ? 0 : b) * N : nullptr;
    OnePassFn f;
    double acc[kTM];
#pragma unroll
    for (int k = 0; k < kTM; ++k) acc[k] = 0.0;
    bool have_c = false;
    for (int64_t nb = 0; nb < N; nb += U * 64) {
        T r[U][7];
        double wv[U];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t nn = nb + lane + u * 64;
            ok[u] = nn < N;
            const int64_t src = ok[u] ? nn : nb;
#pragma unroll
            for (int j = 0; j < 7; ++j) r[u][j] = xb[src * 7 + j];
            wv[u] = wb ? (double)wb[src] : 1.0;
        }
        if (!have_c) {
            first_wave_centre<T>(xb, wb, N, f.c);
            have_c = true;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!ok[u]) continue;
            double xv[7];
#pragma unroll
            for (int j = 0; j < 7; ++j) xv[j] = (double)r[u][j];
            f.accumulate(xv, wv[u], nb + lane + u * 64, acc);
        }
    }
#pragma unroll
    for (int k = 0; k < kTM; ++k) acc[k] = chx_wave_sum(acc[k]);
    if (lane < kTM) {
        // lane k forms out[b][k] (the arithmetic of moments_reduce_finalize_kernel); operands picked without indexing registers
        const int k = lane;
        int i = 0, j = 0;
        if (k >= 8) {
            int rem = k - 8;
            while (rem >= 6 - i) { rem -= 6 - i; ++i; }
            j = i + rem;
        }
        double tk = 0.0, si = 0.0, sj = 0.0, ck = 0.0;
#pragma unroll
        for (int q = 0; q < kTM; ++q)
            if (q == k) tk = acc[q];
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            if (q == i) si = acc[2 + q];
            if (q == j) sj = acc[2 + q];
            if (q == k - 2) ck = f.c[q];
        }
        const double W = acc[0], W2 = acc[1];
        double v;
        if (k == 0) v = W;
        else if (k == 1) v = W2;
        else if (k < 8) v = ck + tk / W;
        else {
            const double mi = si / W, mj = sj / W;
            v = (tk - W * mi * mj) / (W - W2 / W);
        }
        out[b * CHX_MOM_NOUT + k] = v;
        if (k == entry) entry_out[b] = (T)(entry_sqrt ? sqrt(v) : v);
    }
}

// One workgroup per batch row: lane t holds partial block t of all 29 sums (partials[b][k][blk]: coalesced loads, all in
// flight at once), DPP row sums, one LDS exchange of the 64 row sums, then lanes 0..28 add them in a fixed order and lane 0
// re-centres and normalises -> out[b][29]. Replaces reduce_partials x2 + finalize of the two-pass path.
// TH = threads of the workgroup: 1024 for long rows; a row with at most 256 / 64 partial blocks (a vectorised beam of many
// short rows: 4096 x 1000 particles) takes 256 / 64 — the lanes beyond the blocks only added zeros, so the sums are the same
// bits, and a thousand-thread workgroup per row of ONE block cost 42 us where 64 threads cost 6.
template <typename T, int TH>
__global__ __launch_bounds__(TH) void moments_reduce_finalize_kernel(const double* __restrict__ partials, int nblk,
                                                                      const double* __restrict__ centre,
                                                                      double* __restrict__ out, int entry /*-1: none*/,
                                                                      int entry_sqrt, T* __restrict__ entry_out) {
    __shared__ double red[64 * kTM];
    __shared__ double tot[32];
    const int64_t b = blockIdx.x;
    const double* pb = partials + b * kTM * (int64_t)nblk;
    double a[kTM];
#pragma unroll
    for (int k = 0; k < kTM; ++k) a[k] = 0.0;
    for (int i = threadIdx.x; i < nblk; i += TH) {
#pragma unroll
        for (int k = 0; k < kTM; ++k) a[k] += pb[(int64_t)k * nblk + i];
    }
    const int rows = (nblk + 15) / 16 < TH / 16 ? (nblk + 15) / 16 : TH / 16;
    if ((int)(threadIdx.x >> 4) < rows) {       // (whole waves beyond the partial blocks hold zeros: nothing to reduce)
#pragma unroll
        for (int k = 0; k < kTM; ++k) a[k] = chx_row16_sum(a[k]);
        if ((threadIdx.x & 15) == 0) {
            const int row = threadIdx.x >> 4;
#pragma unroll
            for (int k = 0; k < kTM; ++k) red[row * kTM + k] = a[k];
        }
    }
    __syncthreads();
    if (threadIdx.x < kTM) {
        double t = 0.0;
        for (int r = 0; r < rows; ++r) t += red[r * kTM + threadIdx.x];
        tot[threadIdx.x] = t;
    }
    __syncthreads();
    // out[b][k], one lane per entry (the 27 divisions side by side instead of one after the other on lane 0)
    if (threadIdx.x < kTM) {
        const int k = threadIdx.x;
        const double W = tot[0], W2 = tot[1];
        double v;
        if (k == 0) v = W;
        else if (k == 1) v = W2;
        else if (k < 8) v = centre[b * 6 + (k - 2)] + tot[k] / W;
        else {
            int i = 0, rem = k - 8;
            while (rem >= 6 - i) { rem -= 6 - i; ++i; }
            const int j = i + rem;
            const double mi = tot[2 + i] / W, mj = tot[2 + j] / W;
            v = (tot[k] - W * mi * mj) / (W - W2 / W);
        }
        out[b * CHX_MOM_NOUT + k] = v;
        if (k == entry) entry_out[b] = (T)(entry_sqrt ? sqrt(v) : v);   // chx_moments_entry: the one beam property the caller reads
    }
}

// ---- beam sizes for the space-charge grid (chx_sc_beam_geometry): the three variances SpaceChargeKick needs
// (space_charge_kick.py:531-538: sigma_x, sigma_y, sigma_tau) with 8 accumulators instead of the 29 of chx_moments, and the
// reduce / finalize step folded into the geometry kernel — two launches per kick instead of three, a third of the registers.
constexpr int kSG = 8;   // W, W2, s_x, s_y, s_tau, m_xx, m_yy, m_tautau (shifted about the row's first particle)
struct SigmaFn {
    double c[3];
    __device__ __forceinline__ void accumulate(const double (&x)[7], double w, int64_t, double (&a)[kSG]) {
        const double d0 = x[0] - c[0], d1 = x[2] - c[1], d2 = x[4] - c[2];
        a[0] += w;
        a[1] += w * w;
        const double w0 = w * d0, w1 = w * d1, w2 = w * d2;
        a[2] += w0;
        a[3] += w1;
        a[4] += w2;
        a[5] = fma(w0, d0, a[5]);
        a[6] = fma(w1, d1, a[6]);
        a[7] = fma(w2, d2, a[7]);
    }
};

template <typename T>
__global__ __launch_bounds__(CHX_BLOCK) void sc_sigma_kernel(const T* __restrict__ x, const T* __restrict__ w, int64_t Bx,
                                                            int64_t Bw, int64_t N, double* __restrict__ partials,
                                                            const int* __restrict__ tile_hdr) {
    // chain of tile-ordered kicks (chx_sc_tiles.h): `w` is copy 0 of the ordered weights, the copy in force is
    // header.parity ^ header.scatter_now (the flip itself happens in the geometry kernel behind this one)
    if (tile_hdr && ((tile_hdr[0] ^ tile_hdr[1]) & 1)) w += N;
    SigmaFn f;
    const T* x0 = x + ((Bx == 1) ? 0 : (int64_t)blockIdx.y) * N * 7;
    f.c[0] = (double)x0[0];
    f.c[1] = (double)x0[2];
    f.c[2] = (double)x0[4];
    // partials[b][k][blk]
    tiled_reduce<T, kSG, SigmaFn, true>(x, w, Bx, Bw, N, f, partials + (int64_t)blockIdx.y * kSG * gridDim.x + blockIdx.x,
                                        gridDim.x);
}

// one workgroup per batch row: sum the partial blocks (lane t holds block t, t + 256, ...), then lane 0 forms the variances
// exactly like moments_reduce_finalize_kernel and writes the grid geometry
// (TH = 1024 for the chain's ~4000 partial blocks: every lane's loads are in flight at once, the kernel is ONE memory round trip long
// instead of four)
template <typename T, int TH = CHX_BLOCK>
__global__ __launch_bounds__(TH) void sc_geometry_partials_kernel(
    const double* __restrict__ partials, int nblk, const T* __restrict__ ext, const T* __restrict__ energy,
    const T* __restrict__ length, double mass, double pot_factor, int64_t Bext, int64_t Be, int64_t Bl, int gx, int gy, int gz,
    T* __restrict__ half, T* __restrict__ cell, T* __restrict__ gamma_out, T* __restrict__ dt, T* __restrict__ scale,
    T* __restrict__ extent, double* __restrict__ pot_scale, int* __restrict__ tile_hdr, int tile_first) {
    __shared__ double red[(TH / 16) * kSG];
    // chain of tile-ordered kicks (chx_sc_tiles.h): this single-workgroup kernel runs before the deposit / gather kernels of the
    // kick, so it is where the header rolls over: header = {parity, scatter_now, ...}
    if (tile_hdr && !tile_first && blockIdx.x == 0 && threadIdx.x == 0) {
        if (tile_hdr[1]) {          // the previous kick's gather wrote its rows in a new tile order: its arrays are in force now
            tile_hdr[0] ^= 1;
            tile_hdr[1] = 0;
        }
    }
    __shared__ double tot[kSG];
    const int64_t b = blockIdx.x;
    const double* pb = partials + b * kSG * (int64_t)nblk;
    double a[kSG];
#pragma unroll
    for (int k = 0; k < kSG; ++k) a[k] = 0.0;
    // four blocks per lane and step, all 32 loads in flight together (one workgroup: the kernel is a chain of load latencies)
    for (int i0 = threadIdx.x; i0 < nblk; i0 += 4 * TH) {
        double v[4][kSG];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * TH;
#pragma unroll
            for (int k = 0; k < kSG; ++k) v[u][k] = i < nblk ? pb[(int64_t)k * nblk + i] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int k = 0; k < kSG; ++k) a[k] += v[u][k];
    }
#pragma unroll
    for (int k = 0; k < kSG; ++k) a[k] = chx_row16_sum(a[k]);
    if ((threadIdx.x & 15) == 0) {
#pragma unroll
        for (int k = 0; k < kSG; ++k) red[(threadIdx.x >> 4) * kSG + k] = a[k];
    }
    __syncthreads();
    if (threadIdx.x < kSG) {
        double t = 0.0;
#pragma unroll
        for (int r = 0; r < TH / 16; ++r) t += red[r * kSG + threadIdx.x];
        tot[threadIdx.x] = t;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double W = tot[0], W2 = tot[1];
        const double cf = W - W2 / W;
        double var[3];
        for (int d = 0; d < 3; ++d) {
            const double m = tot[2 + d] / W;
            var[d] = (tot[5 + d] - W * m * m) / cf;
        }
        sc_geometry_row<T>(var, ext + ((Bext == 1) ? 0 : b) * 3, energy[(Be == 1) ? 0 : b], length[(Bl == 1) ? 0 : b], mass,
                           pot_factor, gx, gy, gz, half + b * 3, cell + b * 3, gamma_out + b, dt + b, scale + b * 3,
                           extent + b * 6, pot_scale + b);
    }
}

// ---- backward of the cavity epilogue (chx_apply.hip cavity_epilogue; cavity.py:135-151,220-226) ---------------------
// delta' = a delta + b (cos(theta) - cos phi), theta = -tau kb0 + phi;  tau' = (R x)_tau + T566 d^2 + T556 tau d + T555 tau^2
// with c = [a, b, kb0, phi, cos phi, T566, T556, T555]. Given g = dL/dy: adds the epilogue's contribution to
// dL/dx (columns tau, delta) and reduces dL/dc over the particles of each batch row (8 sums, deterministic).
template <typename T>
__global__ __launch_bounds__(CHX_BLOCK) void cavity_bwd_kernel(const T* __restrict__ dY, const T* __restrict__ X,
                                                              const double* __restrict__ coeffs, T* __restrict__ dX,
                                                              int64_t Bx, int64_t N, double* __restrict__ partials) {
    __shared__ double red[4 * 8];
    const int64_t b = blockIdx.y;
    const double* c = coeffs + b * CHX_CAV_NCOEF;
    const double a = c[0], bb = c[1], kb0 = c[2], phi = c[3], cphi = c[4], T566 = c[5], T556 = c[6], T555 = c[7];
    const T* __restrict__ gy = dY + b * N * 7;
    const T* __restrict__ xb = X + ((Bx == 1) ? 0 : b) * N * 7;
    T* __restrict__ dx = dX ? dX + b * N * 7 : nullptr;
    double acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = 0.0;
    for (int64_t n = (int64_t)blockIdx.x * CHX_BLOCK + threadIdx.x; n < N; n += (int64_t)gridDim.x * CHX_BLOCK) {
        const double g4 = (double)gy[n * 7 + 4], g5 = (double)gy[n * 7 + 5];
        const double tau = (double)xb[n * 7 + 4], delta = (double)xb[n * 7 + 5];
        const double theta = -tau * kb0 + phi;
        const double s = sin(theta), co = cos(theta);
        if (dx) {
            dx[n * 7 + 4] = (T)((double)dx[n * 7 + 4] + g5 * bb * s * kb0 + g4 * (T556 * delta + 2.0 * T555 * tau));
            dx[n * 7 + 5] = (T)((double)dx[n * 7 + 5] + g5 * a + g4 * (2.0 * T566 * delta + T556 * tau));
        }
        acc[0] += g5 * delta;
        acc[1] += g5 * (co - cphi);
        acc[2] += g5 * bb * s * tau;
        acc[3] -= g5 * bb * s;
        acc[4] -= g5 * bb;
        acc[5] += g4 * delta * delta;
        acc[6] += g4 * tau * delta;
        acc[7] += g4 * tau * tau;
    }
    chx_block_sum<8>(acc, red);
    if (threadIdx.x == 0) {
        double* p = partials + ((int64_t)b * gridDim.x + blockIdx.x) * 8;
#pragma unroll
        for (int k = 0; k < 8; ++k) p[k] = acc[k];
    }
}

// The sums sc_tile_particle_kernel left behind (partials[kSG][nblk], about the origin) as a chx_moments row: [W, W2, mu(6),
// cov(21)] with the entries of x, y and tau filled (the three variances a SpaceChargeKick reads) and zeros elsewhere. One
// workgroup. This is what a rank of a particle-sharded chain puts into the 29-double all-gather between two kicks
// (chx_merge_moments merges entry by entry, so the zeros stay zeros).
__global__ __launch_bounds__(CHX_BLOCK) void sc_partials_moments_kernel(const double* __restrict__ partials, int nblk,
                                                                       double* __restrict__ out) {
    __shared__ double red[16 * kSG];
    __shared__ double tot[kSG];
    double a[kSG];
#pragma unroll
    for (int k = 0; k < kSG; ++k) a[k] = 0.0;
    for (int i0 = threadIdx.x; i0 < nblk; i0 += 4 * CHX_BLOCK) {
        double v[4][kSG];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * CHX_BLOCK;
#pragma unroll
            for (int k = 0; k < kSG; ++k) v[u][k] = i < nblk ? partials[(int64_t)k * nblk + i] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int k = 0; k < kSG; ++k) a[k] += v[u][k];
    }
#pragma unroll
    for (int k = 0; k < kSG; ++k) a[k] = chx_row16_sum(a[k]);
    if ((threadIdx.x & 15) == 0) {
#pragma unroll
        for (int k = 0; k < kSG; ++k) red[(threadIdx.x >> 4) * kSG + k] = a[k];
    }
    __syncthreads();
    if (threadIdx.x < kSG) {
        double t = 0.0;
#pragma unroll
        for (int r = 0; r < 16; ++r) t += red[r * kSG + threadIdx.x];
        tot[threadIdx.x] = t;
    }
    __syncthreads();
    if (threadIdx.x < CHX_MOM_NOUT) out[threadIdx.x] = 0.0;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double W = tot[0], W2 = tot[1];
        const double cf = W - W2 / W;
        out[0] = W;
        out[1] = W2;
        const int col[3] = {0, 2, 4};
        const int diag[3] = {0, 11, 18};     // cov_xx, cov_yy, cov_tautau in the upper triangle
        for (int d = 0; d < 3; ++d) {
            const double m = tot[2 + d] / W;
            out[2 + col[d]] = m;
            out[8 + diag[d]] = (tot[5 + d] - W * m * m) / cf;
        }
    }
}

int check_red(const void* x, int64_t B, int64_t Bx, int64_t Bw, int64_t N, int dtype) {
    if (!x || B < 1 || N < 1 || B > 65535) return CHX_ERR_INVALID_ARG;
    if (!chx_bcast_ok(Bx, B) || !chx_bcast_ok(Bw, B)) return CHX_ERR_INVALID_ARG;
    if (dtype != CHX_F32 && dtype != CHX_F64) return CHX_ERR_DTYPE;
    return CHX_OK;
}

inline int tile_rows(int dtype) { return dtype == CHX_F32 ? 512 : 256; }

}  // namespace

static size_t partials_bytes(int64_t B, int64_t N) {
    // worst case over dtypes (fp64 tiles are 256 rows -> more tiles)
    int64_t nblk = red_nblk(B, N, 256);
    const int64_t nblk1 = onepass_nblk(B, N, 256);
    if (nblk1 > nblk) nblk = nblk1;
    return (size_t)(B * nblk * 29 * sizeof(double));  // 29 = one-pass accumulators (two-pass passes need 8 / 21)
}

extern "C" size_t chx_moments_workspace_bytes(int64_t B, int64_t N) {
    if (B < 1 || N < 1) return 0;
    // per-workgroup partials + room for sums[B][8] and m2[B][21] used by chx_moments
    return partials_bytes(B, N) + (size_t)B * (CHX_MOM_NSUMS + CHX_MOM_NM2) * sizeof(double);
}

extern "C" int chx_moment_sums(const void* x, const void* w, int64_t B, int64_t Bx, int64_t Bw,
                               int64_t N, int dtype, double* sums, void* workspace,
                               size_t workspace_bytes, void* stream) {
    int st = check_red(x, B, Bx, Bw, N, dtype);
    if (st != CHX_OK) return st;
    if (!sums) return CHX_ERR_INVALID_ARG;
    const int64_t nblk = red_nblk(B, N, tile_rows(dtype));
    if (!workspace || workspace_bytes < (size_t)(B * nblk * 8 * sizeof(double))) return CHX_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    double* part = (double*)workspace;
    dim3 grid((unsigned)nblk, (unsigned)B);
    if (dtype == CHX_F32)
        hipLaunchKernelGGL(moment_sums_kernel<float>, grid, dim3(CHX_BLOCK), 0, s, (const float*)x,
                           (const float*)w, Bx, Bw, N, part);
    else
        hipLaunchKernelGGL(moment_sums_kernel<double>, grid, dim3(CHX_BLOCK), 0, s, (const double*)x,
                           (const double*)w, Bx, Bw, N, part);
    CHX_CHECK_LAUNCH();
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(8, (unsigned)B), dim3(64), 0, s, part, (int)nblk, 8, sums);
    CHX_CHECK_LAUNCH();
    return CHX_OK;
}

extern "C" int chx_moment_centred(const void* x, const void* w, const double* sums, int64_t B,
                                  int64_t Bx, int64_t Bw, int64_t N, int dtype, double* m2,
                                  void* workspace, size_t workspace_bytes, void* stream) {
    int st = check_red(x, B, Bx, Bw, N, dtype);
    if (st != CHX_OK) return st;
    if (!sums || !m2) return CHX_ERR_INVALID_ARG;
    const int64_t nblk = red_nblk(B, N, tile_rows(dtype));
    if (!workspace || workspace_bytes < (size_t)(B * nblk * 21 * sizeof(double))) return CHX_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    double* part = (double*)workspace;
    dim3 grid((unsigned)nblk, (unsigned)B);
    if (dtype == CHX_F32)
        hipLaunchKernelGGL(moment_centred_kernel<float>, grid, dim3(CHX_BLOCK), 0, s, (const float*)x,
                           (const float*)w, sums, Bx, Bw, N, part);
    else
        hipLaunchKernelGGL(moment_centred_kernel<double>, grid, dim3(CHX_BLOCK), 0, s, (const double*)x,
                           (const double*)w, sums, Bx, Bw, N, part);
    CHX_CHECK_LAUNCH();
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(21, (unsigned)B), dim3(64), 0, s, part, (int)nblk, 21, m2);
    CHX_CHECK_LAUNCH();
    return CHX_OK;
}

extern "C" int chx_moment_finalize(const double* sums, const double* m2, int64_t B, double* out,
                                   void* stream) {
    if (!sums || !m2 || !out || B < 1) return CHX_ERR_INVALID_ARG;
    hipLaunchKernelGGL(moment_finalize_kernel, dim3((unsigned)((B + 63) / 64)), dim3(64), 0,
                       (hipStream_t)stream, sums, m2, B, out);
    CHX_CHECK_LAUNCH();
    return CHX_OK;
}

template <typename T>
static void launch_reduce_finalize(int64_t B, int64_t nblk, const double* part, const double* centre, double* out, int index,
                                   int take_sqrt, void* entry_out, hipStream_t s) {
    if (nblk <= 64)
        hipLaunchKernelGGL((moments_reduce_finalize_kernel<T, 64>), dim3((unsigned)B), dim3(64), 0, s, part, (int)nblk, centre, out,
                           index, take_sqrt, (T*)entry_out);
    else if (nblk <= 256)
        hipLaunchKernelGGL((moments_reduce_finalize_kernel<T, 256>), dim3((unsigned)B), dim3(256), 0, s, part, (int)nblk, centre, out,
                           index, take_sqrt, (T*)entry_out);
    else
        hipLaunchKernelGGL((moments_reduce_finalize_kernel<T, 1024>), dim3((unsigned)B), dim3(1024), 0, s, part, (int)nblk, centre,
                           out, index, take_sqrt, (T*)entry_out);
}

// the second launch of chx_moments_entry on partial sums some other pass left (chx_lattice_screen.mom_partials): B = 1
int chx_moments_finalize_sets(const double* partials, int64_t n_sets, const double* centre, int dtype, double* out, int index, int take_sqrt,
                              void* entry_out, void* stream) {
    if (!partials || !centre || !out || n_sets < 1 || n_sets > 0x7fffffffLL) return CHX_ERR_INVALID_ARG;
    if (dtype == CHX_F32) launch_reduce_finalize<float>(1, n_sets, partials, centre, out, index, take_sqrt, entry_out, (hipStream_t)stream);
    else if (dtype == CHX_F64) launch_reduce_finalize<double>(1, n_sets, partials, centre, out, index, take_sqrt, entry_out, (hipStream_t)stream);
    else return CHX_ERR_DTYPE;
    CHX_CHECK_LAUNCH();
    return CHX_OK;
}

extern "C" int chx_moments(const void* x, const void* w, int64_t B, int64_t Bx, int64_t Bw, int64_t N,
                           int dtype, double* out, void* workspace, size_t workspace_bytes,
                           void* stream) {
    return chx_moments_entry(x, w, B, Bx, Bw, N, dtype, out, -1, 0, nullptr, workspace, workspace_bytes, stream);
}

extern "C" int chx_moments_entry(const void* x, const void* w, int64_t B, int64_t Bx, int64_t Bw, int64_t N, int dtype, double* out,
                                 int index, int take_sqrt, void* entry_out, void* workspace, size_t workspace_bytes, void* stream) {
    if (!out) return CHX_ERR_INVALID_ARG;
    if (index >= CHX_MOM_NOUT || (index >= 0 && !entry_out)) return CHX_ERR_INVALID_ARG;
    if (index < 0) index = -1;
    // out doubles as scratch for sums (first 8 of each 29-row are rewritten by finalize):
    // keep sums and m2 at the tail of the workspace instead.
    if (B < 1 || N < 1) return CHX_ERR_INVALID_ARG;
    const size_t need = partials_bytes(B, N);
    const size_t tail = (size_t)B * (CHX_MOM_NSUMS + CHX_MOM_NM2) * sizeof(double);
    if (!workspace || workspace_bytes < need + tail) return CHX_ERR_WORKSPACE;
    hipStream_t s0 = (hipStream_t)stream;
    if (B >= 64 && N <= 2048 && x && (dtype == CHX_F32 || dtype == CHX_F64) && chx_bcast_ok(Bx, B) && chx_bcast_ok(w ? Bw : 1, B) &&
        B <= 0x7fffffffLL) {
        // many short rows: one wave per row, one launch (and no 65 535-row limit: the rows are blockIdx.x)
        const unsigned grid = (unsigned)((B + CHX_BLOCK / 64 - 1) / (CHX_BLOCK / 64));
        if (dtype == CHX_F32)
            hipLaunchKernelGGL(moments_rows_wave_kernel<float>, dim3(grid), dim3(CHX_BLOCK), 0, s0, (const float*)x, (const float*)w, B, Bx,
                               w ? Bw : 1, N, out, index, take_sqrt, (float*)entry_out);
        else
            hipLaunchKernelGGL(moments_rows_wave_kernel<double>, dim3(grid), dim3(CHX_BLOCK), 0, s0, (const double*)x, (const double*)w, B,
                               Bx, w ? Bw : 1, N, out, index, take_sqrt, (double*)entry_out);
        CHX_CHECK_LAUNCH();
        return CHX_OK;
    }
    int st = check_red(x, B, Bx, w ? Bw : 1, N, dtype);
    if (st != CHX_OK) return st;
    if (!w) Bw = 1;
    // one sweep over the particles + one reduce/finalize launch (the split two-pass entry points above remain for
    // the multi-GPU path, where the means are all-reduced between the passes)
    const int64_t nblk = onepass_nblk(B, N, tile_rows(dtype));
    hipStream_t s = (hipStream_t)stream;
    double* part = (double*)workspace;
    double* centre = (double*)((char*)workspace + need);      // [B][6] in the tail behind the partial sums
    dim3 grid((unsigned)nblk, (unsigned)B);
    if (dtype == CHX_F32) {
        hipLaunchKernelGGL(moments_onepass_kernel<float>, grid, dim3(CHX_BLOCK), 0, s, (const float*)x, (const float*)w,
                           Bx, Bw, N, part, centre);
        CHX_CHECK_LAUNCH();
        launch_reduce_finalize<float>(B, nblk, part, centre, out, index, take_sqrt, entry_out, s);
    } else {
        hipLaunchKernelGGL(moments_onepass_kernel<double>, grid, dim3(CHX_BLOCK), 0, s, (const double*)x,
                           (const double*)w, Bx, Bw, N, part, centre);
        CHX_CHECK_LAUNCH();
        launch_reduce_finalize<double>(B, nblk, part, centre, out, index, take_sqrt, entry_out, s);
    }
    CHX_CHECK_LAUNCH();
    return CHX_OK;
}

extern "C" size_t chx_sc_beam_geometry_workspace_bytes(int64_t B, int64_t N) {
    if (B < 1 || N < 1) return 0;
    return (size_t)(B * onepass_nblk(B, N, 256) * kSG * sizeof(double));
}

extern "C" int chx_sc_beam_geometry(const void* x, const void* w, const void* grid_extent, const void* energy,
                                    const void* length, double mass_eV, double pot_factor, int64_t B, int64_t Bx, int64_t Bw,
                                    int64_t Bext, int64_t Be, int64_t Bl, int64_t N, const int32_t* bins, int dtype, void* half,
                                    void* cell, void* gamma, void* dt, void* scale, void* extent, double* pot_scale,
                                    void* workspace, size_t workspace_bytes, void* stream) {
    return chx_sc_beam_geometry_tiles(x, w, grid_extent, energy, length, mass_eV, pot_factor, B, Bx, Bw, Bext, Be, Bl, N, bins,
                                      dtype, half, cell, gamma, dt, scale, extent, pot_scale, workspace, workspace_bytes, nullptr,
                                      0, stream);
}

extern "C" int chx_sc_beam_geometry_tiles(const void* x, const void* w, const void* grid_extent, const void* energy,
                                          const void* length, double mass_eV, double pot_factor, int64_t B, int64_t Bx,
                                          int64_t Bw, int64_t Bext, int64_t Be, int64_t Bl, int64_t N, const int32_t* bins,
                                          int dtype, void* half, void* cell, void* gamma, void* dt, void* scale, void* extent,
                                          double* pot_scale, void* workspace, size_t workspace_bytes, void* tile_header,
                                          int tile_first, void* stream) {
    if (!grid_extent || !energy || !length || !half || !cell || !gamma || !dt || !scale || !extent || !pot_scale || !bins)
        return CHX_ERR_INVALID_ARG;
    int st = check_red(x, B, Bx, w ? Bw : 1, N, dtype);
    if (st != CHX_OK) return st;
    if (!chx_bcast_ok(Bext, B) || !chx_bcast_ok(Be, B) || !chx_bcast_ok(Bl, B)) return CHX_ERR_INVALID_ARG;
    if (!workspace || workspace_bytes < chx_sc_beam_geometry_workspace_bytes(B, N)) return CHX_ERR_WORKSPACE;
    if (!w) Bw = 1;
    const int64_t nblk = onepass_nblk(B, N, tile_rows(dtype));
    hipStream_t s = (hipStream_t)stream;
    double* part = (double*)workspace;
    const dim3 grid((unsigned)nblk, (unsigned)B);
    if (dtype == CHX_F32) {
        hipLaunchKernelGGL(sc_sigma_kernel<float>, grid, dim3(CHX_BLOCK), 0, s, (const float*)x, (const float*)w, Bx, Bw, N, part,
                           tile_first ? (const int*)nullptr : (const int*)tile_header);
        CHX_CHECK_LAUNCH();
        hipLaunchKernelGGL(sc_geometry_partials_kernel<float>, dim3((unsigned)B), dim3(CHX_BLOCK), 0, s, part, (int)nblk,
                           (const float*)grid_extent, (const float*)energy, (const float*)length, mass_eV, pot_factor, Bext, Be,
                           Bl, bins[0], bins[1], bins[2], (float*)half, (float*)cell, (float*)gamma, (float*)dt, (float*)scale,
                           (float*)extent, pot_scale, (int*)tile_header, tile_first);
    } else {
        hipLaunchKernelGGL(sc_sigma_kernel<double>, grid, dim3(CHX_BLOCK), 0, s, (const double*)x, (const double*)w, Bx, Bw, N,
                           part, tile_first ? (const int*)nullptr : (const int*)tile_header);
        CHX_CHECK_LAUNCH();
        hipLaunchKernelGGL(sc_geometry_partials_kernel<double>, dim3((unsigned)B), dim3(CHX_BLOCK), 0, s, part, (int)nblk,
                           (const double*)grid_extent, (const double*)energy, (const double*)length, mass_eV, pot_factor, Bext,
                           Be, Bl, bins[0], bins[1], bins[2], (double*)half, (double*)cell, (double*)gamma, (double*)dt,
                           (double*)scale, (double*)extent, pot_scale, (int*)tile_header, tile_first);
    }
    CHX_CHECK_LAUNCH();
    return CHX_OK;
}

// the geometry kernel alone, from partial sums some other kernel left behind (the gather pass of the previous kick of a chain
// accumulates them over the rows it writes: chx_sc_tile_gather_kick) — layout partials[kSG][nblk]
extern "C" int chx_sc_geometry_from_partials(const double* partials, int64_t nblk, const void* grid_extent, const void* energy,
                                             const void* length, double mass_eV, double pot_factor, const int32_t* bins, int dtype,
                                             void* half, void* cell, void* gamma, void* dt, void* scale, void* extent,
                                             double* pot_scale, void* tile_header, void* stream) {
    if (!partials || nblk < 1 || nblk > 0x7fffffff || !grid_extent || !energy || !length || !half || !cell || !gamma || !dt || !scale ||
        !extent || !pot_scale || !bins)
        return CHX_ERR_INVALID_ARG;
    hipStream_t s = (hipStream_t)stream;
    const bool wide = nblk > 4 * CHX_BLOCK;
    if (dtype == CHX_F32) {
        auto kern = wide ? sc_geometry_partials_kernel<float, 1024> : sc_geometry_partials_kernel<float, CHX_BLOCK>;
        hipLaunchKernelGGL(kern, dim3(1), dim3(wide ? 1024 : CHX_BLOCK), 0, s, partials, (int)nblk,
                           (const float*)grid_extent, (const float*)energy, (const float*)length, mass_eV, pot_factor, 1, 1, 1, bins[0],
                           bins[1], bins[2], (float*)half, (float*)cell, (float*)gamma, (float*)dt, (float*)scale, (float*)extent,
                           pot_scale, (int*)tile_header, 0);
    } else if (dtype == CHX_F64) {
        auto kern = wide ? sc_geometry_partials_kernel<double, 1024> : sc_geometry_partials_kernel<double, CHX_BLOCK>;
        hipLaunchKernelGGL(kern, dim3(1), dim3(wide ? 1024 : CHX_BLOCK), 0, s, partials, (int)nblk,
                           (const double*)grid_extent, (const double*)energy, (const double*)length, mass_eV, pot_factor, 1, 1, 1,
                           bins[0], bins[1], bins[2], (double*)half, (double*)cell, (double*)gamma, (double*)dt, (double*)scale,
                           (double*)extent, pot_scale, (int*)tile_header, 0);
    } else
        return CHX_ERR_DTYPE;
    CHX_CHECK_LAUNCH();
    return CHX_OK;
}

extern "C" int chx_sc_partials_moments(const double* partials, int64_t nblk, double* moments_out, void* stream) {
    if (!partials || !moments_out || nblk < 1 || nblk > 0x7fffffff) return CHX_ERR_INVALID_ARG;
    hipLaunchKernelGGL(sc_partials_moments_kernel, dim3(1), dim3(CHX_BLOCK), 0, (hipStream_t)stream, partials, (int)nblk, moments_out);
    CHX_CHECK_LAUNCH();
    return CHX_OK;
}

extern "C" int chx_cavity_track_bwd(const void* dY, const void* X, const double* coeffs, void* dX, double* dcoeffs,
                                    int64_t B, int64_t Bx, int64_t N, int dtype, void* workspace,
                                    size_t workspace_bytes, void* stream) {
    int st = check_red(X, B, Bx, 1, N, dtype);
    if (st != CHX_OK) return st;
    if (!dY || !coeffs || !dcoeffs) return CHX_ERR_INVALID_ARG;
    const int64_t nblk = red_nblk(B, N, tile_rows(dtype));
    if (!workspace || workspace_bytes < (size_t)(B * nblk * 8 * sizeof(double))) return CHX_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    double* part = (double*)workspace;
    dim3 grid((unsigned)nblk, (unsigned)B);
    if (dtype == CHX_F32)
        hipLaunchKernelGGL(cavity_bwd_kernel<float>, grid, dim3(CHX_BLOCK), 0, s, (const float*)dY, (const float*)X,
                           coeffs, (float*)dX, Bx, N, part);
    else
        hipLaunchKernelGGL(cavity_bwd_kernel<double>, grid, dim3(CHX_BLOCK), 0, s, (const double*)dY, (const double*)X,
                           coeffs, (double*)dX, Bx, N, part);
    CHX_CHECK_LAUNCH();
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(8, (unsigned)B), dim3(64), 0, s, part, (int)nblk, 8, dcoeffs);
    CHX_CHECK_LAUNCH();
    return CHX_OK;
}

// ---- fused track + moments ------------------------------------------------------------------------------------
static int64_t tm_chunks(int64_t B, int64_t N, bool rows_path, int dtype) {
    if (rows_path) {
        // ~1024 workgroups in total; each chunk at least one LDS fill
        const int64_t row_blocks = (B + CHX_BLOCK - 1) / CHX_BLOCK;
        int64_t chunks = (1024 + row_blocks - 1) / row_blocks;
        const int64_t fill = dtype == CHX_F32 ? 1024 : 512;
        const int64_t max_chunks = (N + fill - 1) / fill;
        if (chunks > max_chunks) chunks = max_chunks;
        return chunks < 1 ? 1 : chunks;
    }
    return red_nblk(B, N, tile_rows(dtype));
}
static bool tm_rows_path(int64_t B, int64_t Bx, int64_t BR, int64_t Bw) { return Bx == 1 && Bw == 1 && BR == B && B >= 64; }

extern "C" size_t chx_track_moments_workspace_bytes(int64_t B, int64_t N) {
    if (B < 1 || N < 1) return 0;
    int64_t c = tm_chunks(B, N, true, CHX_F64);
    const int64_t c2 = tm_chunks(B, N, false, CHX_F64);
    if (c2 > c) c = c2;
    return (size_t)(c * B * kTM * sizeof(double));
}

extern "C" int chx_track_moments(const void* x_in, const void* w, const void* R, const double* centre, int64_t B,
                                 int64_t Bx, int64_t BR, int64_t Bw, int64_t N, int dtype, double* out,
                                 void* workspace, size_t workspace_bytes, void* stream) {
    int st = check_red(x_in, B, Bx, w ? Bw : 1, N, dtype);
    if (st != CHX_OK) return st;
    if (!R || !out || !chx_bcast_ok(BR, B)) return CHX_ERR_INVALID_ARG;
    if (!w) Bw = 1;
    const bool rows = tm_rows_path(B, Bx, BR, Bw);
    const int64_t chunks = tm_chunks(B, N, rows, dtype);
    if (!workspace || workspace_bytes < (size_t)(chunks * B * kTM * sizeof(double))) return CHX_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    double* part = (double*)workspace;
    if (rows) {
        const int64_t per_chunk = (N + chunks - 1) / chunks;
        dim3 grid((unsigned)((B + CHX_BLOCK - 1) / CHX_BLOCK), (unsigned)chunks);
        if (dtype == CHX_F32)
            hipLaunchKernelGGL(track_moments_rows_f32_kernel, grid, dim3(CHX_BLOCK), 0, s, (const float*)x_in,
                               (const float*)w, (const float*)R, centre, B, N, per_chunk, part);
        else
            hipLaunchKernelGGL(track_moments_rows_kernel<double>, grid, dim3(CHX_BLOCK), 0, s, (const double*)x_in,
                               (const double*)w, (const double*)R, centre, B, N, per_chunk, part);
    } else {
        dim3 grid((unsigned)chunks, (unsigned)B);
        if (dtype == CHX_F32)
            hipLaunchKernelGGL(track_moments_particles_kernel<float>, grid, dim3(CHX_BLOCK), 0, s, (const float*)x_in,
                               (const float*)w, (const float*)R, centre, Bx, BR, Bw, N, part);
        else
            hipLaunchKernelGGL(track_moments_particles_kernel<double>, grid, dim3(CHX_BLOCK), 0, s, (const double*)x_in,
                               (const double*)w, (const double*)R, centre, Bx, BR, Bw, N, part);
    }
    CHX_CHECK_LAUNCH();
    const unsigned fb = (unsigned)((B + 63) / 64);
    if (dtype == CHX_F32)
        hipLaunchKernelGGL(track_moments_finalize_kernel<float>, dim3(fb), dim3(64), 0, s, part, (int)chunks,
                           (const float*)R, centre, B, Bx, BR, out);
    else
        hipLaunchKernelGGL(track_moments_finalize_kernel<double>, dim3(fb), dim3(64), 0, s, part, (int)chunks,
                           (const double*)R, centre, B, Bx, BR, out);
    CHX_CHECK_LAUNCH();
    return CHX_OK;
}

extern "C" int chx_moments_bwd(const void* x, const void* w, const double* out, const double* d_out,
                               int64_t B, int64_t Bx, int64_t Bw, int64_t N, int dtype, void* dX,
                               void* stream) {
    if (!dX) return CHX_ERR_INVALID_ARG;
    return chx_moments_bwd_w(x, w, out, d_out, B, Bx, Bw, N, dtype, dX, nullptr, stream);
}

extern "C" int chx_moments_bwd_w(const void* x, const void* w, const double* out, const double* d_out, int64_t B, int64_t Bx,
                                 int64_t Bw, int64_t N, int dtype, void* dX, void* dW, void* stream) {
    int st = check_red(x, B, Bx, Bw, N, dtype);
    if (st != CHX_OK) return st;
    if (!out || !d_out || (!dX && !dW)) return CHX_ERR_INVALID_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int tr = tile_rows(dtype);
    dim3 grid((unsigned)((N + tr - 1) / tr), (unsigned)B);
    if (dtype == CHX_F32)
        hipLaunchKernelGGL(moments_bwd_kernel<float>, grid, dim3(CHX_BLOCK), 0, s, (const float*)x,
                           (const float*)w, out, d_out, Bx, Bw, N, (float*)dX, (float*)dW);
    else
        hipLaunchKernelGGL(moments_bwd_kernel<double>, grid, dim3(CHX_BLOCK), 0, s, (const double*)x,
                           (const double*)w, out, d_out, Bx, Bw, N, (double*)dX, (double*)dW);
    CHX_CHECK_LAUNCH();
    return CHX_OK;
}

// workgroups of apply_bwd_dR_kernel per batch row: sized for latency hiding like the one-pass moments kernel
static int64_t dR_nblk(int64_t B, int64_t N) { return onepass_nblk(B, N, CHX_BLOCK); }

extern "C" size_t chx_apply_bwd_workspace_bytes(int64_t B, int64_t N) {
    if (B < 1 || N < 1) return 0;
    return (size_t)(B * dR_nblk(B, N) * 49 * sizeof(double)) + (size_t)B * 49 * sizeof(double);
}

extern "C" int chx_apply_affine7_bwd(const void* dY, const void* R, const void* X, void* dX,
                                     double* dR, int64_t B, int64_t Bx, int64_t BR, int64_t N,
                                     int dtype, void* workspace, size_t workspace_bytes,
                                     void* stream) {
    int st = check_red(dY, B, Bx, BR, N, dtype);
    if (st != CHX_OK) return st;
    if (!workspace || workspace_bytes < chx_apply_bwd_workspace_bytes(B, N)) return CHX_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const int64_t nblk = dR_nblk(B, N);
    double* part = (double*)workspace;
    void* Rt = (char*)workspace + (size_t)(B * nblk * 49 * sizeof(double));
    if (dX) {
        if (!R) return CHX_ERR_INVALID_ARG;
        // dX = dY . R  == apply with R^T
        const unsigned g = (unsigned)((BR * 49 + 255) / 256);
        if (dtype == CHX_F32)
            hipLaunchKernelGGL(transpose_maps_kernel<float>, dim3(g), dim3(256), 0, s, (const float*)R, BR, (float*)Rt);
        else
            hipLaunchKernelGGL(transpose_maps_kernel<double>, dim3(g), dim3(256), 0, s, (const double*)R, BR, (double*)Rt);
        CHX_CHECK_LAUNCH();
        st = chx_apply_affine7(dY, Rt, dX, B, B, BR, N, dtype, stream);
        if (st != CHX_OK) return st;
    }
    if (dR) {
        if (!X) return CHX_ERR_INVALID_ARG;
        dim3 grid((unsigned)nblk, (unsigned)B);
        if (dtype == CHX_F32)
            hipLaunchKernelGGL(apply_bwd_dR_kernel<float>, grid, dim3(CHX_BLOCK), 0, s, (const float*)dY,
                               (const float*)X, Bx, N, part);
        else
            hipLaunchKernelGGL(apply_bwd_dR_kernel<double>, grid, dim3(CHX_BLOCK), 0, s, (const double*)dY,
                               (const double*)X, Bx, N, part);
        CHX_CHECK_LAUNCH();
        hipLaunchKernelGGL(reduce_partials_t_kernel, dim3(49, (unsigned)B), dim3(64), 0, s, part, (int)nblk, 49, dR);
        CHX_CHECK_LAUNCH();
    }
    return CHX_OK;
}

// ---- exact merge of per-rank moments (multi-GPU; cheetah_amd/sharding.py merge_moments) ---------------------------------
// per_rank[R][B][29] = [W, W2, mu(6), unbiased cov upper triangle(21)] of every shard -> out[B][29] of their union
// (Chan et al.: M = sum_r [M_r + W_r (mu_r - mu)(mu_r - mu)^T]); shards without weight contribute nothing. One thread per
// batch row: R is the number of GPUs, this replaces ~30 tiny tensor kernels per merge.
namespace {
__global__ void merge_moments_kernel(const double* __restrict__ per_rank, int R, int64_t B, double* __restrict__ out) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double W = 0.0, W2 = 0.0, mu[6] = {0, 0, 0, 0, 0, 0};
    for (int r = 0; r < R; ++r) {
        const double* p = per_rank + ((int64_t)r * B + b) * CHX_MOM_NOUT;
        if (!(p[0] > 0.0)) continue;
        W += p[0];
        W2 += p[1];
        for (int j = 0; j < 6; ++j) mu[j] += p[0] * p[2 + j];
    }
    for (int j = 0; j < 6; ++j) mu[j] /= W;
    double M[21];
    for (int k = 0; k < 21; ++k) M[k] = 0.0;
    for (int r = 0; r < R; ++r) {
        const double* p = per_rank + ((int64_t)r * B + b) * CHX_MOM_NOUT;
        if (!(p[0] > 0.0)) continue;
        const double cf = p[0] - p[1] / p[0];
        double d[6];
        for (int j = 0; j < 6; ++j) d[j] = p[2 + j] - mu[j];
        int k = 0;
        for (int i = 0; i < 6; ++i)
            for (int j = i; j < 6; ++j, ++k) M[k] += p[8 + k] * cf + p[0] * d[i] * d[j];
    }
    double* o = out + b * CHX_MOM_NOUT;
    o[0] = W;
    o[1] = W2;
    for (int j = 0; j < 6; ++j) o[2 + j] = mu[j];
    const double cf = W - W2 / W;
    for (int k = 0; k < 21; ++k) o[8 + k] = M[k] / cf;
}
}  // namespace

extern "C" int chx_merge_moments(const double* per_rank, int32_t R, int64_t B, double* out, void* stream) {
    if (!per_rank || !out || R < 1 || B < 1) return CHX_ERR_INVALID_ARG;
    hipLaunchKernelGGL(merge_moments_kernel, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, (hipStream_t)stream, per_rank,
                       (int)R, B, out);
    CHX_CHECK_LAUNCH();
    return CHX_OK;
}

// ---- moments of a linearly tracked beam: backward with respect to the MAP, without touching a particle -------------------
// y_n = R x_n (x_6 = 1) with weights w_n: mu' = A mu + b, cov' = A C A^T with A = R[:6,:6], b = R[:6,6], (mu, C) the moments of
// the INCOMING beam (element.py:180-191 followed by statistics.py:4-62). Given d_out[B][29] of chx_moments(y) this is
//   dA = 2 G A C + g_mu mu^T,  db = g_mu,   G = the symmetric matrix whose upper triangle carries d_out[8:29]
//   (off-diagonal entries halved: cov'_ij and cov'_ji are one output).
// Exactly what autograd gives through chx_moments_bwd + chx_apply_affine7_bwd's dR reduction (sum_n dY_n x_n^T) when the
// particles carry no gradient of their own — 32 B/particle (one pass for mu, C, cacheable across steps) instead of 232.
namespace {
// one wave per batch row
template <typename T>
__global__ __launch_bounds__(64) void moments_mapped_bwd_kernel(const double* __restrict__ d_out, const T* __restrict__ R, int64_t BR,
                                                                const double* __restrict__ mom_x, int64_t Bm, int64_t B,
                                                                double* __restrict__ dR) {
    __shared__ double lds[3 * 36];
    const int64_t b = blockIdx.x;
    mapped_bwd_row_wave<T, double>(d_out + b * CHX_MOM_NOUT, R + (BR == 1 ? 0 : b) * 49, mom_x + (Bm == 1 ? 0 : b) * CHX_MOM_NOUT, lds,
                                   dR + b * 49);
}

// one ENTRY of the moment vector (optionally its square root) in the beam dtype, and its backward straight to dR: the node a
// scalar loss like sigma_x(screen) hangs on (particle_beam.py:1672-1943 properties)
template <typename T>
__global__ void moment_entry_kernel(const double* __restrict__ mom, int64_t B, int index, int take_sqrt, T* __restrict__ out) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const double v = mom[b * CHX_MOM_NOUT + index];
    out[b] = (T)(take_sqrt ? sqrt(v) : v);
}

template <typename T, typename TO>
__global__ __launch_bounds__(64) void moment_entry_mapped_bwd_kernel(const T* __restrict__ grad, const double* __restrict__ mom_y,
                                                                     int index, int take_sqrt, const T* __restrict__ R, int64_t BR,
                                                                     const double* __restrict__ mom_x, int64_t Bm, int64_t B,
                                                                     TO* __restrict__ dR) {
    __shared__ double lds[3 * 36];
    __shared__ double g[CHX_MOM_NOUT];
    const int64_t b = blockIdx.x;
    moment_entry_gradient((double)grad[b], mom_y + b * CHX_MOM_NOUT, index, take_sqrt, g);
    mapped_bwd_row_wave<T, TO>(g, R + (BR == 1 ? 0 : b) * 49, mom_x + (Bm == 1 ? 0 : b) * CHX_MOM_NOUT, lds, dR + b * 49);
}
}  // namespace

extern "C" int chx_moments_mapped_bwd(const double* d_out, const void* R, const double* mom_x, int64_t B, int64_t BR,
                                      int64_t Bm, int dtype, double* dR, void* stream) {
    if (!d_out || !R || !mom_x || !dR || B < 1 || B > 0x7fffffffLL || (BR != 1 && BR != B) || (Bm != 1 && Bm != B))
        return CHX_ERR_INVALID_ARG;
    const dim3 grid((unsigned)B), block(64);
    if (dtype == CHX_F32)
        hipLaunchKernelGGL(moments_mapped_bwd_kernel<float>, grid, block, 0, (hipStream_t)stream, d_out, (const float*)R, BR,
                           mom_x, Bm, B, dR);
    else if (dtype == CHX_F64)
        hipLaunchKernelGGL(moments_mapped_bwd_kernel<double>, grid, block, 0, (hipStream_t)stream, d_out, (const double*)R, BR,
                           mom_x, Bm, B, dR);
    else
        return CHX_ERR_DTYPE;
    CHX_CHECK_LAUNCH();
    return CHX_OK;
}

extern "C" int chx_moment_entry(const double* mom, int64_t B, int index, int take_sqrt, int dtype, void* out, void* stream) {
    if (!mom || !out || B < 1 || index < 0 || index >= CHX_MOM_NOUT) return CHX_ERR_INVALID_ARG;
    const dim3 grid((unsigned)((B + 63) / 64)), block(64);
    if (dtype == CHX_F32)
        hipLaunchKernelGGL(moment_entry_kernel<float>, grid, block, 0, (hipStream_t)stream, mom, B, index, take_sqrt, (float*)out);
    else if (dtype == CHX_F64)
        hipLaunchKernelGGL(moment_entry_kernel<double>, grid, block, 0, (hipStream_t)stream, mom, B, index, take_sqrt, (double*)out);
    else
        return CHX_ERR_DTYPE;
    CHX_CHECK_LAUNCH();
    return CHX_OK;
}

extern "C" int chx_moment_entry_mapped_bwd(const void* grad, const double* mom_y, int index, int take_sqrt, const void* R,
                                           const double* mom_x, int64_t B, int64_t BR, int64_t Bm, int dtype, void* dR,
                                           int dR_is_double, void* stream) {
    if (!grad || !mom_y || !R || !mom_x || !dR || B < 1 || B > 0x7fffffffLL || (BR != 1 && BR != B) || (Bm != 1 && Bm != B) ||
        index < 2 || index >= CHX_MOM_NOUT)
        return CHX_ERR_INVALID_ARG;
    const dim3 grid((unsigned)B), block(64);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == CHX_F32) {
        if (dR_is_double)
            hipLaunchKernelGGL((moment_entry_mapped_bwd_kernel<float, double>), grid, block, 0, s, (const float*)grad, mom_y, index,
                               take_sqrt, (const float*)R, BR, mom_x, Bm, B, (double*)dR);
        else
            hipLaunchKernelGGL((moment_entry_mapped_bwd_kernel<float, float>), grid, block, 0, s, (const float*)grad, mom_y, index,
                               take_sqrt, (const float*)R, BR, mom_x, Bm, B, (float*)dR);
    } else if (dtype == CHX_F64) {
        hipLaunchKernelGGL((moment_entry_mapped_bwd_kernel<double, double>), grid, block, 0, s, (const double*)grad, mom_y, index,
                           take_sqrt, (const double*)R, BR, mom_x, Bm, B, (double*)dR);
    } else {
        return CHX_ERR_DTYPE;
    }
    CHX_CHECK_LAUNCH();
    return CHX_OK;
}

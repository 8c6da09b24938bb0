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
#include "chx_common.h"

namespace {

template <typename T> struct red_cfg;
template <> struct red_cfg<float> { static constexpr int PPT = 2; };
template <> struct red_cfg<double> { static constexpr int PPT = 1; };

__host__ __device__ inline int64_t red_nblk(int64_t B, int64_t N, int tile_rows) {
    int64_t tiles = (N + tile_rows - 1) / tile_rows;
    int64_t cap = 1024 / B;  // ~4 workgroups per CU in total; each loops over many rows
    if (cap < 1) cap = 1;
    return tiles < cap ? tiles : cap;
}

// Generic reduction over the particles of batch row b = blockIdx.y. Each lane streams its own 28-/56-byte
// rows straight from global memory (measured on MI355X: dword-strided row reads reach the same bandwidth as
// LDS-staged float4 tiles, benchmarks/apply_variants.hip "direct dword"), 2 rows in flight per lane and
// iteration, no barrier inside the loop; one block reduction at the end.
// F::accumulate(x[7], w, n, acc[K]) is called once per particle.
template <typename T, int K, typename F>
__device__ __forceinline__ void tiled_reduce(const T* __restrict__ x, const T* __restrict__ w,
                                             int64_t Bx, int64_t Bw, int64_t N, F& f,
                                             double* __restrict__ partial_out /*[K]*/) {
    __shared__ double red[4 * K];
    const int64_t b = blockIdx.y;
    const int64_t xrow = (Bx == 1) ? 0 : b, wrow = (Bw == 1) ? 0 : b;
    const T* __restrict__ xb = x + xrow * N * 7;
    const T* __restrict__ wb = w ? w + wrow * N : nullptr;
    double acc[K];
#pragma unroll
    for (int k = 0; k < K; ++k) acc[k] = 0.0;
    const int64_t stride = (int64_t)gridDim.x * CHX_BLOCK;
    int64_t n = (int64_t)blockIdx.x * CHX_BLOCK + threadIdx.x;
    for (; n + stride < N; n += 2 * stride) {
        T r0[7], r1[7];
#pragma unroll
        for (int j = 0; j < 7; ++j) { r0[j] = xb[n * 7 + j]; r1[j] = xb[(n + stride) * 7 + j]; }
        const double w0 = wb ? (double)wb[n] : 1.0, w1 = wb ? (double)wb[n + stride] : 1.0;
        double xv[7];
#pragma unroll
        for (int j = 0; j < 7; ++j) xv[j] = (double)r0[j];
        f.accumulate(xv, w0, n, acc);
#pragma unroll
        for (int j = 0; j < 7; ++j) xv[j] = (double)r1[j];
        f.accumulate(xv, w1, n + stride, acc);
    }
    if (n < N) {
        double xv[7];
#pragma unroll
        for (int j = 0; j < 7; ++j) xv[j] = (double)xb[n * 7 + j];
        f.accumulate(xv, wb ? (double)wb[n] : 1.0, n, acc);
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
                                                               T* __restrict__ dX) {
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
    const int64_t n0 = (int64_t)blockIdx.x * TP;
    const int np = (int)((N - n0 < TP) ? (N - n0) : TP);
    const bool vin = chx_aligned16(x) && (((xrow * N * 7 * (int64_t)sizeof(T)) & 15) == 0);
    const bool vout = chx_aligned16(dX) && (((b * N * 7 * (int64_t)sizeof(T)) & 15) == 0);
    tile_load<T>(x + (xrow * N + n0) * 7, lds, np * 7, vin);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const int p = threadIdx.x + k * CHX_BLOCK;
        if (p < np) {
            double d[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) d[j] = (double)lds[p * 7 + j] - o[2 + j];
            const double wv = w ? (double)w[wrow * N + n0 + p] : 1.0;
#pragma unroll
            for (int a = 0; a < 6; ++a) {
                double s = 0.0;
#pragma unroll
                for (int c = 0; c < 6; ++c) s += G[a][c] * d[c];
                lds[p * 7 + a] = (T)(wv * (g[2 + a] / W + icf * s));
            }
            lds[p * 7 + 6] = (T)0;
        }
    }
    __syncthreads();
    tile_store<T>(dX + (b * N + n0) * 7, lds, np * 7, vout);
}

// dR[i][j] = sum_n dY[n][i] X[n][j]   (49 fp64 accumulators per lane)
template <typename T>
__global__ __launch_bounds__(CHX_BLOCK) void apply_bwd_dR_kernel(const T* __restrict__ dY,
                                                                const T* __restrict__ X, int64_t Bx,
                                                                int64_t N,
                                                                double* __restrict__ partials) {
    constexpr int PPT = red_cfg<T>::PPT;
    constexpr int TP = PPT * CHX_BLOCK;
    __shared__ __attribute__((aligned(16))) T lx[TP * 7];
    __shared__ __attribute__((aligned(16))) T ly[TP * 7];
    __shared__ double red[4 * 49];
    const int64_t b = blockIdx.y;
    const int64_t xrow = (Bx == 1) ? 0 : b;
    const bool vx = chx_aligned16(X) && (((xrow * N * 7 * (int64_t)sizeof(T)) & 15) == 0);
    const bool vy = chx_aligned16(dY) && (((b * N * 7 * (int64_t)sizeof(T)) & 15) == 0);
    const int64_t tiles = (N + TP - 1) / TP;
    double acc[49];
#pragma unroll
    for (int k = 0; k < 49; ++k) acc[k] = 0.0;
    for (int64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
        const int64_t n0 = t * TP;
        const int np = (int)((N - n0 < TP) ? (N - n0) : TP);
        __syncthreads();
        tile_load<T>(X + (xrow * N + n0) * 7, lx, np * 7, vx);
        tile_load<T>(dY + (b * N + n0) * 7, ly, np * 7, vy);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int p = threadIdx.x + k * CHX_BLOCK;
            if (p < np) {
                double xv[7], yv[7];
#pragma unroll
                for (int j = 0; j < 7; ++j) { xv[j] = (double)lx[p * 7 + j]; yv[j] = (double)ly[p * 7 + j]; }
#pragma unroll
                for (int i = 0; i < 7; ++i)
#pragma unroll
                    for (int j = 0; j < 7; ++j) acc[i * 7 + j] += yv[i] * xv[j];
            }
        }
    }
    chx_block_sum<49>(acc, red);
    if (threadIdx.x == 0) {
        double* o = partials + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 49;
#pragma unroll
        for (int k = 0; k < 49; ++k) o[k] = acc[k];
    }
}

template <typename T>
__global__ void transpose_maps_kernel(const T* __restrict__ R, int64_t B, T* __restrict__ Rt) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * 49) return;
    const int64_t b = idx / 49;
    const int k = (int)(idx - b * 49), i = k / 7, j = k - 7 * i;
    Rt[b * 49 + j * 7 + i] = R[idx];
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
    const int64_t nblk = red_nblk(B, N, 256);
    return (size_t)(B * nblk * 21 * sizeof(double));
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

extern "C" int chx_moments(const void* x, const void* w, int64_t B, int64_t Bx, int64_t Bw, int64_t N,
                           int dtype, double* out, void* workspace, size_t workspace_bytes,
                           void* stream) {
    if (!out) return CHX_ERR_INVALID_ARG;
    // out doubles as scratch for sums (first 8 of each 29-row are rewritten by finalize):
    // keep sums and m2 at the tail of the workspace instead.
    if (B < 1 || N < 1) return CHX_ERR_INVALID_ARG;
    const size_t need = partials_bytes(B, N);
    const size_t tail = (size_t)B * (CHX_MOM_NSUMS + CHX_MOM_NM2) * sizeof(double);
    if (!workspace || workspace_bytes < need + tail) return CHX_ERR_WORKSPACE;
    double* sums = (double*)((char*)workspace + need);
    double* m2 = sums + B * CHX_MOM_NSUMS;
    int st = chx_moment_sums(x, w, B, Bx, Bw, N, dtype, sums, workspace, need, stream);
    if (st != CHX_OK) return st;
    st = chx_moment_centred(x, w, sums, B, Bx, Bw, N, dtype, m2, workspace, need, stream);
    if (st != CHX_OK) return st;
    return chx_moment_finalize(sums, m2, B, out, stream);
}

extern "C" int chx_moments_bwd(const void* x, const void* w, const double* out, const double* d_out,
                               int64_t B, int64_t Bx, int64_t Bw, int64_t N, int dtype, void* dX,
                               void* stream) {
    int st = check_red(x, B, Bx, Bw, N, dtype);
    if (st != CHX_OK) return st;
    if (!out || !d_out || !dX) return CHX_ERR_INVALID_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int tr = tile_rows(dtype);
    dim3 grid((unsigned)((N + tr - 1) / tr), (unsigned)B);
    if (dtype == CHX_F32)
        hipLaunchKernelGGL(moments_bwd_kernel<float>, grid, dim3(CHX_BLOCK), 0, s, (const float*)x,
                           (const float*)w, out, d_out, Bx, Bw, N, (float*)dX);
    else
        hipLaunchKernelGGL(moments_bwd_kernel<double>, grid, dim3(CHX_BLOCK), 0, s, (const double*)x,
                           (const double*)w, out, d_out, Bx, Bw, N, (double*)dX);
    CHX_CHECK_LAUNCH();
    return CHX_OK;
}

extern "C" size_t chx_apply_bwd_workspace_bytes(int64_t B, int64_t N) {
    if (B < 1 || N < 1) return 0;
    const int64_t nblk = red_nblk(B, N, 256);
    return (size_t)(B * nblk * 49 * sizeof(double)) + (size_t)B * 49 * sizeof(double);
}

extern "C" int chx_apply_affine7_bwd(const void* dY, const void* R, const void* X, void* dX,
                                     double* dR, int64_t B, int64_t Bx, int64_t BR, int64_t N,
                                     int dtype, void* workspace, size_t workspace_bytes,
                                     void* stream) {
    int st = check_red(dY, B, Bx, BR, N, dtype);
    if (st != CHX_OK) return st;
    if (!workspace || workspace_bytes < chx_apply_bwd_workspace_bytes(B, N)) return CHX_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const int64_t nblk = red_nblk(B, N, tile_rows(dtype));
    double* part = (double*)workspace;
    void* Rt = (char*)workspace + (size_t)(B * red_nblk(B, N, 256) * 49 * sizeof(double));
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
        hipLaunchKernelGGL(reduce_partials_kernel, dim3(49, (unsigned)B), dim3(64), 0, s, part, (int)nblk, 49, dR);
        CHX_CHECK_LAUNCH();
    }
    return CHX_OK;
}

// chx_kde.hip — Gaussian kernel values of Screen(method="kde") (cheetah/utils/kde.py:4-77): for every particle n and bin
// centre c_i,  K[b][n][i] = max(w_bn exp(-((v_bn - c_i) / sigma)^2 / 2) / sqrt(2 pi sigma^2), tiny(dtype)),
// v = column `col` of the 7-vector minus an optional shift, w = |charge| * survival (or 1). The joint density is the
// plain GEMM K1^T K2 over the particle axis, which the host layer hands to rocBLAS through torch.matmul.
#include <cfloat>

#include "chx_common.h"

namespace {

template <typename T> __device__ __forceinline__ T tiny_of();
template <> __device__ __forceinline__ float tiny_of<float>() { return FLT_MIN; }
template <> __device__ __forceinline__ double tiny_of<double>() { return DBL_MIN; }

template <typename T>
__global__ __launch_bounds__(CHX_BLOCK) void kde_values_kernel(const T* __restrict__ x, const T* __restrict__ q,
                                                              const T* __restrict__ s, const T* __restrict__ shift,
                                                              const T* __restrict__ centres, const T* __restrict__ sigma,
                                                              int col, int64_t Bx, int64_t Bq, int64_t Bs, int64_t Bsh,
                                                              int64_t N, int64_t n0, int64_t nchunk, int nbins,
                                                              T* __restrict__ out) {
    const int64_t b = blockIdx.y;
    const T sg = sigma[0];
    // the reference evaluates  w * exp(-0.5 (r / sigma)^2) / sqrt(2 pi sigma^2)  in this order, in the working dtype
    const T norm = sqrt((T)(2.0 * 3.14159265358979323846) * (sg * sg));
    const int64_t total = nchunk * nbins;
    for (int64_t idx = (int64_t)blockIdx.x * CHX_BLOCK + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * CHX_BLOCK) {
        const int64_t n = idx / nbins;
        const int i = (int)(idx - n * nbins);
        const int64_t gn = n0 + n;
        T v = x[(((Bx == 1) ? 0 : b) * N + gn) * 7 + col];
        if (shift) v = v - shift[((Bsh == 1) ? 0 : b) * 2 + (col == 0 ? 0 : 1)];
        T w = (T)1;
        if (q) w = fabs(q[((Bq == 1) ? 0 : b) * N + gn]);
        if (s) w = w * s[((Bs == 1) ? 0 : b) * N + gn];
        const T r = (v - centres[i]) / sg;
        T k = w * exp((T)-0.5 * (r * r)) / norm;
        const T tiny = tiny_of<T>();
        out[(b * nchunk + n) * nbins + i] = k < tiny ? tiny : k;
    }
}

}  // namespace

extern "C" int chx_kde_values(const void* x, const void* charge, const void* survival, const void* shift,
                              const void* centres, const void* sigma, int col, int64_t B, int64_t Bx, int64_t Bq,
                              int64_t Bs, int64_t Bsh, int64_t N, int64_t n0, int64_t nchunk, int32_t nbins, int dtype,
                              void* out, void* stream) {
    if (!x || !centres || !sigma || !out || B < 1 || B > 65535 || N < 1 || nbins < 1) return CHX_ERR_INVALID_ARG;
    if (n0 < 0 || nchunk < 1 || n0 + nchunk > N || (col != 0 && col != 2)) return CHX_ERR_INVALID_ARG;
    if (!chx_bcast_ok(Bx, B) || (charge && !chx_bcast_ok(Bq, B)) || (survival && !chx_bcast_ok(Bs, B)) ||
        (shift && !chx_bcast_ok(Bsh, B)))
        return CHX_ERR_INVALID_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int64_t total = nchunk * (int64_t)nbins;
    int64_t g = (total + CHX_BLOCK - 1) / CHX_BLOCK;
    const int64_t cap = 65536 / B > 0 ? 65536 / B : 1;
    if (g > cap) g = cap;
    const dim3 grid((unsigned)g, (unsigned)B);
    if (dtype == CHX_F32)
        hipLaunchKernelGGL(kde_values_kernel<float>, grid, dim3(CHX_BLOCK), 0, s, (const float*)x, (const float*)charge,
                           (const float*)survival, (const float*)shift, (const float*)centres, (const float*)sigma, col, Bx,
                           Bq, Bs, Bsh, N, n0, nchunk, (int)nbins, (float*)out);
    else if (dtype == CHX_F64)
        hipLaunchKernelGGL(kde_values_kernel<double>, grid, dim3(CHX_BLOCK), 0, s, (const double*)x, (const double*)charge,
                           (const double*)survival, (const double*)shift, (const double*)centres, (const double*)sigma, col,
                           Bx, Bq, Bs, Bsh, N, n0, nchunk, (int)nbins, (double*)out);
    else
        return CHX_ERR_DTYPE;
    CHX_CHECK_LAUNCH();
    return CHX_OK;
}

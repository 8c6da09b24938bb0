// chx_aperture.hip — Aperture survival mask (cheetah/accelerator/aperture.py:90-135), SURVEY section 8 row f3.
//
// survival_out[b][n] = survival_in[b][n] * inside(x[b][n], y[b][n]) with
//   rectangular: -x_max < x < x_max  and  -y_max < y < y_max           (strict, aperture.py:106-115)
//   elliptical : x^2 / x_max^2 + y^2 / y_max^2 <= 1                     (aperture.py:116-120)
// evaluated in the storage dtype, operation by operation (-ffp-contract=off), so particles sitting on the boundary
// fall on the same side as in the reference. One streaming pass: the x and y columns of the AoS rows (the whole
// 28-/56-byte row travels through the cache line anyway) + 4/8 B survival in and out per particle.
#include "chx_common.h"

namespace {

template <typename T>
__global__ __launch_bounds__(CHX_BLOCK) void aperture_kernel(const T* __restrict__ x, const T* __restrict__ surv,
                                                             const T* __restrict__ limits, int elliptical, int64_t B,
                                                             int64_t Bx, int64_t Bs, int64_t Bl, int64_t N,
                                                             T* __restrict__ out) {
    const int64_t b = blockIdx.y;
    const T* __restrict__ xb = x + ((Bx == 1) ? 0 : b) * N * 7;
    const T* __restrict__ sb = surv ? surv + ((Bs == 1) ? 0 : b) * N : nullptr;
    const T x_max = limits[((Bl == 1) ? 0 : b) * 2 + 0], y_max = limits[((Bl == 1) ? 0 : b) * 2 + 1];
    const T x_max2 = x_max * x_max, y_max2 = y_max * y_max;
    T* __restrict__ ob = out + b * N;
    for (int64_t n = (int64_t)blockIdx.x * CHX_BLOCK + threadIdx.x; n < N; n += (int64_t)gridDim.x * CHX_BLOCK) {
        const T px = xb[n * 7 + 0], py = xb[n * 7 + 2];
        bool inside;
        if (elliptical) {
            const T a = (px * px) / x_max2;
            const T c = (py * py) / y_max2;
            inside = (a + c) <= (T)1;
        } else {
            inside = (px > -x_max) && (px < x_max) && (py > -y_max) && (py < y_max);
        }
        const T s = sb ? sb[n] : (T)1;
        ob[n] = s * (inside ? (T)1 : (T)0);
    }
}

}  // namespace

extern "C" int chx_aperture_mask(const void* x_in, const void* survival_in, const void* limits, int shape, int64_t B,
                                 int64_t Bx, int64_t Bs, int64_t Bl, int64_t N, int dtype, void* survival_out,
                                 void* stream) {
    if (dtype != CHX_F32 && dtype != CHX_F64) return CHX_ERR_DTYPE;
    if (shape != CHX_APERTURE_RECTANGULAR && shape != CHX_APERTURE_ELLIPTICAL) return CHX_ERR_INVALID_ARG;
    if (B < 0 || N < 0 || B > 65535) return CHX_ERR_INVALID_ARG;
    if (B == 0 || N == 0) return CHX_OK;
    if (!x_in || !limits || !survival_out) return CHX_ERR_INVALID_ARG;
    if (!chx_bcast_ok(Bx, B) || !chx_bcast_ok(Bl, B) || (survival_in && !chx_bcast_ok(Bs, B))) return CHX_ERR_INVALID_ARG;
    const int gx = chx_grid_for(N, CHX_BLOCK * 4, 4096);
    dim3 grid((unsigned)gx, (unsigned)B);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == CHX_F32)
        hipLaunchKernelGGL(aperture_kernel<float>, grid, dim3(CHX_BLOCK), 0, s, (const float*)x_in,
                           (const float*)survival_in, (const float*)limits, shape == CHX_APERTURE_ELLIPTICAL, B, Bx, Bs, Bl,
                           N, (float*)survival_out);
    else
        hipLaunchKernelGGL(aperture_kernel<double>, grid, dim3(CHX_BLOCK), 0, s, (const double*)x_in,
                           (const double*)survival_in, (const double*)limits, shape == CHX_APERTURE_ELLIPTICAL, B, Bx, Bs,
                           Bl, N, (double*)survival_out);
    CHX_CHECK_LAUNCH();
    return CHX_OK;
}

// chx_spacecharge_bwd.hip — derivative kernels of the space-charge kick (the reference differentiates
// cheetah/accelerator/space_charge_kick.py with torch autograd; tests/test_space_charge_kick.py:202-327 rely on it).
//
//   chx_sc_igf_table_grad    d(corner table of the integrated Green function)/d(dx, dy, dtau): three tables that go
//                            through the same Green-spectrum + convolution kernels as the function itself, so that
//                            dL/d(cell) = sum dphi . conv(rho, dG/d(cell)) needs no correlation pass
//   chx_sc_gradient_bwd      adjoint of the central-difference field kernel
//   chx_sc_gather_kick_bwd   backward of the fused SI conversion + trilinear gather + kick: forward-mode dual numbers,
//                            one seeded evaluation per input (6 coordinates, 3 + 3 grid parameters, dt, energy), the
//                            force-grid cotangent scattered with atomics
#include "chx_common.h"
#include "chx_sc_math.h"

namespace {

inline dim3 bwd_cell_grid(int64_t n, int64_t B) {
    int64_t g = (n + CHX_BLOCK - 1) / CHX_BLOCK;
    int64_t cap = 16384 / B;
    if (cap < 1) cap = 1;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return dim3((unsigned)g, (unsigned)B);
}

bool bwd_bins_ok(const int32_t* bins) {
    return bins && bins[0] >= 2 && bins[1] >= 2 && bins[2] >= 2 && bins[0] <= 1024 && bins[1] <= 1024 && bins[2] <= 1024;
}

// tables[d][b][i][j][k] = d F((i-1/2) dx, (j-1/2) dy, (k-1/2) dt) / d (dx, dy, dt)[d]
template <typename T>
__global__ __launch_bounds__(CHX_BLOCK) void igf_table_grad_kernel(const T* __restrict__ cell, const T* __restrict__ gamma,
                                                                  int gx, int gy, int gz, int64_t B,
                                                                  double* __restrict__ tables) {
    const int64_t b = blockIdx.y;
    const int64_t npts = (int64_t)(gx + 1) * (gy + 1) * (gz + 1);
    const double dx = (double)cell[b * 3 + 0], dy = (double)cell[b * 3 + 1];
    const double dt = (double)(T)(cell[b * 3 + 2] * gamma[b]);
    for (int64_t idx = (int64_t)blockIdx.x * CHX_BLOCK + threadIdx.x; idx < npts; idx += (int64_t)gridDim.x * CHX_BLOCK) {
        const int k = (int)(idx % (gz + 1));
        const int j = (int)((idx / (gz + 1)) % (gy + 1));
        const int i = (int)(idx / ((int64_t)(gz + 1) * (gy + 1)));
        const double f[3] = {i - 0.5, j - 0.5, k - 0.5};
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const Dual r = igf_primitive<Dual>(mk(f[0] * dx, d == 0 ? f[0] : 0.0), mk(f[1] * dy, d == 1 ? f[1] : 0.0),
                                               mk(f[2] * dt, d == 2 ? f[2] : 0.0));
            tables[((int64_t)d * B + b) * npts + idx] = r.d;
        }
    }
}

// F_d[c] = coef_d (phi[c + e_d] - phi[c - e_d]) for 0 < c_d < g_d - 1, coef_d = -(1/gamma^2) 0.5 / cell_d
template <typename T>
__global__ __launch_bounds__(CHX_BLOCK) void gradient_bwd_kernel(const T* __restrict__ dF, const T* __restrict__ cell,
                                                                const T* __restrict__ gamma, int gx, int gy, int gz,
                                                                T* __restrict__ dphi) {
    const int64_t b = blockIdx.y;
    const int64_t ncell = (int64_t)gx * gy * gz;
    const T gm = gamma[b];
    const T ig2 = (gm != (T)0) ? (T)1 / (gm * gm) : (T)0;
    const T cf[3] = {-ig2 * ((T)0.5 * ((T)1 / cell[b * 3 + 0])), -ig2 * ((T)0.5 * ((T)1 / cell[b * 3 + 1])),
                     -ig2 * ((T)0.5 * ((T)1 / cell[b * 3 + 2]))};
    const int g[3] = {gx, gy, gz};
    const int64_t st[3] = {(int64_t)gy * gz, gz, 1};
    const T* dFb = dF + b * ncell * 4;
    for (int64_t idx = (int64_t)blockIdx.x * CHX_BLOCK + threadIdx.x; idx < ncell; idx += (int64_t)gridDim.x * CHX_BLOCK) {
        const int c[3] = {(int)(idx / st[0]), (int)((idx / gz) % gy), (int)(idx % gz)};
        T acc = (T)0;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const int lo = c[d] - 1, hi = c[d] + 1;  // cells whose F_d reads phi[c]
            if (lo > 0 && lo < g[d] - 1) acc += cf[d] * dFb[(idx - st[d]) * 4 + d];
            if (hi > 0 && hi < g[d] - 1) acc -= cf[d] * dFb[(idx + st[d]) * 4 + d];
        }
        dphi[b * ncell + idx] = acc;
    }
}

// One particle of sc_particle_kernel<T, 0>, templated: v (cheetah coordinates) -> kicked cheetah coordinates. Also
// returns the interpolated, un-kicked SI state needed by the force-grid scatter.
template <typename S>
__device__ __forceinline__ void gather_kick(const RefFrame<S>& rf, const S (&v)[7], const double* __restrict__ Fb4,
                                            const S (&half)[3], const S (&cell)[3], S dt, const int (&g)[3], S (&out)[7]) {
    S s[7];
    to_si<S>(rf, v, s);
    const S pos[3] = {s[0], s[2], s[4]};
    S u[3];
    int i0[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        u[d] = (pos[d] + half[d]) / cell[d];
        double fl = floor(val(u[d]));
        fl = fl > 2.0e9 ? 2.0e9 : (fl < -2.0e9 ? -2.0e9 : fl);
        i0[d] = (int)fl;
    }
    S fx = cst<S>(0.0), fy = cst<S>(0.0), fz = cst<S>(0.0);
#pragma unroll
    for (int ox = 0; ox < 2; ++ox)
#pragma unroll
        for (int oy = 0; oy < 2; ++oy)
#pragma unroll
            for (int oz = 0; oz < 2; ++oz) {
                const int ix = i0[0] + ox, iy = i0[1] + oy, iz = i0[2] + oz;
                const bool valid = ix >= 0 && ix < g[0] && iy >= 0 && iy < g[1] && iz >= 0 && iz < g[2];
                if (valid) {
                    const S w = (1.0 - m_abs(u[0] - (double)ix)) * (1.0 - m_abs(u[1] - (double)iy)) *
                                (1.0 - m_abs(u[2] - (double)iz)) * kElementaryCharge;
                    const double* f4 = Fb4 + (((int64_t)ix * g[1] + iy) * g[2] + iz) * 4;
                    fx = fx + w * f4[0];
                    fy = fy + w * f4[1];
                    fz = fz + w * f4[2];
                }
            }
    s[1] = s[1] + fx * dt;
    s[3] = s[3] + fy * dt;
    s[5] = s[5] + fz * dt;
    from_si<S>(rf, s, out);
}

enum { SCB_HALF = 0, SCB_CELL = 3, SCB_DT = 6, SCB_ENERGY = 7, SCB_NP = 8 };

template <typename T>
__device__ __forceinline__ void atomic_add_T(T* p, double v);
template <> __device__ __forceinline__ void atomic_add_T<float>(float* p, double v) { unsafeAtomicAdd(p, (float)v); }
template <> __device__ __forceinline__ void atomic_add_T<double>(double* p, double v) { unsafeAtomicAdd(p, v); }

// The force grid is read through a per-row fp64 copy-free view: F is stored in T, converted on load.
template <typename T>
__global__ __launch_bounds__(CHX_BLOCK) void sc_particle_bwd_kernel(
    const T* __restrict__ x_in, const T* __restrict__ F, const T* __restrict__ half, const T* __restrict__ cell,
    const T* __restrict__ energy, const T* __restrict__ dt, const T* __restrict__ dY, double mass_eV, int64_t Bx,
    int64_t Be, int64_t N, int gx, int gy, int gz, T* __restrict__ dx, T* __restrict__ dF, double* __restrict__ partials) {
    __shared__ double red[CHX_BLOCK / 64];
    const int64_t b = blockIdx.y;
    const int64_t tiles = gridDim.x;
    const int64_t n = (int64_t)blockIdx.x * CHX_BLOCK + threadIdx.x;
    const bool live = n < N;
    const int64_t xrow = (Bx == 1) ? 0 : b;
    const double E = (double)energy[Be == 1 ? 0 : b];
    const double hv[3] = {(double)half[b * 3], (double)half[b * 3 + 1], (double)half[b * 3 + 2]};
    const double cv[3] = {(double)cell[b * 3], (double)cell[b * 3 + 1], (double)cell[b * 3 + 2]};
    const double dtv = (double)dt[b];
    const int64_t ncell = (int64_t)gx * gy * gz;
    const T* Fb = F + b * ncell * 4;

    double xv[7], gy_[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        xv[j] = live ? (double)x_in[(xrow * N + n) * 7 + j] : 0.0;
        gy_[j] = live ? (double)dY[(b * N + n) * 7 + j] : 0.0;
    }

    // the 8 corner force vectors of this particle, fetched once (positions do not depend on the seeds' tangents)
    double Fc[8 * 4];
    int i0[3] = {0, 0, 0};
    {
        const RefFrame<double> rf = ref_frame<double>(E, mass_eV);
        double s[7];
        to_si<double>(rf, xv, s);
        const double pos[3] = {s[0], s[2], s[4]};
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            double fl = floor((pos[d] + hv[d]) / cv[d]);
            fl = fl > 2.0e9 ? 2.0e9 : (fl < -2.0e9 ? -2.0e9 : fl);
            i0[d] = (int)fl;
        }
    }
    // local 2x2x2 force block laid out like a g = (2,2,2) grid with origin i0 (invalid corners hold zeros: they
    // contribute nothing in the forward pass either)
    bool cvalid[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int ix = i0[0] + (c >> 2), iy = i0[1] + ((c >> 1) & 1), iz = i0[2] + (c & 1);
        cvalid[c] = live && ix >= 0 && ix < gx && iy >= 0 && iy < gy && iz >= 0 && iz < gz;
#pragma unroll
        for (int k = 0; k < 4; ++k) Fc[c * 4 + k] = 0.0;
        if (cvalid[c]) {
            const T* f4 = Fb + (((int64_t)ix * gy + iy) * gz + iz) * 4;
            Fc[c * 4 + 0] = (double)f4[0];
            Fc[c * 4 + 1] = (double)f4[1];
            Fc[c * 4 + 2] = (double)f4[2];
        }
    }
    // evaluate with the local block: shift `half` so that floor((pos + half') / cell) lands on local index 0
    auto eval = [&](int seed, double (&tang)[7]) {
        // seed: 0..5 coordinate, 6..8 half, 9..11 cell, 12 dt, 13 energy
        Dual v[7], hf[3], cl[3], out[7];
#pragma unroll
        for (int j = 0; j < 7; ++j) v[j] = mk(xv[j], (j < 6 && seed == j) ? 1.0 : 0.0);
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            cl[d] = mk(cv[d], seed == 9 + d ? 1.0 : 0.0);
            // u_local = u - i0  <=>  half_local = half - i0 * cell
            hf[d] = mk(hv[d], seed == 6 + d ? 1.0 : 0.0) - (double)i0[d] * cl[d];
        }
        const Dual dtd = mk(dtv, seed == 12 ? 1.0 : 0.0);
        const RefFrame<Dual> rf = ref_frame<Dual>(mk(E, seed == 13 ? 1.0 : 0.0), mass_eV);
        const int g2[3] = {2, 2, 2};
        gather_kick<Dual>(rf, v, Fc, hf, cl, dtd, g2, out);
#pragma unroll
        for (int j = 0; j < 7; ++j) tang[j] = out[j].d;
    };

    if (dx) {
        if (live) {
#pragma unroll 1
            for (int m = 0; m < 6; ++m) {
                double tg[7];
                eval(m, tg);
                double acc = 0.0;
#pragma unroll
                for (int j = 0; j < 7; ++j) acc += gy_[j] * tg[j];
                dx[(b * N + n) * 7 + m] = (T)acc;
            }
            dx[(b * N + n) * 7 + 6] = (T)gy_[6];  // column 6 passes through unchanged
        }
    }
    if (partials) {
#pragma unroll 1
        for (int k = 0; k < SCB_NP; ++k) {
            double acc = 0.0;
            if (live) {
                double tg[7];
                eval(6 + k, tg);
#pragma unroll
                for (int j = 0; j < 7; ++j) acc += gy_[j] * tg[j];
            }
            acc = chx_wave_sum(acc);
            if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
            __syncthreads();
            if (threadIdx.x == 0) {
                double tot = 0.0;
                for (int w = 0; w < CHX_BLOCK / 64; ++w) tot += red[w];
                partials[(b * tiles + blockIdx.x) * SCB_NP + k] = tot;
            }
            __syncthreads();
        }
    }
    if (dF && live) {
        // cotangent of the kicked SI momenta: gs_c = sum_j dY_j d out_j / d s_c for c in (1, 3, 5), from_si only
        const RefFrame<double> rfd = ref_frame<double>(E, mass_eV);
        double s[7], u[3];
        to_si<double>(rfd, xv, s);
        const double pos[3] = {s[0], s[2], s[4]};
        double f[3] = {0.0, 0.0, 0.0}, wc[8];
#pragma unroll
        for (int d = 0; d < 3; ++d) u[d] = (pos[d] + hv[d]) / cv[d];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int ix = i0[0] + (c >> 2), iy = i0[1] + ((c >> 1) & 1), iz = i0[2] + (c & 1);
            wc[c] = (1.0 - fabs(u[0] - ix)) * (1.0 - fabs(u[1] - iy)) * (1.0 - fabs(u[2] - iz)) * kElementaryCharge;
            if (cvalid[c]) {
                f[0] += wc[c] * Fc[c * 4 + 0];
                f[1] += wc[c] * Fc[c * 4 + 1];
                f[2] += wc[c] * Fc[c * 4 + 2];
            }
        }
        s[1] += f[0] * dtv;
        s[3] += f[1] * dtv;
        s[5] += f[2] * dtv;
        const RefFrame<Dual> rf = ref_frame<Dual>(mk(E, 0.0), mass_eV);
        double gs[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            Dual sd[7], out[7];
#pragma unroll
            for (int j = 0; j < 7; ++j) sd[j] = mk(s[j], j == 1 + 2 * c ? 1.0 : 0.0);
            from_si<Dual>(rf, sd, out);
            double acc = 0.0;
#pragma unroll
            for (int j = 0; j < 7; ++j) acc += gy_[j] * out[j].d;
            gs[c] = acc * dtv;
        }
        T* dFb = dF + b * ncell * 4;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            if (!cvalid[c]) continue;
            const int ix = i0[0] + (c >> 2), iy = i0[1] + ((c >> 1) & 1), iz = i0[2] + (c & 1);
            T* o = dFb + (((int64_t)ix * gy + iy) * gz + iz) * 4;
            atomic_add_T<T>(o + 0, wc[c] * gs[0]);
            atomic_add_T<T>(o + 1, wc[c] * gs[1]);
            atomic_add_T<T>(o + 2, wc[c] * gs[2]);
        }
    }
}

}  // namespace

extern "C" int chx_sc_igf_table_grad(const void* cell, const void* gamma, int64_t B, const int32_t* bins, int dtype,
                                     double* tables, void* stream) {
    if (!cell || !gamma || !tables || B < 1 || B > 65535 || !bwd_bins_ok(bins)) return CHX_ERR_INVALID_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int64_t npts = (int64_t)(bins[0] + 1) * (bins[1] + 1) * (bins[2] + 1);
    if (dtype == CHX_F32)
        hipLaunchKernelGGL(igf_table_grad_kernel<float>, bwd_cell_grid(npts, B), dim3(CHX_BLOCK), 0, s, (const float*)cell,
                           (const float*)gamma, bins[0], bins[1], bins[2], B, tables);
    else if (dtype == CHX_F64)
        hipLaunchKernelGGL(igf_table_grad_kernel<double>, bwd_cell_grid(npts, B), dim3(CHX_BLOCK), 0, s, (const double*)cell,
                           (const double*)gamma, bins[0], bins[1], bins[2], B, tables);
    else
        return CHX_ERR_DTYPE;
    CHX_CHECK_LAUNCH();
    return CHX_OK;
}

extern "C" int chx_sc_gradient_bwd(const void* dF, const void* cell, const void* gamma, int64_t B, const int32_t* bins,
                                   int dtype, void* dphi, void* stream) {
    if (!dF || !cell || !gamma || !dphi || B < 1 || B > 65535 || !bwd_bins_ok(bins)) return CHX_ERR_INVALID_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int64_t ncell = (int64_t)bins[0] * bins[1] * bins[2];
    if (dtype == CHX_F32)
        hipLaunchKernelGGL(gradient_bwd_kernel<float>, bwd_cell_grid(ncell, B), dim3(CHX_BLOCK), 0, s, (const float*)dF,
                           (const float*)cell, (const float*)gamma, bins[0], bins[1], bins[2], (float*)dphi);
    else if (dtype == CHX_F64)
        hipLaunchKernelGGL(gradient_bwd_kernel<double>, bwd_cell_grid(ncell, B), dim3(CHX_BLOCK), 0, s, (const double*)dF,
                           (const double*)cell, (const double*)gamma, bins[0], bins[1], bins[2], (double*)dphi);
    else
        return CHX_ERR_DTYPE;
    CHX_CHECK_LAUNCH();
    return CHX_OK;
}

extern "C" int64_t chx_sc_gather_kick_bwd_partials_count(int64_t B, int64_t N) {
    if (B < 0 || N < 0) return 0;
    return B * ((N + CHX_BLOCK - 1) / CHX_BLOCK) * SCB_NP;
}

extern "C" int chx_sc_gather_kick_bwd(const void* x_in, const void* F, const void* half, const void* cell,
                                      const void* energy, const void* dt, const void* dY, double mass_eV, int64_t B,
                                      int64_t Bx, int64_t Be, int64_t N, const int32_t* bins, int dtype, void* dx,
                                      void* dF, double* partials, void* stream) {
    if (!x_in || !F || !half || !cell || !energy || !dt || !dY || (!dx && !dF && !partials)) return CHX_ERR_INVALID_ARG;
    if (B < 1 || N < 1 || B > 65535 || !bwd_bins_ok(bins)) return CHX_ERR_INVALID_ARG;
    if (!chx_bcast_ok(Bx, B) || !chx_bcast_ok(Be, B)) return CHX_ERR_INVALID_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int64_t tiles = (N + CHX_BLOCK - 1) / CHX_BLOCK;
    if (tiles > 0x7fffffffLL) return CHX_ERR_INVALID_ARG;
    const dim3 grid((unsigned)tiles, (unsigned)B);
    if (dtype == CHX_F32)
        hipLaunchKernelGGL(sc_particle_bwd_kernel<float>, grid, dim3(CHX_BLOCK), 0, s, (const float*)x_in, (const float*)F,
                           (const float*)half, (const float*)cell, (const float*)energy, (const float*)dt,
                           (const float*)dY, mass_eV, Bx, Be, N, bins[0], bins[1], bins[2], (float*)dx, (float*)dF, partials);
    else if (dtype == CHX_F64)
        hipLaunchKernelGGL(sc_particle_bwd_kernel<double>, grid, dim3(CHX_BLOCK), 0, s, (const double*)x_in,
                           (const double*)F, (const double*)half, (const double*)cell, (const double*)energy,
                           (const double*)dt, (const double*)dY, mass_eV, Bx, Be, N, bins[0], bins[1], bins[2],
                           (double*)dx, (double*)dF, partials);
    else
        return CHX_ERR_DTYPE;
    CHX_CHECK_LAUNCH();
    return CHX_OK;
}

// chx_nonlinear.hip — per-particle non-linear tracking for gfx950.
//
//  * drift-kick-drift (Bmad-X) tracking of Drift, Quadrupole, Dipole and TransverseDeflectingCavity:
//    cheetah/accelerator/drift.py:106-154, quadrupole.py:168-251, dipole.py:183-370,
//    transverse_deflecting_cavity.py:122-209 on top of cheetah/utils/bmadx.py;
//  * second-order (MAD-convention T tensor) tracking: element.py:195-228 with the tensors of
//    track_methods.py:80-296 (base_ttensor) dressed per element in drift.py:68-84,
//    quadrupole.py:113-146, dipole.py:397-428 and sextupole.py:91-116.
//
// Both are one streaming pass over particles[B][N][7] (56 B/particle fp32, 112 B fp64) through the same
// LDS tile staging as chx_apply.hip; the per-particle arithmetic runs in fp64 whatever the storage
// dtype is (transcendental-heavy but far below the HBM time of the pass on 256 CUs).
#include "chx_common.h"

namespace {

constexpr double kPi = 3.14159265358979323846;
constexpr double kC = 299792458.0;  // scipy.constants.speed_of_light

// ---------------------------------------------------------------------------------------------
// Bmad-X helpers (utils/bmadx.py)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double sinc1(double x) { return x == 0.0 ? 1.0 : sin(x) / x; }      // bmadx.py:318
__device__ __forceinline__ double cosc1(double x) { const double s = sinc1(0.5 * x); return -0.5 * s * s; }  // :323
__device__ __forceinline__ double sqrt_one(double x) { return x / (sqrt(1.0 + x) + 1.0); }     // :255

struct Bmad {
    double x, px, y, py, z, pz;
};

// bmadx.py:7-31
__device__ __forceinline__ void to_bmad(double tau, double delta, double E, double mc2, double p0c, double& z,
                                        double& pz) {
    const double en = E + delta * p0c;
    const double p = sqrt(en * en - mc2 * mc2);
    const double beta = p / en;
    z = -beta * tau;
    pz = (p - p0c) / p0c;
}
// bmadx.py:34-56
__device__ __forceinline__ void from_bmad(double z, double pz, double p0c, double mc2, double& tau, double& delta) {
    const double ref = sqrt(p0c * p0c + mc2 * mc2);
    const double p = (1.0 + pz) * p0c;
    const double en = sqrt(p * p + mc2 * mc2);
    const double beta = p / en;
    tau = -z / beta;
    delta = (en - ref) / p0c;
}
// bmadx.py:117-147 / 150-181
__device__ __forceinline__ void offset_set(double xo, double yo, double s, double c, Bmad& q) {
    const double xi = q.x - xo, yi = q.y - yo;
    const double x = xi * c + yi * s, y = -xi * s + yi * c;
    const double px = q.px * c + q.py * s, py = -q.px * s + q.py * c;
    q.x = x; q.y = y; q.px = px; q.py = py;
}
__device__ __forceinline__ void offset_unset(double xo, double yo, double s, double c, Bmad& q) {
    const double xi = q.x * c - q.y * s, yi = q.x * s + q.y * c;
    const double px = q.px * c - q.py * s, py = q.px * s + q.py * c;
    q.x = xi + xo; q.y = yi + yo; q.px = px; q.py = py;
}
// bmadx.py:263-298
__device__ __forceinline__ void track_a_drift(double L, Bmad& q, double p0c, double mc2) {
    const double P = 1.0 + q.pz;
    const double Px = q.px / P, Py = q.py / P;
    const double Pxy2 = Px * Px + Py * Py;
    const double Pl = sqrt(1.0 - Pxy2);
    const double pc = p0c * P;
    const double dz =
        L * (sqrt_one((mc2 * mc2 * (2.0 * q.pz + q.pz * q.pz)) / (pc * pc + mc2 * mc2)) + sqrt_one(-Pxy2) / Pl);
    q.x = q.x + L * Px / Pl;
    q.y = q.y + L * Py / Pl;
    q.z = q.z + dz;
}
// bmadx.py:184-216
__device__ __forceinline__ double low_energy_z_correction(double pz, double p0c, double mc2, double ds) {
    const double pc = (1.0 + pz) * p0c;
    const double beta = pc / sqrt(pc * pc + mc2 * mc2);
    const double e_tot = sqrt(p0c * p0c + mc2 * mc2);
    const double beta0 = p0c / e_tot;
    const double b0pz = beta0 * pz;
    const double evaluation = mc2 * (b0pz * b0pz);
    const double me = mc2 / e_tot, me2 = me * me, b02 = beta0 * beta0;
    if (evaluation < 3e-7 * e_tot)
        return ds * pz * (1.0 - 3.0 * (pz * b02) / 2.0 + pz * pz * b02 * (2.0 * b02 - me2 / 2.0)) * me2;
    return ds * (beta - beta0) / beta0;
}
// bmadx.py:219-252 with k1 real: kx = sqrt(-k1) is real (k1 < 0), imaginary (k1 > 0) or zero
struct QuadCoef {
    double a11, a12, a21, a22, c1, c2, c3;
};
__device__ __forceinline__ QuadCoef quad_coefficients(double k1, double len, double rel_p) {
    double cx, sx;
    const double w = -k1;
    if (w > 0.0) {
        const double k = sqrt(w);
        cx = cos(k * len);
        sx = sin(k * len) / k;
    } else if (w < 0.0) {
        const double k = sqrt(-w);
        cx = cosh(k * len);
        sx = sinh(k * len) / k;
    } else {
        cx = 1.0;
        sx = len;
    }
    QuadCoef q;
    q.a11 = cx;
    q.a12 = sx / rel_p;
    q.a21 = k1 * sx * rel_p;
    q.a22 = cx;
    q.c1 = k1 * (-cx * sx + len) / 4.0;
    q.c2 = -k1 * (sx * sx) / (2.0 * rel_p);
    q.c3 = -(cx * sx + len) / (4.0 * (rel_p * rel_p));
    return q;
}
// utils/autograd.py:669-670
__device__ __forceinline__ double sqrta2minusbdiva(double a, double b) {
    return b != 0.0 ? (sqrt(a * a + b) - a) / b : 1.0 / (2.0 * a);
}

// ---------------------------------------------------------------------------------------------
// per-batch-row constants (computed once per workgroup by lane 0, shared through LDS)
// ---------------------------------------------------------------------------------------------
enum { C_E = 0, C_P0C, C_SIN, C_COS, C_XO, C_YO, C_A, C_B, C_C, C_D, C_E2, C_F, C_G, C_H, C_I, C_J, C_N };

template <typename T>
__device__ void dkd_constants(int kind, const T* __restrict__ par, double E, double mc2, double nq, int fringe,
                              double* c) {
    c[C_E] = E;
    const double p0c = sqrt(E * E - mc2 * mc2);
    c[C_P0C] = p0c;
    c[C_SIN] = 0.0; c[C_COS] = 1.0; c[C_XO] = 0.0; c[C_YO] = 0.0;
    if (kind == CHX_DKD_DRIFT) {
        c[C_A] = (double)par[0];
    } else if (kind == CHX_DKD_QUADRUPOLE) {
        const double L = (double)par[0], k1 = (double)par[1], tilt = (double)par[2];
        c[C_SIN] = sin(tilt); c[C_COS] = cos(tilt);
        c[C_XO] = (double)par[3]; c[C_YO] = (double)par[4];
        c[C_A] = L;
        c[C_B] = k1 * L;  // b1 (quadrupole.py:199)
    } else if (kind == CHX_DKD_DIPOLE) {
        const double L = (double)par[0], angle = (double)par[1], e1 = (double)par[2], e2 = (double)par[3];
        const double tilt = (double)par[4], fint = (double)par[5], fintx = (double)par[6];
        const double gap = (double)par[7], gapx = (double)par[8];
        c[C_SIN] = sin(tilt); c[C_COS] = cos(tilt);
        const double g = angle / L;
        c[C_A] = L; c[C_B] = angle; c[C_C] = g;
        c[C_D] = sinc1(angle); c[C_E2] = cosc1(angle); c[C_F] = cos(angle); c[C_G] = sin(angle);
        // linear fringe kicks (dipole.py:355-366): hx, hy at entrance and exit
        const double hg1 = 0.5 * gap, hg2 = 0.5 * gapx;
        const double s1 = sin(e1), s2 = sin(e2);
        c[C_H] = (fringe & 1) ? g * tan(e1) : 0.0;
        c[C_I] = (fringe & 1) ? -g * tan(e1 - 2.0 * fint * hg1 * g * (1.0 + s1 * s1) / cos(e1)) : 0.0;
        c[C_J] = (fringe & 2) ? g * tan(e2) : 0.0;
        c[C_N] = (fringe & 2) ? -g * tan(e2 - 2.0 * fintx * hg2 * g * (1.0 + s2 * s2) / cos(e2)) : 0.0;
    } else {  // CHX_DKD_TDC
        const double L = (double)par[0], V = (double)par[1], phase = (double)par[2], freq = (double)par[3];
        const double tilt = (double)par[4];
        c[C_SIN] = sin(tilt); c[C_COS] = cos(tilt);
        c[C_XO] = (double)par[5]; c[C_YO] = (double)par[6];
        c[C_A] = L;
        c[C_B] = V * -1.0 * nq / p0c;        // transverse_deflecting_cavity.py:156
        c[C_C] = 2.0 * kPi * freq / kC;      // k_rf
        c[C_D] = phase; c[C_E2] = freq;
    }
}

// dipole.py:246-336
__device__ __forceinline__ void dipole_body(const double* c, Bmad& q, double p0c, double mc2) {
    const double L = c[C_A], angle = c[C_B], g = c[C_C], sinc_a = c[C_D], cosc_a = c[C_E2];
    const double cos_a = c[C_F], sin_a = c[C_G];
    const double px_norm = sqrt((1.0 + q.pz) * (1.0 + q.pz) - q.py * q.py);
    const double phi1 = asin(q.px / px_norm);
    const double gp = g / px_norm;
    const double gx1 = 1.0 + g * q.x;
    const double sap = sin(angle + phi1), cap = cos(angle + phi1);
    const double t = gx1 * L * sinc_a;
    const double alpha = 2.0 * gx1 * sap * L * sinc_a - gp * (t * t);
    const double x2_t1 = q.x * cos_a + L * L * g * cosc_a;
    const double x2_t2 = sqrt(cap * cap + gp * alpha);
    const double x2_t3 = cap;
    const double x2 = (fabs(angle + phi1) < kPi / 2.0) ? x2_t1 + alpha / (x2_t2 + x2_t3)
                                                     : x2_t1 + alpha * sqrta2minusbdiva(x2_t3, gp * alpha);
    const double Lcu = x2 - L * L * g * cosc_a - q.x * cos_a;
    const double Lcv = -L * sinc_a - q.x * sin_a;
    const double theta_p = 2.0 * (angle + phi1 - kPi / 2.0 - atan2(Lcv, Lcu));
    const double Lc = sqrt(Lcu * Lcu + Lcv * Lcv);
    const double Lp = Lc / sinc1(theta_p / 2.0);
    const double P = p0c * (1.0 + q.pz);
    const double E = sqrt(P * P + mc2 * mc2);
    const double E0 = sqrt(p0c * p0c + mc2 * mc2);
    const double beta = P / E, beta0 = p0c / E0;
    q.x = x2;
    q.px = px_norm * sin(angle + phi1 - theta_p);
    q.y = q.y + q.py * Lp / px_norm;
    q.z = q.z + (beta * L / beta0) - ((1.0 + q.pz) * Lp / px_norm);
}

template <int KIND>
__device__ __forceinline__ void dkd_particle(const double* c, Bmad& q, double mc2, int num_steps) {
    const double p0c = c[C_P0C];
    if (KIND == CHX_DKD_DRIFT) {
        track_a_drift(c[C_A], q, p0c, mc2);
    } else if (KIND == CHX_DKD_QUADRUPOLE) {
        const double L = c[C_A], b1 = c[C_B];
        const double step = L / (double)num_steps;
        offset_set(c[C_XO], c[C_YO], c[C_SIN], c[C_COS], q);
        // pz does not change inside the magnet, so the per-step coefficients are the same for every step
        const double rel_p = 1.0 + q.pz;
        const double k1 = b1 / (L * rel_p);
        const QuadCoef tx = quad_coefficients(-k1, step, rel_p);
        const QuadCoef ty = quad_coefficients(k1, step, rel_p);
        const double dzc = low_energy_z_correction(q.pz, p0c, mc2, step);
        for (int s = 0; s < num_steps; ++s) {
            q.z = q.z + tx.c1 * (q.x * q.x) + tx.c2 * q.x * q.px + tx.c3 * (q.px * q.px) + ty.c1 * (q.y * q.y) +
                  ty.c2 * q.y * q.py + ty.c3 * (q.py * q.py);
            const double xn = tx.a11 * q.x + tx.a12 * q.px, pxn = tx.a21 * q.x + tx.a22 * q.px;
            const double yn = ty.a11 * q.y + ty.a12 * q.py, pyn = ty.a21 * q.y + ty.a22 * q.py;
            q.x = xn; q.px = pxn; q.y = yn; q.py = pyn;
            q.z = q.z + dzc;
        }
        offset_unset(c[C_XO], c[C_YO], c[C_SIN], c[C_COS], q);
    } else if (KIND == CHX_DKD_DIPOLE) {
        offset_set(0.0, 0.0, c[C_SIN], c[C_COS], q);
        q.px = q.px + q.x * c[C_H];
        q.py = q.py + q.y * c[C_I];
        dipole_body(c, q, p0c, mc2);
        q.px = q.px + q.x * c[C_J];
        q.py = q.py + q.y * c[C_N];
        offset_unset(0.0, 0.0, c[C_SIN], c[C_COS], q);
    } else {  // TDC, transverse_deflecting_cavity.py:147-193
        const double half = c[C_A] / 2.0, volt = c[C_B], k_rf = c[C_C], phase0 = c[C_D], freq = c[C_E2];
        offset_set(c[C_XO], c[C_YO], c[C_SIN], c[C_COS], q);
        track_a_drift(half, q, p0c, mc2);
        const double pc_old = (1.0 + q.pz) * p0c;
        const double beta_old = pc_old / sqrt(pc_old * pc_old + mc2 * mc2);
        const double time = -q.z / (beta_old * kC);  // bmadx.py:301-310
        const double phase = 2.0 * kPi * (phase0 - time * freq);
        q.px = q.px + volt * sin(phase);
        const double E_old = pc_old / beta_old;
        const double E_new = E_old + volt * cos(phase) * k_rf * q.x * p0c;
        const double pc = sqrt(E_new * E_new - mc2 * mc2);
        const double beta = pc / E_new;
        q.pz = (pc - p0c) / p0c;
        q.z = q.z * beta / beta_old;
        track_a_drift(half, q, p0c, mc2);
        offset_unset(c[C_XO], c[C_YO], c[C_SIN], c[C_COS], q);
    }
}

template <typename T, int KIND>
__global__ __launch_bounds__(CHX_BLOCK) void dkd_kernel(const T* __restrict__ x_in, const T* __restrict__ params,
                                                        const T* __restrict__ energy, double mc2, double nq,
                                                        int num_steps, int fringe, int P, int64_t B, int64_t Bx,
                                                        int64_t Bp, int64_t Be, int64_t N, T* __restrict__ x_out,
                                                        T* __restrict__ energy_out, int in_vec_ok, int out_vec_ok) {
    constexpr int TP = CHX_BLOCK;
    __shared__ __attribute__((aligned(16))) T lds[TP * 7];
    __shared__ double cst[C_N + 1];

    const int64_t tiles_per_row = (N + TP - 1) / TP;
    const int64_t b = blockIdx.x / tiles_per_row;
    const int64_t t = blockIdx.x - b * tiles_per_row;
    const int64_t n0 = t * TP;
    const int np = (int)((N - n0 < TP) ? (N - n0) : TP);
    const int64_t in_row = (Bx == 1) ? 0 : b;
    const bool in_vec = in_vec_ok && (((in_row * N * 7 * (int64_t)sizeof(T)) & 15) == 0);
    const bool out_vec = out_vec_ok && (((b * N * 7 * (int64_t)sizeof(T)) & 15) == 0);

    if (threadIdx.x == 0) {
        const T Eb = energy[(Be == 1) ? 0 : b];
        dkd_constants<T>(KIND, params + ((Bp == 1) ? 0 : b) * P, (double)Eb, mc2, nq, fringe, cst);
        if (t == 0 && energy_out) {
            // ref_energy of bmad_to_cheetah_z_pz (bmadx.py:49), in the storage dtype like the reference
            const T m = (T)mc2;
            const T p0 = sqrt(Eb * Eb - m * m);
            energy_out[b] = sqrt(p0 * p0 + m * m);
        }
    }
    tile_load<T>(x_in + (in_row * N + n0) * 7, lds, np * 7, in_vec);
    __syncthreads();

    const int p = threadIdx.x;
    if (p < np) {
        Bmad q;
        q.x = (double)lds[p * 7 + 0];
        q.px = (double)lds[p * 7 + 1];
        q.y = (double)lds[p * 7 + 2];
        q.py = (double)lds[p * 7 + 3];
        const double tau = (double)lds[p * 7 + 4], delta = (double)lds[p * 7 + 5];
        to_bmad(tau, delta, cst[C_E], mc2, cst[C_P0C], q.z, q.pz);
        dkd_particle<KIND>(cst, q, mc2, num_steps);
        double tau_o, delta_o;
        from_bmad(q.z, q.pz, cst[C_P0C], mc2, tau_o, delta_o);
        lds[p * 7 + 0] = (T)q.x;
        lds[p * 7 + 1] = (T)q.px;
        lds[p * 7 + 2] = (T)q.y;
        lds[p * 7 + 3] = (T)q.py;
        lds[p * 7 + 4] = (T)tau_o;
        lds[p * 7 + 5] = (T)delta_o;
        lds[p * 7 + 6] = (T)1;
    }
    __syncthreads();
    tile_store<T>(x_out + (b * N + n0) * 7, lds, np * 7, out_vec);
}

template <typename T, int KIND>
int launch_dkd(const void* x_in, const void* params, const void* energy, double mc2, double nq, int num_steps,
               int fringe, int P, int64_t B, int64_t Bx, int64_t Bp, int64_t Be, int64_t N, void* x_out,
               void* energy_out, hipStream_t s) {
    const int64_t tiles = ((N + CHX_BLOCK - 1) / CHX_BLOCK) * B;
    if (tiles > 0x7fffffffLL) return CHX_ERR_INVALID_ARG;
    hipLaunchKernelGGL((dkd_kernel<T, KIND>), dim3((unsigned)tiles), dim3(CHX_BLOCK), 0, s, (const T*)x_in,
                       (const T*)params, (const T*)energy, mc2, nq, num_steps, fringe, P, B, Bx, Bp, Be, N,
                       (T*)x_out, (T*)energy_out, (int)chx_aligned16(x_in), (int)chx_aligned16(x_out));
    CHX_CHECK_LAUNCH();
    return CHX_OK;
}

template <typename T>
int dispatch_dkd(int kind, const void* x_in, const void* params, const void* energy, double mc2, double nq,
                 int num_steps, int fringe, int64_t B, int64_t Bx, int64_t Bp, int64_t Be, int64_t N,
                 void* x_out, void* energy_out, hipStream_t s) {
    const int P = chx_dkd_num_params(kind);
    switch (kind) {
        case CHX_DKD_DRIFT:
            return launch_dkd<T, CHX_DKD_DRIFT>(x_in, params, energy, mc2, nq, num_steps, fringe, P, B, Bx, Bp, Be,
                                                N, x_out, energy_out, s);
        case CHX_DKD_QUADRUPOLE:
            return launch_dkd<T, CHX_DKD_QUADRUPOLE>(x_in, params, energy, mc2, nq, num_steps, fringe, P, B, Bx, Bp,
                                                     Be, N, x_out, energy_out, s);
        case CHX_DKD_DIPOLE:
            return launch_dkd<T, CHX_DKD_DIPOLE>(x_in, params, energy, mc2, nq, num_steps, fringe, P, B, Bx, Bp, Be,
                                                 N, x_out, energy_out, s);
        case CHX_DKD_TDC:
            return launch_dkd<T, CHX_DKD_TDC>(x_in, params, energy, mc2, nq, num_steps, fringe, P, B, Bx, Bp, Be, N,
                                              x_out, energy_out, s);
    }
    return CHX_ERR_INVALID_ARG;
}

}  // namespace

extern "C" int chx_dkd_num_params(int kind) {
    switch (kind) {
        case CHX_DKD_DRIFT: return 1;
        case CHX_DKD_QUADRUPOLE: return 5;
        case CHX_DKD_DIPOLE: return 9;
        case CHX_DKD_TDC: return 7;
    }
    return -1;
}

extern "C" int chx_dkd_track(int kind, const void* x_in, const void* params, const void* energy, double mass_eV,
                             double n_charges, int32_t num_steps, int32_t fringe_at, int64_t B, int64_t Bx,
                             int64_t Bp, int64_t Be, int64_t N, int dtype, void* x_out, void* energy_out,
                             void* stream) {
    if (chx_dkd_num_params(kind) < 0) return CHX_ERR_INVALID_ARG;
    if (dtype != CHX_F32 && dtype != CHX_F64) return CHX_ERR_DTYPE;
    if (B < 0 || N < 0) return CHX_ERR_INVALID_ARG;
    if (B == 0 || N == 0) return CHX_OK;
    if (!x_in || !params || !energy || !x_out) return CHX_ERR_INVALID_ARG;
    if (!chx_bcast_ok(Bx, B) || !chx_bcast_ok(Bp, B) || !chx_bcast_ok(Be, B)) return CHX_ERR_INVALID_ARG;
    if (kind == CHX_DKD_QUADRUPOLE && num_steps < 1) return CHX_ERR_INVALID_ARG;
    if (fringe_at < 0 || fringe_at > 3) return CHX_ERR_INVALID_ARG;
    hipStream_t s = (hipStream_t)stream;
    return dtype == CHX_F32 ? dispatch_dkd<float>(kind, x_in, params, energy, mass_eV, n_charges, num_steps, fringe_at,
                                                  B, Bx, Bp, Be, N, x_out, energy_out, s)
                            : dispatch_dkd<double>(kind, x_in, params, energy, mass_eV, n_charges, num_steps,
                                                   fringe_at, B, Bx, Bp, Be, N, x_out, energy_out, s);
}

// ---------------------------------------------------------------------------------------------
// second-order tracking: x_out_i = sum_jk T_ijk x_j x_k (element.py:207-217)
// ---------------------------------------------------------------------------------------------
namespace {

__device__ __forceinline__ float nl_fma(float a, float b, float c) { return fmaf(a, b, c); }
__device__ __forceinline__ double nl_fma(double a, double b, double c) { return fma(a, b, c); }

// The quadratic form is symmetric in (j,k): the 343 coefficients of a batch row are folded once per
// workgroup into U[i][j<=k] = T_ijk + T_ikj (196 doubles in LDS, read wave-uniformly), so a particle costs
// 28 products + 196 FMAs instead of 343 + 49.
template <typename T>
__global__ __launch_bounds__(CHX_BLOCK) void second_order_kernel(const T* __restrict__ x_in, const T* __restrict__ Tt,
                                                                 T* __restrict__ x_out, int64_t B, int64_t Bx,
                                                                 int64_t BT, int64_t N, int in_vec_ok,
                                                                 int out_vec_ok) {
    constexpr int TP = CHX_BLOCK;
    __shared__ __attribute__((aligned(16))) T lds[TP * 7];
    __shared__ T U[7 * 28];  // arithmetic in the storage dtype, like the reference's einsum

    const int64_t tiles_per_row = (N + TP - 1) / TP;
    const int64_t b = blockIdx.x / tiles_per_row;
    const int64_t t = blockIdx.x - b * tiles_per_row;
    const int64_t n0 = t * TP;
    const int np = (int)((N - n0 < TP) ? (N - n0) : TP);
    const int64_t in_row = (Bx == 1) ? 0 : b;
    const bool in_vec = in_vec_ok && (((in_row * N * 7 * (int64_t)sizeof(T)) & 15) == 0);
    const bool out_vec = out_vec_ok && (((b * N * 7 * (int64_t)sizeof(T)) & 15) == 0);

    if (threadIdx.x < 7 * 28) {
        const int i = threadIdx.x / 28;
        int r = threadIdx.x - i * 28, j = 0;
        while (r >= 7 - j) { r -= 7 - j; ++j; }
        const int k = j + r;
        const T* Tb = Tt + ((BT == 1) ? 0 : b) * 343 + i * 49;
        U[threadIdx.x] = (j == k) ? Tb[j * 7 + k] : Tb[j * 7 + k] + Tb[k * 7 + j];
    }
    tile_load<T>(x_in + (in_row * N + n0) * 7, lds, np * 7, in_vec);
    __syncthreads();

    const int p = threadIdx.x;
    if (p < np) {
        T x[7], q[28], y[7];
#pragma unroll
        for (int j = 0; j < 7; ++j) x[j] = lds[p * 7 + j];
        {
            int c = 0;
#pragma unroll
            for (int j = 0; j < 7; ++j)
#pragma unroll
                for (int k = j; k < 7; ++k) q[c++] = x[j] * x[k];
        }
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            T acc = U[i * 28] * q[0];
#pragma unroll
            for (int c = 1; c < 28; ++c) acc = nl_fma(U[i * 28 + c], q[c], acc);
            y[i] = acc;
        }
#pragma unroll
        for (int j = 0; j < 7; ++j) lds[p * 7 + j] = y[j];
    }
    __syncthreads();
    tile_store<T>(x_out + (b * N + n0) * 7, lds, np * 7, out_vec);
}

}  // namespace

extern "C" int chx_apply_second_order(const void* x_in, const void* T, void* x_out, int64_t B, int64_t Bx,
                                      int64_t BT, int64_t N, int dtype, void* stream) {
    if (dtype != CHX_F32 && dtype != CHX_F64) return CHX_ERR_DTYPE;
    if (B < 0 || N < 0) return CHX_ERR_INVALID_ARG;
    if (B == 0 || N == 0) return CHX_OK;
    if (!x_in || !T || !x_out) return CHX_ERR_INVALID_ARG;
    if (!chx_bcast_ok(Bx, B) || !chx_bcast_ok(BT, B)) return CHX_ERR_INVALID_ARG;
    const int64_t tiles = ((N + CHX_BLOCK - 1) / CHX_BLOCK) * B;
    if (tiles > 0x7fffffffLL) return CHX_ERR_INVALID_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == CHX_F32)
        hipLaunchKernelGGL(second_order_kernel<float>, dim3((unsigned)tiles), dim3(CHX_BLOCK), 0, s, (const float*)x_in,
                           (const float*)T, (float*)x_out, B, Bx, BT, N, (int)chx_aligned16(x_in),
                           (int)chx_aligned16(x_out));
    else
        hipLaunchKernelGGL(second_order_kernel<double>, dim3((unsigned)tiles), dim3(CHX_BLOCK), 0, s,
                           (const double*)x_in, (const double*)T, (double*)x_out, B, Bx, BT, N,
                           (int)chx_aligned16(x_in), (int)chx_aligned16(x_out));
    CHX_CHECK_LAUNCH();
    return CHX_OK;
}

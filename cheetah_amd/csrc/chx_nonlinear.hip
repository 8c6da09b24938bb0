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
#include <cstdlib>

#include "chx_common.h"
#include "chx_dual.h"

namespace {

constexpr double kPi = 3.14159265358979323846;
constexpr double kC = 299792458.0;  // scipy.constants.speed_of_light

// ---------------------------------------------------------------------------------------------
// Bmad-X helpers (utils/bmadx.py). Templates over S = double (tracking) or Dual (chx_dkd_track_bwd).
// ---------------------------------------------------------------------------------------------
template <typename S> __device__ __forceinline__ S sinc1(S x) { return val(x) == 0.0 ? cst<S>(1.0) : m_sin(x) / x; }  // bmadx.py:318
template <typename S> __device__ __forceinline__ S cosc1(S x) { const S s = sinc1<S>(0.5 * x); return -0.5 * s * s; }  // :323
template <typename S> __device__ __forceinline__ S sqrt_one(S x) { return x / (m_sqrt(1.0 + x) + 1.0); }              // :255

template <typename S>
struct Bmad {
    S x, px, y, py, z, pz;
};

// bmadx.py:7-31
template <typename S>
__device__ __forceinline__ void to_bmad(S tau, S delta, S E, double mc2, S p0c, S& z, S& pz) {
    const S en = E + delta * p0c;
    const S p = m_sqrt(en * en - mc2 * mc2);
    const S beta = p / en;
    z = -beta * tau;
    pz = (p - p0c) / p0c;
}
// bmadx.py:34-56
template <typename S>
__device__ __forceinline__ void from_bmad(S z, S pz, S p0c, double mc2, S& tau, S& delta) {
    const S ref = m_sqrt(p0c * p0c + mc2 * mc2);
    const S p = (1.0 + pz) * p0c;
    const S en = m_sqrt(p * p + mc2 * mc2);
    const S beta = p / en;
    tau = -z / beta;
    delta = (en - ref) / p0c;
}
// bmadx.py:117-147 / 150-181
template <typename S>
__device__ __forceinline__ void offset_set(S xo, S yo, S s, S c, Bmad<S>& q) {
    const S xi = q.x - xo, yi = q.y - yo;
    const S x = xi * c + yi * s, y = -xi * s + yi * c;
    const S px = q.px * c + q.py * s, py = -q.px * s + q.py * c;
    q.x = x; q.y = y; q.px = px; q.py = py;
}
template <typename S>
__device__ __forceinline__ void offset_unset(S xo, S yo, S s, S c, Bmad<S>& q) {
    const S xi = q.x * c - q.y * s, yi = q.x * s + q.y * c;
    const S px = q.px * c - q.py * s, py = q.px * s + q.py * c;
    q.x = xi + xo; q.y = yi + yo; q.px = px; q.py = py;
}
// bmadx.py:263-298
template <typename S>
__device__ __forceinline__ void track_a_drift(S L, Bmad<S>& q, S p0c, double mc2) {
    const S P = 1.0 + q.pz;
    const S Px = q.px / P, Py = q.py / P;
    const S Pxy2 = Px * Px + Py * Py;
    const S Pl = m_sqrt(1.0 - Pxy2);
    const S pc = p0c * P;
    const S dz =
        L * (sqrt_one<S>((mc2 * mc2 * (2.0 * q.pz + q.pz * q.pz)) / (pc * pc + mc2 * mc2)) + sqrt_one<S>(-Pxy2) / Pl);
    q.x = q.x + L * Px / Pl;
    q.y = q.y + L * Py / Pl;
    q.z = q.z + dz;
}
// bmadx.py:184-216
template <typename S>
__device__ __forceinline__ S low_energy_z_correction(S pz, S p0c, double mc2, S ds) {
    const S pc = (1.0 + pz) * p0c;
    const S beta = pc / m_sqrt(pc * pc + mc2 * mc2);
    const S e_tot = m_sqrt(p0c * p0c + mc2 * mc2);
    const S beta0 = p0c / e_tot;
    const S b0pz = beta0 * pz;
    const S evaluation = mc2 * (b0pz * b0pz);
    const S me = mc2 / e_tot, me2 = me * me, b02 = beta0 * beta0;
    if (val(evaluation) < 3e-7 * val(e_tot))
        return ds * pz * (1.0 - 3.0 * (pz * b02) / 2.0 + pz * pz * b02 * (2.0 * b02 - me2 / 2.0)) * me2;
    return ds * (beta - beta0) / beta0;
}
// bmadx.py:219-252 with k1 real: kx = sqrt(-k1) is real (k1 < 0), imaginary (k1 > 0) or zero
template <typename S>
struct QuadCoef {
    S a11, a12, a21, a22, c1, c2, c3;
};
template <typename S>
__device__ __forceinline__ QuadCoef<S> quad_coefficients(S k1, S len, S rel_p) {
    S cx, sx;
    const S w = -k1;
    if (val(w) > 0.0) {
        const S k = m_sqrt(w);
        cx = m_cos(k * len);
        sx = m_sin(k * len) / k;
    } else if (val(w) < 0.0) {
        const S k = m_sqrt(-w);
        cx = m_cosh(k * len);
        sx = m_sinh(k * len) / k;
    } else {
        cx = cst<S>(1.0);
        sx = len;
    }
    QuadCoef<S> q;
    q.a11 = cx;
    q.a12 = sx / rel_p;
    q.a21 = k1 * sx * rel_p;
    q.a22 = cx;
    q.c1 = k1 * (-cx * sx + len) / 4.0;
    q.c2 = -k1 * (sx * sx) / (2.0 * rel_p);
    q.c3 = -(cx * sx + len) / (4.0 * (rel_p * rel_p));
    return q;
}
// utils/autograd.py:669-670 (value) and :672-700 (derivatives, with the b == 0 limits -1/(2a^2), -1/(8a^3))
__device__ __forceinline__ double sqrta2minusbdiva(double a, double b) {
    return b != 0.0 ? (sqrt(a * a + b) - a) / b : 1.0 / (2.0 * a);
}
__device__ __forceinline__ F32 sqrta2minusbdiva(F32 a, F32 b) {
    return b.v != 0.0f ? mkf((sqrtf(a.v * a.v + b.v) - a.v) / b.v) : mkf(1.0f / (2.0f * a.v));
}
__device__ __forceinline__ Dual sqrta2minusbdiva(Dual a, Dual b) {
    if (b.v != 0.0) return (m_sqrt(a * a + b) - a) / b;
    return mk(1.0 / (2.0 * a.v), -a.d / (2.0 * a.v * a.v) - b.d / (8.0 * a.v * a.v * a.v));
}

// ---------------------------------------------------------------------------------------------
// per-batch-row constants (computed once per workgroup by lane 0, shared through LDS)
// ---------------------------------------------------------------------------------------------
enum { C_E = 0, C_P0C, C_SIN, C_COS, C_XO, C_YO, C_A, C_B, C_C, C_D, C_E2, C_F, C_G, C_H, C_I, C_J, C_N };

template <typename S>
__device__ void dkd_constants(int kind, const S* par, S E, double mc2, double nq, int fringe, S* c) {
    const S zero = cst<S>(0.0), one = cst<S>(1.0);
    c[C_E] = E;
    const S p0c = m_sqrt(E * E - mc2 * mc2);
    c[C_P0C] = p0c;
    c[C_SIN] = zero; c[C_COS] = one; c[C_XO] = zero; c[C_YO] = zero;
    if (kind == CHX_DKD_DRIFT) {
        c[C_A] = par[0];
    } else if (kind == CHX_DKD_QUADRUPOLE) {
        const S L = par[0], k1 = par[1], tilt = par[2];
        c[C_SIN] = m_sin(tilt); c[C_COS] = m_cos(tilt);
        c[C_XO] = par[3]; c[C_YO] = par[4];
        c[C_A] = L;
        c[C_B] = k1 * L;  // b1 (quadrupole.py:199)
    } else if (kind == CHX_DKD_DIPOLE) {
        const S L = par[0], angle = par[1], e1 = par[2], e2 = par[3];
        const S tilt = par[4], fint = par[5], fintx = par[6];
        const S gap = par[7], gapx = par[8];
        c[C_SIN] = m_sin(tilt); c[C_COS] = m_cos(tilt);
        const S g = angle / L;
        c[C_A] = L; c[C_B] = angle; c[C_C] = g;
        c[C_D] = sinc1<S>(angle); c[C_E2] = cosc1<S>(angle); c[C_F] = m_cos(angle); c[C_G] = m_sin(angle);
        // linear fringe kicks (dipole.py:355-366): hx, hy at entrance and exit
        const S hg1 = 0.5 * gap, hg2 = 0.5 * gapx;
        const S s1 = m_sin(e1), s2 = m_sin(e2);
        c[C_H] = (fringe & 1) ? g * m_tan(e1) : zero;
        c[C_I] = (fringe & 1) ? -g * m_tan(e1 - 2.0 * fint * hg1 * g * (1.0 + s1 * s1) / m_cos(e1)) : zero;
        c[C_J] = (fringe & 2) ? g * m_tan(e2) : zero;
        c[C_N] = (fringe & 2) ? -g * m_tan(e2 - 2.0 * fintx * hg2 * g * (1.0 + s2 * s2) / m_cos(e2)) : zero;
    } else {  // CHX_DKD_TDC
        const S L = par[0], V = par[1], phase = par[2], freq = par[3];
        const S tilt = par[4];
        c[C_SIN] = m_sin(tilt); c[C_COS] = m_cos(tilt);
        c[C_XO] = par[5]; c[C_YO] = par[6];
        c[C_A] = L;
        c[C_B] = V * -1.0 * nq / p0c;        // transverse_deflecting_cavity.py:156
        c[C_C] = 2.0 * kPi * freq / kC;      // k_rf
        c[C_D] = phase; c[C_E2] = freq;
    }
}

// dipole.py:246-336
template <typename S>
__device__ __forceinline__ void dipole_body(const S* c, Bmad<S>& q, S p0c, double mc2) {
    const S L = c[C_A], angle = c[C_B], g = c[C_C], sinc_a = c[C_D], cosc_a = c[C_E2];
    const S cos_a = c[C_F], sin_a = c[C_G];
    const S px_norm = m_sqrt((1.0 + q.pz) * (1.0 + q.pz) - q.py * q.py);
    const S phi1 = m_asin(q.px / px_norm);
    const S gp = g / px_norm;
    const S gx1 = 1.0 + g * q.x;
    const S sap = m_sin(angle + phi1), cap = m_cos(angle + phi1);
    const S t = gx1 * L * sinc_a;
    const S alpha = 2.0 * gx1 * sap * L * sinc_a - gp * (t * t);
    const S x2_t1 = q.x * cos_a + L * L * g * cosc_a;
    const S x2_t2 = m_sqrt(cap * cap + gp * alpha);
    const S x2_t3 = cap;
    const S x2 = (fabs(val(angle) + val(phi1)) < kPi / 2.0) ? x2_t1 + alpha / (x2_t2 + x2_t3)
                                                            : x2_t1 + alpha * sqrta2minusbdiva(x2_t3, gp * alpha);
    const S Lcu = x2 - L * L * g * cosc_a - q.x * cos_a;
    const S Lcv = -L * sinc_a - q.x * sin_a;
    const S theta_p = 2.0 * (angle + phi1 - kPi / 2.0 - m_atan2(Lcv, Lcu));
    const S Lc = m_sqrt(Lcu * Lcu + Lcv * Lcv);
    const S Lp = Lc / sinc1<S>(theta_p / 2.0);
    const S P = p0c * (1.0 + q.pz);
    const S E = m_sqrt(P * P + mc2 * mc2);
    const S E0 = m_sqrt(p0c * p0c + mc2 * mc2);
    const S beta = P / E, beta0 = p0c / E0;
    q.x = x2;
    q.px = px_norm * m_sin(angle + phi1 - theta_p);
    q.y = q.y + q.py * Lp / px_norm;
    q.z = q.z + (beta * L / beta0) - ((1.0 + q.pz) * Lp / px_norm);
}

template <int KIND, typename S>
__device__ __forceinline__ void dkd_particle(const S* c, Bmad<S>& q, double mc2, int num_steps) {
    const S p0c = c[C_P0C];
    if (KIND == CHX_DKD_DRIFT) {
        track_a_drift<S>(c[C_A], q, p0c, mc2);
    } else if (KIND == CHX_DKD_QUADRUPOLE) {
        const S L = c[C_A], b1 = c[C_B];
        const S step = L / (double)num_steps;
        offset_set<S>(c[C_XO], c[C_YO], c[C_SIN], c[C_COS], q);
        // pz does not change inside the magnet, so the per-step coefficients are the same for every step
        const S rel_p = 1.0 + q.pz;
        const S k1 = b1 / (L * rel_p);
        const QuadCoef<S> tx = quad_coefficients<S>(-k1, step, rel_p);
        const QuadCoef<S> ty = quad_coefficients<S>(k1, step, rel_p);
        const S dzc = low_energy_z_correction<S>(q.pz, p0c, mc2, step);
        for (int s = 0; s < num_steps; ++s) {
            q.z = q.z + tx.c1 * (q.x * q.x) + tx.c2 * q.x * q.px + tx.c3 * (q.px * q.px) + ty.c1 * (q.y * q.y) +
                  ty.c2 * q.y * q.py + ty.c3 * (q.py * q.py);
            const S xn = tx.a11 * q.x + tx.a12 * q.px, pxn = tx.a21 * q.x + tx.a22 * q.px;
            const S yn = ty.a11 * q.y + ty.a12 * q.py, pyn = ty.a21 * q.y + ty.a22 * q.py;
            q.x = xn; q.px = pxn; q.y = yn; q.py = pyn;
            q.z = q.z + dzc;
        }
        offset_unset<S>(c[C_XO], c[C_YO], c[C_SIN], c[C_COS], q);
    } else if (KIND == CHX_DKD_DIPOLE) {
        const S zero = cst<S>(0.0);
        offset_set<S>(zero, zero, c[C_SIN], c[C_COS], q);
        q.px = q.px + q.x * c[C_H];
        q.py = q.py + q.y * c[C_I];
        dipole_body<S>(c, q, p0c, mc2);
        q.px = q.px + q.x * c[C_J];
        q.py = q.py + q.y * c[C_N];
        offset_unset<S>(zero, zero, c[C_SIN], c[C_COS], q);
    } else {  // TDC, transverse_deflecting_cavity.py:147-193
        const S half = c[C_A] / 2.0, volt = c[C_B], k_rf = c[C_C], phase0 = c[C_D], freq = c[C_E2];
        offset_set<S>(c[C_XO], c[C_YO], c[C_SIN], c[C_COS], q);
        track_a_drift<S>(half, q, p0c, mc2);
        const S pc_old = (1.0 + q.pz) * p0c;
        const S beta_old = pc_old / m_sqrt(pc_old * pc_old + mc2 * mc2);
        const S time = -q.z / (beta_old * kC);  // bmadx.py:301-310
        const S phase = 2.0 * kPi * (phase0 - time * freq);
        q.px = q.px + volt * m_sin(phase);
        const S E_old = pc_old / beta_old;
        const S E_new = E_old + volt * m_cos(phase) * k_rf * q.x * p0c;
        const S pc = m_sqrt(E_new * E_new - mc2 * mc2);
        const S beta = pc / E_new;
        q.pz = (pc - p0c) / p0c;
        q.z = q.z * beta / beta_old;
        track_a_drift<S>(half, q, p0c, mc2);
        offset_unset<S>(c[C_XO], c[C_YO], c[C_SIN], c[C_COS], q);
    }
}

// cheetah coordinates -> Bmad -> element -> cheetah (out[0..5])
template <int KIND, typename S>
__device__ __forceinline__ void dkd_map(const S* c, const S (&in)[6], double mc2, int num_steps, S (&out)[6]) {
    Bmad<S> q;
    q.x = in[0]; q.px = in[1]; q.y = in[2]; q.py = in[3];
    to_bmad<S>(in[4], in[5], c[C_E], mc2, c[C_P0C], q.z, q.pz);
    dkd_particle<KIND, S>(c, q, mc2, num_steps);
    out[0] = q.x; out[1] = q.px; out[2] = q.y; out[3] = q.py;
    from_bmad<S>(q.z, q.pz, c[C_P0C], mc2, out[4], out[5]);
}

// C: the scalar the per-particle map is evaluated in — double (default, whatever the storage dtype) or F32 (float32 beams with
// `precision = storage`: the reference's own arithmetic width, HBM-bound instead of fp64-VALU-bound).
template <typename C> struct dkd_scalar;
template <> struct dkd_scalar<double> {
    __device__ static __forceinline__ double from(double v) { return v; }
    __device__ static __forceinline__ double to(double v) { return v; }
};
template <> struct dkd_scalar<F32> {
    __device__ static __forceinline__ F32 from(double v) { return mkf((float)v); }
    __device__ static __forceinline__ double to(F32 v) { return (double)v.v; }
};

template <typename T, int KIND, typename C = double>
__global__ __launch_bounds__(CHX_BLOCK) void dkd_kernel(const T* __restrict__ x_in, const T* __restrict__ params,
                                                        const T* __restrict__ energy, double mc2, double nq,
                                                        int num_steps, int fringe, int P, int64_t B, int64_t Bx,
                                                        int64_t Bp, int64_t Be, int64_t N, T* __restrict__ x_out,
                                                        T* __restrict__ energy_out, int in_vec_ok, int out_vec_ok) {
    constexpr int TP = CHX_BLOCK;
    __shared__ __attribute__((aligned(16))) T lds[TP * 7];
    __shared__ double cst_[C_N + 1];
    __shared__ C cstc_[C_N + 1];

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
        double par[CHX_MAX_PARAMS];
        for (int k = 0; k < P; ++k) par[k] = (double)params[((Bp == 1) ? 0 : b) * P + k];
        dkd_constants<double>(KIND, par, (double)Eb, mc2, nq, fringe, cst_);      // per-row constants: always in double
        for (int k = 0; k <= C_N; ++k) cstc_[k] = dkd_scalar<C>::from(cst_[k]);
        if (t == 0 && energy_out) {
            // ref_energy of bmad_to_cheetah_z_pz (bmadx.py:49), in the storage dtype like the reference
            const T m = (T)mc2;
            const T p0 = sqrt(Eb * Eb - m * m);
            energy_out[b] = sqrt(p0 * p0 + m * m);
        }
    }
    tile_load<T, TP>(x_in + (in_row * N + n0) * 7, lds, np * 7, in_vec, !(Bx == 1 && B > 1));
    __syncthreads();

    const int p = threadIdx.x;
    if (p < np) {
        C in[6], out[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) in[j] = dkd_scalar<C>::from((double)lds[p * 7 + j]);
        dkd_map<KIND, C>(cstc_, in, mc2, num_steps, out);
#pragma unroll
        for (int j = 0; j < 6; ++j) lds[p * 7 + j] = (T)dkd_scalar<C>::to(out[j]);
        lds[p * 7 + 6] = (T)1;
    }
    __syncthreads();
    tile_store<T, TP>(x_out + (b * N + n0) * 7, lds, np * 7, out_vec, true);
}

// ---- float32 beams, MIXED arithmetic (Drift, Quadrupole): the default of float32 beams --------------------------------------
// Where does a float32 evaluation of the Bmad-X maps lose its digits? Not in the transverse map (x, px, y, py move by parts in
// 1e3 per element; seven significant digits of each step are what the float32 STORAGE keeps anyway) — in the longitudinal pair:
// pz = (p - p0c) / p0c and delta = (E - E_ref) / p0c subtract numbers that agree to three or four digits (bmadx.py:7-56), the
// low-energy correction subtracts beta - beta0 ~ 1e-8 (bmadx.py:184-216), and z collects increments of 1e-9 into a value of
// 1e-5. Measured over the 100 elements of the benchmark lattice: tau off by 1e-2 of its scale, delta by 4e-4 in float32
// arithmetic (the reference's own float32 run: 8e-3 / 4e-5), 5e-7 / 7e-14 in float64 arithmetic — at 2-3 times the kernel time,
// because then the per-particle cos / sin / cosh / sinh of the quadrupole and five square roots run in fp64 as well.
// Here: the (tau, delta) <-> (z, pz) conversions, the z accumulator, the low-energy correction and the misalignment shift run in
// fp64 — one square root and one reciprocal per particle, both from the hardware seeds (delta' is formed from the particle's
// own energy: pz does not change inside these two elements, so the round trip through p is the identity) — everything else in
// float32, including the increments of z (a RELATIVE error of 1e-7 on an increment is 1e-16 of z's scale).
enum { C_ETOT = C_N + 1, C_BETA0, C_ME2, C_INVP0C, C_INVBETA0, C_MIXED_N };
// the extra constants of the mixed evaluation, behind dkd_constants' (lane 0 of a workgroup / of a prepare kernel)
__device__ __forceinline__ void dkd_mixed_constants(double* cst_, double mc2) {
    const double p0c = cst_[C_P0C];
    const double e_tot = sqrt(p0c * p0c + mc2 * mc2);
    cst_[C_ETOT] = e_tot;
    cst_[C_BETA0] = p0c / e_tot;
    cst_[C_ME2] = (mc2 / e_tot) * (mc2 / e_tot);
    cst_[C_INVP0C] = 1.0 / p0c;
    cst_[C_INVBETA0] = e_tot / p0c;
}
// fp64 square root and reciprocal from the hardware seeds (v_rsq_f64 / v_rcp_f64, ~2^-26) and Newton steps in fused multiply-adds:
// a few ulp of a double, a third of the instructions of the correctly rounded library forms. What they feed is rounded to float32
// at the end of the element (2^-24): the extra 2^-51 is invisible there.
__device__ __forceinline__ double dkd_sqrt_fast(double a) {
    double y = __builtin_amdgcn_rsq(a);
    y = y * fma(-0.5 * a, y * y, 1.5);
    y = y * fma(-0.5 * a, y * y, 1.5);
    double s = a * y;
    s = fma(0.5 * y, fma(-s, s, a), s);
    return a == 0.0 ? 0.0 : s;
}
__device__ __forceinline__ double dkd_rcp_fast(double d) {
    double r = __builtin_amdgcn_rcp(d);
    r = fma(r, fma(-d, r, 1.0), r);
    r = fma(r, fma(-d, r, 1.0), r);
    return r;
}
// one particle through one element, v = (x, px, y, py, tau, delta) in float32 before and after
template <int KIND>
__device__ __forceinline__ void dkd_mixed_particle(const double* __restrict__ cst_, double mc2, int num_steps, float (&v)[6]) {
    // A quadrupole that is shifted off the axis (quadrupole.py:199-215 subtracts the misalignment before its map): the shifted
    // coordinate may be hundreds of beam sizes, float32 steps on it lose digits of the BEAM's scale and its path-length terms
    // dwarf tau — such an element is evaluated in float64 like dkd_kernel<float, ., double> (workgroup-uniform decision).
    if (KIND == CHX_DKD_QUADRUPOLE && (cst_[C_XO] != 0.0 || cst_[C_YO] != 0.0)) {
        double in[6], out[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) in[j] = (double)v[j];
        dkd_map<CHX_DKD_QUADRUPOLE, double>(cst_, in, mc2, num_steps, out);
#pragma unroll
        for (int j = 0; j < 6; ++j) v[j] = (float)out[j];
        return;
    }
    const double E = cst_[C_E], p0c = cst_[C_P0C];
    float x = v[0], px = v[1], y = v[2], py = v[3];
    const double tau = (double)v[4], delta = (double)v[5];
    // (tau, delta) -> (z, pz), bmadx.py:7-31, fp64
    const double inv_p0c = cst_[C_INVP0C];
    const double en = E + delta * p0c;
    const double pc = dkd_sqrt_fast(en * en - mc2 * mc2);
    const double r = dkd_rcp_fast(pc * en);
    const double beta = (pc * pc) * r, inv_beta = (en * en) * r;          // pc / en and en / pc from one reciprocal
    double z = -beta * tau;
    const double pz = (pc - p0c) * inv_p0c;
    const float pzf = (float)pz, mc2f = (float)mc2, p0cf = (float)p0c;
    const float relp = 1.0f + pzf;
    if (KIND == CHX_DKD_DRIFT) {                                  // bmadx.py:263-298
        const float L = (float)cst_[C_A];
        const float ir = __builtin_amdgcn_rcpf(relp);
        const float Px = px * ir, Py = py * ir;
        const float Pxy2 = Px * Px + Py * Py;
        const float Pl = __builtin_amdgcn_sqrtf(1.0f - Pxy2);
        const float iPl = __builtin_amdgcn_rcpf(Pl);
        const float pcf = p0cf * relp, m2 = mc2f * mc2f;
        const float a = (m2 * (2.0f * pzf + pzf * pzf)) * __builtin_amdgcn_rcpf(pcf * pcf + m2);
        const float so_a = a * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(1.0f + a) + 1.0f);          // sqrt_one(a)
        const float so_b = -Pxy2 * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(1.0f - Pxy2) + 1.0f);   // sqrt_one(-Pxy2)
        x = x + L * Px * iPl;
        y = y + L * Py * iPl;
        z = z + (double)(L * (so_a + so_b * iPl));
    } else {                                                      // quadrupole.py:168-251, bmadx.py:219-252
        const double xo = cst_[C_XO], yo = cst_[C_YO], sn = cst_[C_SIN], cs = cst_[C_COS];
        // (here xo = yo = 0.) An upright quadrupole — sin(tilt) = 0, cos(tilt) = 1 — is rotated by the identity: x * 1 + y * 0 is
        // x again, so the two rotations are left out (wave-uniform), value for value the same result
        const bool upright = sn == 0.0 && cs == 1.0;
        if (!upright) {   // offset_set: the shift by the misalignment in fp64 (it may be hundreds of beam sizes), the rotation with it
            const double xi = (double)x - xo, yi = (double)y - yo;
            const float xr = (float)(xi * cs + yi * sn), yr = (float)(-xi * sn + yi * cs);
            const float pxr = (float)((double)px * cs + (double)py * sn), pyr = (float)(-(double)px * sn + (double)py * cs);
            x = xr; y = yr; px = pxr; py = pyr;
        }
        const float L = (float)cst_[C_A], b1 = (float)cst_[C_B];
        const float step = L / (float)num_steps;
        const F32 k1 = mkf(b1 * __builtin_amdgcn_rcpf(L * relp));
        const QuadCoef<F32> tx = quad_coefficients<F32>(-k1, mkf(step), mkf(relp));
        const QuadCoef<F32> ty = quad_coefficients<F32>(k1, mkf(step), mkf(relp));
        // low_energy_z_correction (bmadx.py:184-216) in fp64: beta - beta0 cancels to 1e-8
        double dzc;
        {
            const double e_tot = cst_[C_ETOT], beta0 = cst_[C_BETA0], me2 = cst_[C_ME2];
            const double b0pz = beta0 * pz, b02 = beta0 * beta0, ds = (double)step;
            if (mc2 * (b0pz * b0pz) < 3e-7 * e_tot)
                dzc = ds * pz * (1.0 - 3.0 * (pz * b02) / 2.0 + pz * pz * b02 * (2.0 * b02 - me2 / 2.0)) * me2;
            else
                dzc = ds * (beta - beta0) * cst_[C_INVBETA0];     // (beta of this particle: pc / en above)
        }
        for (int s = 0; s < num_steps; ++s) {
            const float dz = tx.c1.v * (x * x) + tx.c2.v * x * px + tx.c3.v * (px * px) + ty.c1.v * (y * y) + ty.c2.v * y * py +
                             ty.c3.v * (py * py);
            const float xn = tx.a11.v * x + tx.a12.v * px, pxn = tx.a21.v * x + tx.a22.v * px;
            const float yn = ty.a11.v * y + ty.a12.v * py, pyn = ty.a21.v * y + ty.a22.v * py;
            x = xn; px = pxn; y = yn; py = pyn;
            z = z + (double)dz + dzc;
        }
        if (!upright) {   // offset_unset
            const double xi = (double)x * cs - (double)y * sn, yi = (double)x * sn + (double)y * cs;
            const float pxr = (float)((double)px * cs - (double)py * sn), pyr = (float)((double)px * sn + (double)py * cs);
            x = (float)(xi + xo); y = (float)(yi + yo); px = pxr; py = pyr;
        }
    }
    // (z, pz) -> (tau, delta), bmadx.py:34-56: pz is unchanged, so p = pc and the particle's energy is `en` again
    const double ref = cst_[C_ETOT];
    v[0] = x; v[1] = px; v[2] = y; v[3] = py;
    v[4] = (float)(-z * inv_beta);
    v[5] = (float)((en - ref) * inv_p0c);
}

template <int KIND>
__global__ __launch_bounds__(CHX_BLOCK) void dkd_mixed_kernel(const float* __restrict__ x_in, const float* __restrict__ params,
                                                              const float* __restrict__ energy, double mc2, double nq, int num_steps,
                                                              int P, int64_t B, int64_t Bx, int64_t Bp, int64_t Be, int64_t N,
                                                              float* __restrict__ x_out, float* __restrict__ energy_out,
                                                              int in_vec_ok, int out_vec_ok) {
    constexpr int TP = CHX_BLOCK;
    __shared__ __attribute__((aligned(16))) float lds[TP * 7];
    __shared__ double cst_[C_MIXED_N];
    const int64_t tiles_per_row = (N + TP - 1) / TP;
    const int64_t b = blockIdx.x / tiles_per_row;
    const int64_t t = blockIdx.x - b * tiles_per_row;
    const int64_t n0 = t * TP;
    const int np = (int)((N - n0 < TP) ? (N - n0) : TP);
    const int64_t in_row = (Bx == 1) ? 0 : b;
    const bool in_vec = in_vec_ok && (((in_row * N * 7 * (int64_t)sizeof(float)) & 15) == 0);
    const bool out_vec = out_vec_ok && (((b * N * 7 * (int64_t)sizeof(float)) & 15) == 0);
    tile_load<float, TP>(x_in + (in_row * N + n0) * 7, lds, np * 7, in_vec, !(Bx == 1 && B > 1));   // rows first: the constants
    if (threadIdx.x == 0) {                                                                          // are formed under the loads
        const float Eb = energy[(Be == 1) ? 0 : b];
        double par[CHX_MAX_PARAMS];
        for (int k = 0; k < P; ++k) par[k] = (double)params[((Bp == 1) ? 0 : b) * P + k];
        dkd_constants<double>(KIND, par, (double)Eb, mc2, nq, 3, cst_);
        dkd_mixed_constants(cst_, mc2);
        if (t == 0 && energy_out) {
            const float m = (float)mc2;
            const float p0 = sqrtf(Eb * Eb - m * m);
            energy_out[b] = sqrtf(p0 * p0 + m * m);
        }
    }
    __syncthreads();
    const int p = threadIdx.x;
    if (p < np) {
        float v[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) v[j] = lds[p * 7 + j];
        dkd_mixed_particle<KIND>(cst_, mc2, num_steps, v);
#pragma unroll
        for (int j = 0; j < 6; ++j) lds[p * 7 + j] = v[j];
        lds[p * 7 + 6] = 1.0f;
    }
    __syncthreads();
    tile_store<float, TP>(x_out + (b * N + n0) * 7, lds, np * 7, out_vec, true);
}

// Backward of the above. dx[n][m] = sum_i dY[n][i] d out_i / d in_m (six passes, coordinate m seeded), and per
// parameter k (then the energy) the workgroup's sum over its particles of dY . d out / d theta_k, written to
// partials[b][tile][k] — summed by the caller (no atomics: deterministic, and 10 same-address atomics per workgroup
// would serialise). Forward-mode costs (6 + P + 1) evaluations per particle; the pass stays far cheaper than the
// reference's autograd graph through ~100 tensor ops per element.
template <typename T, int KIND>
__global__ __launch_bounds__(CHX_BLOCK) void dkd_bwd_kernel(const T* __restrict__ x_in, const T* __restrict__ params,
                                                            const T* __restrict__ energy, const T* __restrict__ dY,
                                                            double mc2, double nq, int num_steps, int fringe, int P,
                                                            int64_t B, int64_t Bx, int64_t Bp, int64_t Be, int64_t N,
                                                            T* __restrict__ dx, double* __restrict__ partials) {
    constexpr int TP = CHX_BLOCK;
    __shared__ Dual cd[C_N + 1];
    __shared__ double red[CHX_BLOCK / 64];

    const int64_t tiles_per_row = (N + TP - 1) / TP;
    const int64_t b = blockIdx.x / tiles_per_row;
    const int64_t t = blockIdx.x - b * tiles_per_row;
    const int64_t n = t * TP + threadIdx.x;
    const bool live = n < N;
    const int64_t in_row = (Bx == 1) ? 0 : b;
    const T Eb = energy[(Be == 1) ? 0 : b];
    const T* par = params + ((Bp == 1) ? 0 : b) * P;

    double xv[6], g[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        xv[j] = live ? (double)x_in[(in_row * N + n) * 7 + j] : 0.0;
        g[j] = live ? (double)dY[(b * N + n) * 7 + j] : 0.0;
    }

    auto constants = [&](int seed) {  // seed in [0, P): parameter, P: energy, -1: none
        if (threadIdx.x == 0) {
            Dual pd[CHX_MAX_PARAMS];
            for (int k = 0; k < P; ++k) pd[k] = mk((double)par[k], k == seed ? 1.0 : 0.0);
            dkd_constants<Dual>(KIND, pd, mk((double)Eb, seed == P ? 1.0 : 0.0), mc2, nq, fringe, cd);
        }
        __syncthreads();
    };

    if (dx) {
        constants(-1);
        if (live) {
#pragma unroll 1
            for (int m = 0; m < 6; ++m) {
                Dual in[6], out[6];
#pragma unroll
                for (int j = 0; j < 6; ++j) in[j] = mk(xv[j], j == m ? 1.0 : 0.0);
                dkd_map<KIND, Dual>(cd, in, mc2, num_steps, out);
                double acc = 0.0;
#pragma unroll
                for (int j = 0; j < 6; ++j) acc += g[j] * out[j].d;
                dx[(b * N + n) * 7 + m] = (T)acc;
            }
            dx[(b * N + n) * 7 + 6] = (T)0;
        }
        __syncthreads();
    }
    if (partials) {
#pragma unroll 1
        for (int k = 0; k <= P; ++k) {
            constants(k);
            double acc = 0.0;
            if (live) {
                Dual in[6], out[6];
#pragma unroll
                for (int j = 0; j < 6; ++j) in[j] = mk(xv[j], 0.0);
                dkd_map<KIND, Dual>(cd, in, mc2, num_steps, out);
#pragma unroll
                for (int j = 0; j < 6; ++j) acc += g[j] * out[j].d;
            }
            acc = chx_wave_sum(acc);
            if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
            __syncthreads();
            if (threadIdx.x == 0) {
                double tot = 0.0;
                for (int w = 0; w < CHX_BLOCK / 64; ++w) tot += red[w];
                partials[(b * tiles_per_row + t) * (P + 1) + k] = tot;
            }
            __syncthreads();
        }
    }
}

template <typename T, int KIND, typename C = double>
int launch_dkd(const void* x_in, const void* params, const void* energy, double mc2, double nq, int num_steps,
               int fringe, int P, int64_t B, int64_t Bx, int64_t Bp, int64_t Be, int64_t N, void* x_out,
               void* energy_out, hipStream_t s) {
    const int64_t tiles = ((N + CHX_BLOCK - 1) / CHX_BLOCK) * B;
    if (tiles > 0x7fffffffLL) return CHX_ERR_INVALID_ARG;
    hipLaunchKernelGGL((dkd_kernel<T, KIND, C>), dim3((unsigned)tiles), dim3(CHX_BLOCK), 0, s, (const T*)x_in,
                       (const T*)params, (const T*)energy, mc2, nq, num_steps, fringe, P, B, Bx, Bp, Be, N,
                       (T*)x_out, (T*)energy_out, (int)chx_aligned16(x_in), (int)chx_aligned16(x_out));
    CHX_CHECK_LAUNCH();
    return CHX_OK;
}

template <typename T, typename C = double>
int dispatch_dkd(int kind, const void* x_in, const void* params, const void* energy, double mc2, double nq,
                 int num_steps, int fringe, int64_t B, int64_t Bx, int64_t Bp, int64_t Be, int64_t N,
                 void* x_out, void* energy_out, hipStream_t s) {
    const int P = chx_dkd_num_params(kind);
    switch (kind) {
        case CHX_DKD_DRIFT:
            return launch_dkd<T, CHX_DKD_DRIFT, C>(x_in, params, energy, mc2, nq, num_steps, fringe, P, B, Bx, Bp, Be,
                                                N, x_out, energy_out, s);
        case CHX_DKD_QUADRUPOLE:
            return launch_dkd<T, CHX_DKD_QUADRUPOLE, C>(x_in, params, energy, mc2, nq, num_steps, fringe, P, B, Bx, Bp,
                                                     Be, N, x_out, energy_out, s);
        case CHX_DKD_DIPOLE:
            return launch_dkd<T, CHX_DKD_DIPOLE, C>(x_in, params, energy, mc2, nq, num_steps, fringe, P, B, Bx, Bp, Be,
                                                 N, x_out, energy_out, s);
        case CHX_DKD_TDC:
            return launch_dkd<T, CHX_DKD_TDC, C>(x_in, params, energy, mc2, nq, num_steps, fringe, P, B, Bx, Bp, Be, N,
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
    return chx_dkd_track_p(kind, x_in, params, energy, mass_eV, n_charges, num_steps, fringe_at, B, Bx, Bp, Be, N, dtype, 0, x_out,
                           energy_out, stream);
}

extern "C" int chx_dkd_track_p(int kind, const void* x_in, const void* params, const void* energy, double mass_eV,
                               double n_charges, int32_t num_steps, int32_t fringe_at, int64_t B, int64_t Bx, int64_t Bp,
                               int64_t Be, int64_t N, int dtype, int storage_precision, void* x_out, void* energy_out,
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
    if (dtype == CHX_F32 && storage_precision == 2 && (kind == CHX_DKD_DRIFT || kind == CHX_DKD_QUADRUPOLE)) {
        // mixed arithmetic (see dkd_mixed_kernel); the other kinds keep the fp64 evaluation
        const int64_t tiles = ((N + CHX_BLOCK - 1) / CHX_BLOCK) * B;
        if (tiles > 0x7fffffffLL) return CHX_ERR_INVALID_ARG;
        const int P = chx_dkd_num_params(kind);
        if (kind == CHX_DKD_DRIFT)
            hipLaunchKernelGGL(dkd_mixed_kernel<CHX_DKD_DRIFT>, dim3((unsigned)tiles), dim3(CHX_BLOCK), 0, s, (const float*)x_in,
                               (const float*)params, (const float*)energy, mass_eV, n_charges, num_steps, P, B, Bx, Bp, Be, N,
                               (float*)x_out, (float*)energy_out, (int)chx_aligned16(x_in), (int)chx_aligned16(x_out));
        else
            hipLaunchKernelGGL(dkd_mixed_kernel<CHX_DKD_QUADRUPOLE>, dim3((unsigned)tiles), dim3(CHX_BLOCK), 0, s, (const float*)x_in,
                               (const float*)params, (const float*)energy, mass_eV, n_charges, num_steps, P, B, Bx, Bp, Be, N,
                               (float*)x_out, (float*)energy_out, (int)chx_aligned16(x_in), (int)chx_aligned16(x_out));
        CHX_CHECK_LAUNCH();
        return CHX_OK;
    }
    if (dtype == CHX_F32 && storage_precision == 1)
        return dispatch_dkd<float, F32>(kind, x_in, params, energy, mass_eV, n_charges, num_steps, fringe_at, B, Bx, Bp, Be, N, x_out,
                                        energy_out, s);
    return dtype == CHX_F32 ? dispatch_dkd<float>(kind, x_in, params, energy, mass_eV, n_charges, num_steps, fringe_at,
                                                  B, Bx, Bp, Be, N, x_out, energy_out, s)
                            : dispatch_dkd<double>(kind, x_in, params, energy, mass_eV, n_charges, num_steps,
                                                   fringe_at, B, Bx, Bp, Be, N, x_out, energy_out, s);
}

namespace {
constexpr int kDkdSChunk = 64;
struct DkdLengthPtrs {
    const void* p[kDkdSChunk];
};
// s_out = (((s_in + l_0) + l_1) + ...) in T: the additions the reference makes one element at a time (element.py `s=incoming.s +
// self.length`), so that the path length carries the same rounding
template <typename T>
__global__ void dkd_path_length_kernel(DkdLengthPtrs args, int n, const T* __restrict__ s_in, T* __restrict__ s_out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    T s = *s_in;
    for (int e = 0; e < n; ++e) s = s + *(const T*)args.p[e];
    *s_out = s;
}

// ---- a RUN of Drifts, Quadrupoles and Dipoles on one beam, particles kept in registers (chx_dkd_chain) ---------------------
// The per-element kernels above are bound by their arithmetic (a quadrupole in mixed arithmetic: ~450 instructions per particle),
// yet every one of them also moves its 56 bytes per particle through HBM and waits for them at both ends. A run of E elements of
// one arithmetic mode is two launches:
//   dkd_chain_prepare_kernel  a workgroup per element: its reference energy (the float32 round trip of bmadx.py:49 applied once
//                             per element in front of it, as the element-by-element kernels hand it on), dkd_constants in fp64,
//                             the energy it leaves (`energies[e]`);
//   dkd_chain_kernel<MODE>    a particle per lane through all elements; the element's constants are wave-uniform scalar loads;
//                             between two elements the coordinates are rounded to float32 — exactly what the store and the
//                             load of two separate launches do: the same bits as E calls of chx_dkd_track_p.
constexpr int kDkdChainMax = 192;          // elements per launch pair (kernel-argument space: 20 bytes each)
constexpr int kDkdCstStride = 56;          // doubles per element: C_MIXED_N constants — or the 49 entries of a first-order map in the
constexpr int kDkdMeta = 54;               // beam's dtype —, and at kDkdMeta the kind and the step count as two ints
constexpr int kDkdLinear = 4;              // CHX_DKD_LINEAR: a merged run of linear elements between two others
struct DkdChainArgs {
    const void* params[kDkdChainMax];      // (a linear run: its composed map R[7][7])
    const void* length[kDkdChainMax];      // the item's length if it is not the first parameter (a linear run's summed length), else null
    int32_t meta[kDkdChainMax];            // kind | fringe_at << 4 | num_steps << 6
};

template <typename T>
__device__ __forceinline__ T dkd_energy_round_trip(T E, T m) {       // ref_energy of bmadx.py:49, storage dtype
    const T p0 = sqrt(E * E - m * m);
    return sqrt(p0 * p0 + m * m);
}

template <typename T>
__global__ void dkd_chain_prepare_kernel(DkdChainArgs args, int E, const T* __restrict__ energy_in, double mc2, double nq,
                                         double* __restrict__ cst, T* __restrict__ energies, const T* s_in, T* s_out) {
    const int e = blockIdx.x;
    if (e == E) {
        // one workgroup more: the path length behind the run, s = ((s_in + l_0) + l_1) + ... in the beam's dtype like the
        // reference's element-by-element additions (every kind's first parameter is its length; s_out may be s_in)
        __shared__ T len[kDkdChainMax];
        for (int k = threadIdx.x; k < E; k += blockDim.x) len[k] = args.length[k] ? *(const T*)args.length[k] : ((const T*)args.params[k])[0];
        __syncthreads();
        if (threadIdx.x == 0) {
            T sum = *s_in;
            for (int k = 0; k < E; ++k) sum = sum + len[k];
            *s_out = sum;
        }
        return;
    }
    const int kind = args.meta[e] & 15, fringe = (args.meta[e] >> 4) & 3, steps = args.meta[e] >> 6;
    double* out = cst + (int64_t)e * kDkdCstStride;
    if (kind == kDkdLinear && threadIdx.x < 49) reinterpret_cast<T*>(out)[threadIdx.x] = ((const T*)args.params[e])[threadIdx.x];
    if (threadIdx.x != 0) return;
    // the reference energy in front of element e: every drift-kick-drift element in front of it leaves the float round trip of
    // what it received (bmadx.py:49); a linear run hands its energy on
    const T m = (T)mc2;
    T Ee = *energy_in;
    for (int k = 0; k < e; ++k) {
        if ((args.meta[k] & 15) == kDkdLinear) continue;
        const T En = dkd_energy_round_trip<T>(Ee, m);
        if (En == Ee) break;              // a fixed point: every later round trip returns it again
        Ee = En;
    }
    int32_t* w = reinterpret_cast<int32_t*>(out + kDkdMeta);
    w[0] = kind;
    w[1] = steps;
    if (kind == kDkdLinear) {
        energies[e] = Ee;
        return;
    }
    const int P = kind == CHX_DKD_DRIFT ? 1 : (kind == CHX_DKD_QUADRUPOLE ? 5 : 9);
    double par[CHX_MAX_PARAMS];
    const T* pe = (const T*)args.params[e];
    for (int k = 0; k < P; ++k) par[k] = (double)pe[k];
    double c[C_MIXED_N];
    for (int k = 0; k < C_MIXED_N; ++k) c[k] = 0.0;
    dkd_constants<double>(kind, par, (double)Ee, mc2, nq, fringe, c);
    dkd_mixed_constants(c, mc2);
    for (int k = 0; k < C_MIXED_N; ++k) out[k] = c[k];
    energies[e] = dkd_energy_round_trip<T>(Ee, m);
}

struct DkdKindsBlock {
    int32_t k[256];
};
__global__ void dkd_store_kinds_kernel(DkdKindsBlock b, int n, int32_t* __restrict__ out) {
    if ((int)threadIdx.x < n) out[threadIdx.x] = b.k[threadIdx.x];
}

// energies[e] = the reference energy behind item e of such a run, nothing else (chx_dkd_energy_chain: a caller that builds the
// maps of the linear runs in between needs the energy in front of each of them)
template <typename T>
__global__ void dkd_energy_chain_kernel(const int32_t* __restrict__ kinds_dev, int E, const T* __restrict__ energy_in, double mc2,
                                        T* __restrict__ energies) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const T m = (T)mc2;
    T Ee = *energy_in;
    for (int e = 0; e < E; ++e) {
        if (kinds_dev[e] != kDkdLinear) Ee = dkd_energy_round_trip<T>(Ee, m);
        energies[e] = Ee;
    }
}

// MODE: chx_dkd_track_p's storage_precision — 0 fp64 evaluation, 1 float32 evaluation, 2 mixed
template <int MODE, int KIND>
__device__ __forceinline__ void dkd_chain_step(const double* __restrict__ c, double mc2, int num_steps, float (&v)[6]) {
    if (MODE == 2 && KIND != CHX_DKD_DIPOLE) {          // (a Dipole in mixed mode: the fp64 evaluation, as chx_dkd_track_p does)
        dkd_mixed_particle<KIND>(c, mc2, num_steps, v);
    } else if (MODE == 0 || MODE == 2) {
        double in[6], out[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) in[j] = (double)v[j];
        dkd_map<KIND, double>(c, in, mc2, num_steps, out);
#pragma unroll
        for (int j = 0; j < 6; ++j) v[j] = (float)out[j];
    } else {
        F32 cf[C_N + 1], in[6], out[6];
#pragma unroll
        for (int k = 0; k <= C_N; ++k) cf[k] = dkd_scalar<F32>::from(c[k]);
#pragma unroll
        for (int j = 0; j < 6; ++j) in[j] = mkf(v[j]);
        dkd_map<KIND, F32>(cf, in, mc2, num_steps, out);
#pragma unroll
        for (int j = 0; j < 6; ++j) v[j] = out[j].v;
    }
}

// BEND: the run contains Dipoles (their body — asin, atan2, a dozen square roots — costs registers the other runs keep)
template <int MODE, bool BEND>
__global__ __launch_bounds__(CHX_BLOCK) void dkd_chain_kernel(const float* x_in, const double* __restrict__ cst, int E, double mc2,
                                                              float* x_out, int64_t N, int in_vec_ok, int out_vec_ok) {
    // (x_out may be x_in: a tile is read whole before it is written)
    constexpr int TP = CHX_BLOCK;
    __shared__ __attribute__((aligned(16))) float lds[TP * 7];
    const int64_t n0 = (int64_t)blockIdx.x * TP;
    const int np = (int)((N - n0 < TP) ? (N - n0) : TP);
    tile_load<float, TP>(x_in + n0 * 7, lds, np * 7, in_vec_ok != 0, true);
    __syncthreads();
    const int p = threadIdx.x;
    float v[6], v6 = p < np ? lds[p * 7 + 6] : 1.0f;       // (the drift-kick-drift kernels write 1 into the seventh column)
#pragma unroll
    for (int j = 0; j < 6; ++j) v[j] = p < np ? lds[p * 7 + j] : 0.0f;
    for (int e = 0; e < E; ++e) {
        const double* __restrict__ c = cst + (int64_t)e * kDkdCstStride;
        const int32_t* w = reinterpret_cast<const int32_t*>(c + kDkdMeta);
        const int kind = w[0], steps = w[1];
        if (kind == kDkdLinear) {
            // a merged run of linear elements: apply7's arithmetic on all seven coordinates, as chx_apply_affine7 does it
            const float* __restrict__ R = reinterpret_cast<const float*>(c);
            const float x[7] = {v[0], v[1], v[2], v[3], v[4], v[5], v6};
            float y[7];
#pragma unroll
            for (int i = 0; i < 7; ++i) {
                float acc = R[i * 7] * x[0];
#pragma unroll
                for (int j = 1; j < 7; ++j) acc = fmaf(R[i * 7 + j], x[j], acc);
                y[i] = acc;
            }
#pragma unroll
            for (int j = 0; j < 6; ++j) v[j] = y[j];
            v6 = y[6];
            continue;
        }
        if (kind == CHX_DKD_DRIFT) dkd_chain_step<MODE, CHX_DKD_DRIFT>(c, mc2, steps, v);
        else if (!BEND || kind == CHX_DKD_QUADRUPOLE) dkd_chain_step<MODE, CHX_DKD_QUADRUPOLE>(c, mc2, steps, v);
        else dkd_chain_step<MODE, CHX_DKD_DIPOLE>(c, mc2, steps, v);
        v6 = 1.0f;
    }
    if (p < np) {
#pragma unroll
        for (int j = 0; j < 6; ++j) lds[p * 7 + j] = v[j];
        lds[p * 7 + 6] = v6;
    }
    __syncthreads();
    tile_store<float, TP>(x_out + n0 * 7, lds, np * 7, out_vec_ok != 0, true);
}

// float64 beams: the map in fp64 like dkd_kernel<double, ., double>, a particle per lane
template <bool BEND>
__global__ __launch_bounds__(CHX_BLOCK) void dkd_chain_kernel_f64(const double* x_in, const double* __restrict__ cst, int E, double mc2,
                                                                  double* x_out, int64_t N, int in_vec_ok, int out_vec_ok) {
    constexpr int TP = CHX_BLOCK;              // (x_out may be x_in: a tile is read whole before it is written)
    __shared__ __attribute__((aligned(16))) double lds[TP * 7];
    const int64_t n0 = (int64_t)blockIdx.x * TP;
    const int np = (int)((N - n0 < TP) ? (N - n0) : TP);
    tile_load<double, TP>(x_in + n0 * 7, lds, np * 7, in_vec_ok != 0, true);
    __syncthreads();
    const int p = threadIdx.x;
    double v[6], v6 = p < np ? lds[p * 7 + 6] : 1.0;
#pragma unroll
    for (int j = 0; j < 6; ++j) v[j] = p < np ? lds[p * 7 + j] : 0.0;
    for (int e = 0; e < E; ++e) {
        const double* __restrict__ c = cst + (int64_t)e * kDkdCstStride;
        const int32_t* w = reinterpret_cast<const int32_t*>(c + kDkdMeta);
        const int kind = w[0], steps = w[1];
        double out[6];
        if (kind == kDkdLinear) {
            const double x[7] = {v[0], v[1], v[2], v[3], v[4], v[5], v6};
            double y[7];
#pragma unroll
            for (int i = 0; i < 7; ++i) {
                double acc = c[i * 7] * x[0];
#pragma unroll
                for (int j = 1; j < 7; ++j) acc = fma(c[i * 7 + j], x[j], acc);
                y[i] = acc;
            }
#pragma unroll
            for (int j = 0; j < 6; ++j) v[j] = y[j];
            v6 = y[6];
            continue;
        }
        if (kind == CHX_DKD_DRIFT) dkd_map<CHX_DKD_DRIFT, double>(c, v, mc2, steps, out);
        else if (!BEND || kind == CHX_DKD_QUADRUPOLE) dkd_map<CHX_DKD_QUADRUPOLE, double>(c, v, mc2, steps, out);
        else dkd_map<CHX_DKD_DIPOLE, double>(c, v, mc2, steps, out);
#pragma unroll
        for (int j = 0; j < 6; ++j) v[j] = out[j];
        v6 = 1.0;
    }
    if (p < np) {
#pragma unroll
        for (int j = 0; j < 6; ++j) lds[p * 7 + j] = v[j];
        lds[p * 7 + 6] = v6;
    }
    __syncthreads();
    tile_store<double, TP>(x_out + n0 * 7, lds, np * 7, out_vec_ok != 0, true);
}
}  // namespace

// A run of drift-kick-drift elements on ONE beam with scalar settings. float32 Drifts and Quadrupoles of one arithmetic mode: two
// launches, the particles in registers for the whole run (dkd_chain_kernel; x_tmp holds the elements' constants). Otherwise E
// launches of chx_dkd_track_p from one call: the reference energy is handed from element to element on the device (energies[e] =
// what element e leaves), the particle rows ping-pong between x_out and x_tmp so that the last element writes x_out. Either way
// the same results as E separate calls, bit for bit.
extern "C" int chx_dkd_chain(const int32_t* kinds, const void* const* params, const int32_t* num_steps, const int32_t* fringe_at,
                             const int32_t* storage_precision, int64_t E, const void* x_in, const void* energy_in, double mass_eV,
                             double n_charges, int64_t N, int dtype, void* x_out, void* x_tmp, void* energies, const void* s_in,
                             void* s_out, void* stream) {
    return chx_dkd_chain_mixed(kinds, params, nullptr, num_steps, fringe_at, storage_precision, E, x_in, energy_in, mass_eV, n_charges, N,
                               dtype, x_out, x_tmp, energies, s_in, s_out, stream);
}

// energies[e] (dtype) = the reference energy behind item e of a run: a drift-kick-drift element leaves the round trip E ->
// sqrt(p0c^2 + m^2) of what it received in dtype (bmadx.py:49), an item of kind CHX_DKD_LINEAR hands its energy on. One launch.
extern "C" int chx_dkd_energy_chain(const int32_t* kinds, int64_t E, const void* energy_in, double mass_eV, int dtype, void* energies,
                                    void* kinds_scratch, void* stream) {
    if (!kinds || E < 1 || E > 65535 || !energy_in || !energies || !kinds_scratch) return CHX_ERR_INVALID_ARG;
    if (dtype != CHX_F32 && dtype != CHX_F64) return CHX_ERR_DTYPE;
    hipStream_t s = (hipStream_t)stream;
    // the kinds travel as kernel arguments in blocks (no host buffer has to outlive the call)
    for (int64_t done = 0; done < E; done += 256) {
        DkdKindsBlock b;
        const int n = (int)std::min<int64_t>(256, E - done);
        for (int e = 0; e < 256; ++e) b.k[e] = e < n ? kinds[done + e] : 0;
        hipLaunchKernelGGL(dkd_store_kinds_kernel, dim3(1), dim3(256), 0, s, b, n, (int32_t*)kinds_scratch + done);
        CHX_CHECK_LAUNCH();
    }
    if (dtype == CHX_F32)
        hipLaunchKernelGGL(dkd_energy_chain_kernel<float>, dim3(1), dim3(1), 0, s, (const int32_t*)kinds_scratch, (int)E,
                           (const float*)energy_in, mass_eV, (float*)energies);
    else
        hipLaunchKernelGGL(dkd_energy_chain_kernel<double>, dim3(1), dim3(1), 0, s, (const int32_t*)kinds_scratch, (int)E,
                           (const double*)energy_in, mass_eV, (double*)energies);
    CHX_CHECK_LAUNCH();
    return CHX_OK;
}

// The same run with merged runs of linear elements in between: kinds[e] = CHX_DKD_LINEAR marks params[e] as a [7][7] first-order
// map (dtype) applied with the arithmetic of chx_apply_affine7, lengths[e] as the device scalar with that run's summed length
// (lengths may be NULL when no item is linear; an entry of a drift-kick-drift element may be NULL: its first parameter). A lattice
// whose drifts are tracked linearly and whose magnets with the Bmad-X maps is still one pass over the beam.
extern "C" int chx_dkd_chain_mixed(const int32_t* kinds, const void* const* params, const void* const* lengths, const int32_t* num_steps,
                                   const int32_t* fringe_at, const int32_t* storage_precision, int64_t E, const void* x_in,
                                   const void* energy_in, double mass_eV, double n_charges, int64_t N, int dtype, void* x_out,
                                   void* x_tmp, void* energies, const void* s_in, void* s_out, void* stream) {
    if (!kinds || !params || !num_steps || !fringe_at || !storage_precision || E < 1 || E > 65535) return CHX_ERR_INVALID_ARG;
    if ((s_in == nullptr) != (s_out == nullptr)) return CHX_ERR_INVALID_ARG;
    if (!x_in || !energy_in || !x_out || !energies || (E > 1 && !x_tmp) || N < 1) return CHX_ERR_INVALID_ARG;
    if (dtype != CHX_F32 && dtype != CHX_F64) return CHX_ERR_DTYPE;
    if (x_out == x_in || x_tmp == x_in || x_tmp == x_out) return CHX_ERR_INVALID_ARG;
    const size_t esz = dtype == CHX_F32 ? 4 : 8;
    int64_t first = -1;                               // the first drift-kick-drift element: its arithmetic mode is the run's
    for (int64_t e = 0; e < E; ++e) {
        if (!params[e]) return CHX_ERR_INVALID_ARG;
        if (kinds[e] == kDkdLinear) {
            if (s_out && !(lengths && lengths[e])) return CHX_ERR_INVALID_ARG;
        } else if (first < 0) {
            first = e;
        }
    }
    // Drifts, Quadrupoles, Dipoles (float32: of one arithmetic mode) and linear runs, the constants (448 bytes per item) in x_tmp:
    // the particles stay in registers (dkd_chain_kernel, dkd_chain_kernel_f64), two launches per 192 items, the same bits
    static const bool fused_off = [] { const char* v = getenv("CHX_DKD_CHAIN_FUSED"); return v && v[0] == '0'; }();
    // (a longer run takes several such pairs, the later ones in place on x_out: a workgroup holds its whole tile in registers
    // before it writes)
    const int64_t per_pass = std::min<int64_t>(kDkdChainMax, N * 7 * (int64_t)esz / (kDkdCstStride * (int64_t)sizeof(double)));
    bool fuse = !fused_off && E >= 2 && per_pass >= 2 && chx_aligned16(x_tmp) && first >= 0;
    bool bend = false;
    for (int64_t e = 0; fuse && e < E; ++e) {      // (float64 beams ignore storage_precision, like chx_dkd_track_p)
        if (kinds[e] == kDkdLinear) continue;
        fuse = (kinds[e] == CHX_DKD_DRIFT || kinds[e] == CHX_DKD_QUADRUPOLE || kinds[e] == CHX_DKD_DIPOLE) &&
               (dtype == CHX_F64 || (storage_precision[e] == storage_precision[first] && storage_precision[e] >= 0 && storage_precision[e] <= 2)) &&
               num_steps[e] >= 1 && num_steps[e] < (1 << 25) && fringe_at[e] >= 0 && fringe_at[e] <= 3;
        bend = bend || kinds[e] == CHX_DKD_DIPOLE;
    }
    if (fuse) {
        hipStream_t s = (hipStream_t)stream;
        const int64_t tiles = (N + CHX_BLOCK - 1) / CHX_BLOCK;
        if (tiles > 0x7fffffffLL) return CHX_ERR_INVALID_ARG;
        const int mode = storage_precision[first];
        for (int64_t done = 0; done < E; done += per_pass) {
            const int n = (int)std::min<int64_t>(per_pass, E - done);
            DkdChainArgs a;
            for (int e = 0; e < kDkdChainMax; ++e) {
                const bool lin = e < n && kinds[done + e] == kDkdLinear;
                a.params[e] = e < n ? params[done + e] : nullptr;
                a.length[e] = e < n && lengths ? lengths[done + e] : nullptr;
                a.meta[e] = e >= n ? 0 : (lin ? kDkdLinear : (kinds[done + e] | (fringe_at[done + e] << 4) | (num_steps[done + e] << 6)));
            }
            const void* e_from = done == 0 ? energy_in : (const void*)((const char*)energies + (size_t)(done - 1) * esz);
            void* e_to = (char*)energies + (size_t)done * esz;
            const void* s_from = done == 0 ? s_in : s_out;
            const unsigned blocks = (unsigned)(n + (s_out ? 1 : 0));
            const void* src = done == 0 ? x_in : x_out;
            const int in_ok = (int)chx_aligned16(src), out_ok = (int)chx_aligned16(x_out);
            if (dtype == CHX_F64) {
                hipLaunchKernelGGL(dkd_chain_prepare_kernel<double>, dim3(blocks), dim3(64), 0, s, a, n, (const double*)e_from, mass_eV,
                                   n_charges, (double*)x_tmp, (double*)e_to, (const double*)s_from, (double*)s_out);
                CHX_CHECK_LAUNCH();
                if (bend)
                    hipLaunchKernelGGL(dkd_chain_kernel_f64<true>, dim3((unsigned)tiles), dim3(CHX_BLOCK), 0, s, (const double*)src,
                                       (const double*)x_tmp, n, mass_eV, (double*)x_out, N, in_ok, out_ok);
                else
                    hipLaunchKernelGGL(dkd_chain_kernel_f64<false>, dim3((unsigned)tiles), dim3(CHX_BLOCK), 0, s, (const double*)src,
                                       (const double*)x_tmp, n, mass_eV, (double*)x_out, N, in_ok, out_ok);
                CHX_CHECK_LAUNCH();
                continue;
            }
            hipLaunchKernelGGL(dkd_chain_prepare_kernel<float>, dim3(blocks), dim3(64), 0, s, a, n, (const float*)e_from, mass_eV,
                               n_charges, (double*)x_tmp, (float*)e_to, (const float*)s_from, (float*)s_out);
            CHX_CHECK_LAUNCH();
#define CHX_DKD_CHAIN_LAUNCH(M, B)                                                                                                \
    hipLaunchKernelGGL((dkd_chain_kernel<M, B>), dim3((unsigned)tiles), dim3(CHX_BLOCK), 0, s, (const float*)src, (const double*)x_tmp, \
                       n, mass_eV, (float*)x_out, N, in_ok, out_ok)
            if (mode == 2 && bend) CHX_DKD_CHAIN_LAUNCH(2, true);
            else if (mode == 2) CHX_DKD_CHAIN_LAUNCH(2, false);
            else if (mode == 1 && bend) CHX_DKD_CHAIN_LAUNCH(1, true);
            else if (mode == 1) CHX_DKD_CHAIN_LAUNCH(1, false);
            else if (bend) CHX_DKD_CHAIN_LAUNCH(0, true);
            else CHX_DKD_CHAIN_LAUNCH(0, false);
#undef CHX_DKD_CHAIN_LAUNCH
            CHX_CHECK_LAUNCH();
        }
    } else {
        const void* src = x_in;
        const void* e_src = energy_in;
        for (int64_t e = 0; e < E; ++e) {
            void* dst = ((E - 1 - e) & 1) ? x_tmp : x_out;          // the last element lands in x_out
            void* e_dst = (char*)energies + (size_t)e * esz;
            int st;
            if (kinds[e] == kDkdLinear) {
                st = chx_apply_affine7(src, params[e], dst, 1, 1, 1, N, dtype, stream);
                if (st == CHX_OK && hipMemcpyAsync(e_dst, e_src, esz, hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess)
                    st = CHX_ERR_LAUNCH;
            } else {
                st = chx_dkd_track_p(kinds[e], src, params[e], e_src, mass_eV, n_charges, num_steps[e], fringe_at[e], 1, 1, 1, 1, N, dtype,
                                     storage_precision[e], dst, e_dst, stream);
            }
            if (st != CHX_OK) return st;
            src = dst;
            e_src = e_dst;
        }
    }
    // the path length behind the run: every kind's first parameter is its length (the fused form has it from its first launch)
    for (int64_t done = 0; s_out && !fuse && done < E; done += kDkdSChunk) {
        DkdLengthPtrs a;
        const int n = (int)((E - done < kDkdSChunk) ? (E - done) : kDkdSChunk);
        for (int e = 0; e < kDkdSChunk; ++e)
            a.p[e] = e >= n ? nullptr : ((lengths && lengths[done + e]) ? lengths[done + e] : params[done + e]);
        const void* from = done == 0 ? s_in : s_out;
        if (dtype == CHX_F32)
            hipLaunchKernelGGL(dkd_path_length_kernel<float>, dim3(1), dim3(1), 0, (hipStream_t)stream, a, n, (const float*)from, (float*)s_out);
        else
            hipLaunchKernelGGL(dkd_path_length_kernel<double>, dim3(1), dim3(1), 0, (hipStream_t)stream, a, n, (const double*)from,
                               (double*)s_out);
        CHX_CHECK_LAUNCH();
    }
    return CHX_OK;
}

namespace {
// ---- a RUN of second-order elements on one float32 beam, particles kept in registers (chx_second_order_chain) ---------------
// E launches of second_order_pk_kernel move 56 B per particle and element through HBM for ~50 useful multiply-adds; a run of
// elements needs none of that traffic: Segment.track only hands out the beam behind the run. Two launches for the whole run:
//   so_chain_coeff_kernel  (a workgroup per element) folds T_ijk + T_ikj of every element into its 7 x 28 coefficients, finds
//                          which of them are non-zero and files the element under one of three evaluation schemes;
//   so_chain_kernel        every lane carries two particles through all elements; the coefficients are wave-uniform and come
//                          through scalar loads: no LDS, no barrier inside the loop.
// The schemes differ only in WHICH exact zeros are skipped (a skipped term is 0 * q, an exact no-op on a finite beam), never in
// the order of the remaining multiply-adds: per element the result is second_order_pk_kernel's, value for value.
//   scheme 0: the 27 coefficients an upright Quadrupole can have (a Drift's 15 are among them): 13 products + 27 multiply-adds;
//   scheme 1: the 55 of upright Dipoles (edges, gradient), RBends and Sextupoles: 22 products + 55 multiply-adds;
//   scheme 2: anything else (tilted, misaligned, custom maps): groups of four coefficients, skipped when all four are zero.
constexpr int kSoChainMax = 224;           // elements per launch pair (kernel-argument space: a map and a length pointer each)
constexpr int kSoCoefStride = 256;         // floats per element in the scratch: 196 coefficients, 2 group-mask words, the scheme,
constexpr int kSoPacked = 200;             // one unused; from kSoPacked on the coefficients of the element's pattern, back to back
struct SoChainPtrs {
    const void* T[kSoChainMax];
    const void* length[kSoChainMax];
    unsigned char linear[kSoChainMax];     // 1: T[e] is a 7 x 7 first-order map (a merged run of linear elements between two others)
};
// bit c of row i = coefficient (i, c) of the folded map, c = the pair (j <= k) in the order 00 01 .. 06 11 12 .. 66
struct SoPatternQuad {
    static constexpr unsigned int rows[7] = {0x1860u, 0x1860u, 0x330000u, 0x330000u, 0x7046083u, 0x4000000u, 0x8000000u};
};
struct SoPatternBend {
    static constexpr unsigned int rows[7] = {0x60478e3u, 0x60478e3u, 0x33030cu, 0x33030cu, 0x70478e3u, 0x4000000u, 0x8000000u};
};
constexpr int kSoJ[28] = {0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 5, 5, 6};
constexpr int kSoK[28] = {0, 1, 2, 3, 4, 5, 6, 1, 2, 3, 4, 5, 6, 2, 3, 4, 5, 6, 3, 4, 5, 6, 4, 5, 6, 5, 6, 6};

template <typename T>
__global__ __launch_bounds__(CHX_BLOCK) void so_chain_coeff_kernel(SoChainPtrs maps, int E, T* __restrict__ coef, const T* s_in, T* s_out) {
    const int e = blockIdx.x;
    if (e == E) {
        // one workgroup more: the path length behind the run, s = ((s_in + l_0) + l_1) + ... in the beam's dtype like the reference's
        // element-by-element additions (s_out may be s_in)
        static_assert(kSoChainMax <= CHX_BLOCK, "one lane per length");
        __shared__ T len[kSoChainMax];
        if (threadIdx.x < E) len[threadIdx.x] = *(const T*)maps.length[threadIdx.x];
        __syncthreads();
        if (threadIdx.x == 0) {
            T sum = *s_in;
            for (int k = 0; k < E; ++k) sum = sum + len[k];
            *s_out = sum;
        }
        return;
    }
    const T* Tt = (const T*)maps.T[e];
    T* out = coef + (int64_t)e * kSoCoefStride;
    if (maps.linear[e]) {                      // scheme 3: the 49 entries of a first-order map, applied like chx_apply_affine7
        if (threadIdx.x < 49) out[kSoPacked + threadIdx.x] = Tt[threadIdx.x];
        if (threadIdx.x == 0) {
            unsigned int* w = reinterpret_cast<unsigned int*>(out + 196);
            w[0] = 0u; w[1] = 0u; w[2] = 3u; w[3] = 0u;
        }
        return;
    }
    T u = (T)0;
    int i = 0, c = 0;
    if (threadIdx.x < 7 * 28) {
        i = threadIdx.x / 28;
        c = threadIdx.x - i * 28;
        const int j = kSoJ[c], k = kSoK[c];
        const T* Tb = Tt + i * 49;
        u = (j == k) ? Tb[j * 7 + k] : Tb[j * 7 + k] + Tb[k * 7 + j];
        out[threadIdx.x] = u;
    }
    __shared__ unsigned int any4[64];          // group g = coefficients 4g .. 4g + 3
    __shared__ unsigned int outside[2];        // a non-zero coefficient outside pattern 0 / 1
    if (threadIdx.x < 64) any4[threadIdx.x] = 0u;
    if (threadIdx.x < 2) outside[threadIdx.x] = 0u;
    __syncthreads();
    if (threadIdx.x < 7 * 28 && u != (T)0) {
        atomicOr(&any4[threadIdx.x >> 2], 1u);
        if (!((SoPatternQuad::rows[i] >> c) & 1u)) atomicOr(&outside[0], 1u);
        if (!((SoPatternBend::rows[i] >> c) & 1u)) atomicOr(&outside[1], 1u);
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        const unsigned long long m = __ballot(threadIdx.x < 49 && any4[threadIdx.x] != 0u);
        if (threadIdx.x == 0) {
            unsigned int* w = reinterpret_cast<unsigned int*>(out + 196);        // three 32-bit words behind the coefficients
            w[0] = (unsigned int)(m & 0xffffffffull);
            w[1] = (unsigned int)(m >> 32);
            w[2] = outside[0] == 0u ? 0u : (outside[1] == 0u ? 1u : 2u);
            w[3] = 0u;
        }
    }
    // the pattern's coefficients back to back (row by row, ascending c): three or four wide scalar loads per element and wave
    // instead of seventeen narrow ones — the scalar memory path of a CU serves its 32 waves one request at a time
    if (threadIdx.x < 7 * 28 && outside[1] == 0u) {
        const unsigned int* rows = outside[0] == 0u ? SoPatternQuad::rows : SoPatternBend::rows;
        if ((rows[i] >> c) & 1u) {
            int n = __popc(rows[i] & ((1u << c) - 1u));
            for (int r = 0; r < i; ++r) n += __popc(rows[r]);
            out[kSoPacked + n] = u;
        }
    }
}

// one element, coefficients of a fixed pattern: every product once, every row its multiply-adds in ascending c
template <class P>
struct SoPatternSize {
    static constexpr int count() {
        int n = 0;
        for (int i = 0; i < 7; ++i)
            for (int c = 0; c < 28; ++c) n += (P::rows[i] >> c) & 1u;
        return n;
    }
};
template <class P>
__device__ __forceinline__ void so_load_pattern(const float* __restrict__ U, float (&u)[SoPatternSize<P>::count()]) {
#pragma unroll
    for (int n = 0; n < SoPatternSize<P>::count(); ++n) u[n] = U[kSoPacked + n];
}
template <class P>
__device__ __forceinline__ void so_eval_pattern(const float (&u)[SoPatternSize<P>::count()], chx_v2f (&x)[7], chx_v2f probe) {
    constexpr unsigned int cols = P::rows[0] | P::rows[1] | P::rows[2] | P::rows[3] | P::rows[4] | P::rows[5] | P::rows[6];
    chx_v2f q[28];
#pragma unroll
    for (int c = 0; c < 28; ++c)
        if ((cols >> c) & 1u) q[c] = x[kSoJ[c]] * x[kSoK[c]];
    chx_v2f y[7];
    int n = 0;
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        chx_v2f acc = probe;
#pragma unroll
        for (int c = 0; c < 28; ++c)
            if ((P::rows[i] >> c) & 1u) {
                const float uu = u[n++];
                acc = __builtin_elementwise_fma(chx_v2f{uu, uu}, q[c], acc);
            }
        y[i] = acc;
    }
#pragma unroll
    for (int j = 0; j < 7; ++j) x[j] = y[j];
}
template <class P>
__device__ __forceinline__ void so_step_pattern(const float* __restrict__ U, chx_v2f (&x)[7], chx_v2f probe) {
    float u[SoPatternSize<P>::count()];
    so_load_pattern<P>(U, u);
    so_eval_pattern<P>(u, x, probe);
}

// a first-order map between two second-order elements: apply7's order (R_i0 x_0, then six multiply-adds), both particles of
// the lane at once — the values chx_apply_affine7 writes
__device__ __forceinline__ void so_step_linear(const float* __restrict__ U, chx_v2f (&x)[7]) {
    float r[49];
#pragma unroll
    for (int k = 0; k < 49; ++k) r[k] = U[kSoPacked + k];
    chx_v2f y[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        chx_v2f acc = x[0] * r[i * 7];
#pragma unroll
        for (int j = 1; j < 7; ++j) acc = __builtin_elementwise_fma(chx_v2f{r[i * 7 + j], r[i * 7 + j]}, x[j], acc);
        y[i] = acc;
    }
#pragma unroll
    for (int j = 0; j < 7; ++j) x[j] = y[j];
}

__device__ __forceinline__ void so_step_groups(const float* __restrict__ U, chx_v2f (&x)[7], chx_v2f probe) {
    const unsigned long long g = (unsigned long long)reinterpret_cast<const unsigned int*>(U + 196)[0] |
                                 ((unsigned long long)reinterpret_cast<const unsigned int*>(U + 196)[1] << 32);
    chx_v2f y[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        chx_v2f acc = probe;
        if ((g >> (i * 7)) & 0x7full) {                            // wave-uniform: a row without coefficients is skipped whole
            float u[28];
#pragma unroll
            for (int c = 0; c < 28; ++c) u[c] = U[i * 28 + c];
#pragma unroll
            for (int m = 0; m < 7; ++m) {
                if (!((g >> (i * 7 + m)) & 1ull)) continue;       // wave-uniform
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const int c = 4 * m + q4;
                    const chx_v2f q = x[kSoJ[c]] * x[kSoK[c]];
                    acc = __builtin_elementwise_fma(chx_v2f{u[c], u[c]}, q, acc);
                }
            }
        }
        y[i] = acc;
    }
#pragma unroll
    for (int j = 0; j < 7; ++j) x[j] = y[j];
}

__global__ __launch_bounds__(CHX_BLOCK) __attribute__((amdgpu_waves_per_eu(8, 8)))
void so_chain_kernel(const float* x_in, const float* __restrict__ coef, int E, float* x_out, int64_t N, int in_vec_ok,
                     int out_vec_ok) {                  // (x_out may be x_in: a tile is read whole before it is written)
    constexpr int TP = 2 * CHX_BLOCK;
    __shared__ __attribute__((aligned(16))) float lds[TP * 7];
    const int64_t n0 = (int64_t)blockIdx.x * TP;
    const int np = (int)((N - n0 < TP) ? (N - n0) : TP);
    tile_load<float, TP>(x_in + n0 * 7, lds, np * 7, in_vec_ok != 0, true);
    __syncthreads();
    const int p0 = threadIdx.x, p1 = threadIdx.x + CHX_BLOCK;
    const bool on0 = p0 < np, on1 = p1 < np;
    chx_v2f x[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) x[j] = chx_v2f{on0 ? lds[p0 * 7 + j] : 0.0f, on1 ? lds[p1 * 7 + j] : 0.0f};
    for (int e = 0; e < E; ++e) {
        const float* __restrict__ U = coef + (int64_t)e * kSoCoefStride;
        const unsigned int scheme = reinterpret_cast<const unsigned int*>(U + 196)[2];
        // 0 * x is NaN exactly for a non-finite x and +0 otherwise: every row's sum STARTS from this value, so a particle that
        // left the finite range comes out as NaN in all seven coordinates (second_order_pk_kernel's rule) and nothing changes
        // for the others (+0 + a = a)
        chx_v2f probe = chx_v2f{0.0f, 0.0f};
#pragma unroll
        for (int j = 0; j < 7; ++j) probe = __builtin_elementwise_fma(chx_v2f{0.0f, 0.0f}, x[j], probe);
        if (scheme == 0u) so_step_pattern<SoPatternQuad>(U, x, probe);
        else if (scheme == 1u) so_step_pattern<SoPatternBend>(U, x, probe);
        else if (scheme == 3u) so_step_linear(U, x);
        else so_step_groups(U, x, probe);
    }
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        if (on0) lds[p0 * 7 + j] = x[j].x;
        if (on1) lds[p1 * 7 + j] = x[j].y;
    }
    __syncthreads();
    tile_store<float, TP>(x_out + n0 * 7, lds, np * 7, out_vec_ok != 0, true);
}


// ---- the same for float64 beams: one particle per lane, fp64 multiply-adds. second_order_kernel<double> sums a row densely
// (U_i0 q_0, then 27 multiply-adds); leaving out the terms whose coefficient is an exact zero changes no value of a finite beam
// (fma(0, q, acc) = acc), a non-finite coordinate turns the whole particle into NaN there (0 * inf) and here (the probe).
template <class P>
__device__ __forceinline__ void so_step_pattern_f64(const double* __restrict__ U, double (&x)[7], double probe) {
    constexpr unsigned int cols = P::rows[0] | P::rows[1] | P::rows[2] | P::rows[3] | P::rows[4] | P::rows[5] | P::rows[6];
    double q[28];
#pragma unroll
    for (int c = 0; c < 28; ++c)
        if ((cols >> c) & 1u) q[c] = x[kSoJ[c]] * x[kSoK[c]];
    double y[7];
    int n = 0;
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        double acc = probe;
#pragma unroll
        for (int c = 0; c < 28; ++c)
            if ((P::rows[i] >> c) & 1u) acc = fma(U[kSoPacked + n++], q[c], acc);
        y[i] = acc;
    }
#pragma unroll
    for (int j = 0; j < 7; ++j) x[j] = y[j];
}

__device__ __forceinline__ void so_step_groups_f64(const double* __restrict__ U, double (&x)[7], double probe) {
    const unsigned long long g = (unsigned long long)reinterpret_cast<const unsigned int*>(U + 196)[0] |
                                 ((unsigned long long)reinterpret_cast<const unsigned int*>(U + 196)[1] << 32);
    double y[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        double acc = probe;
#pragma unroll
        for (int m = 0; m < 7; ++m) {
            if (!((g >> (i * 7 + m)) & 1ull)) continue;           // wave-uniform
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const int c = 4 * m + q4;
                acc = fma(U[i * 28 + c], x[kSoJ[c]] * x[kSoK[c]], acc);
            }
        }
        y[i] = acc;
    }
#pragma unroll
    for (int j = 0; j < 7; ++j) x[j] = y[j];
}

__device__ __forceinline__ void so_step_linear_f64(const double* __restrict__ U, double (&x)[7]) {
    double y[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        double acc = U[kSoPacked + i * 7] * x[0];
#pragma unroll
        for (int j = 1; j < 7; ++j) acc = fma(U[kSoPacked + i * 7 + j], x[j], acc);
        y[i] = acc;
    }
#pragma unroll
    for (int j = 0; j < 7; ++j) x[j] = y[j];
}

__global__ __launch_bounds__(CHX_BLOCK) void so_chain_kernel_f64(const double* x_in, const double* __restrict__ coef, int E, double* x_out,
                                                                 int64_t N, int in_vec_ok, int out_vec_ok) {
    constexpr int TP = CHX_BLOCK;              // (x_out may be x_in: a tile is read whole before it is written)
    __shared__ __attribute__((aligned(16))) double lds[TP * 7];
    const int64_t n0 = (int64_t)blockIdx.x * TP;
    const int np = (int)((N - n0 < TP) ? (N - n0) : TP);
    tile_load<double, TP>(x_in + n0 * 7, lds, np * 7, in_vec_ok != 0, true);
    __syncthreads();
    const int p = threadIdx.x;
    double x[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) x[j] = p < np ? lds[p * 7 + j] : 0.0;
    for (int e = 0; e < E; ++e) {
        const double* __restrict__ U = coef + (int64_t)e * kSoCoefStride;
        const unsigned int scheme = reinterpret_cast<const unsigned int*>(U + 196)[2];
        double probe = 0.0;
#pragma unroll
        for (int j = 0; j < 7; ++j) probe = fma(0.0, x[j], probe);
        if (scheme == 0u) so_step_pattern_f64<SoPatternQuad>(U, x, probe);
        else if (scheme == 1u) so_step_pattern_f64<SoPatternBend>(U, x, probe);
        else if (scheme == 3u) so_step_linear_f64(U, x);
        else so_step_groups_f64(U, x, probe);
    }
    if (p < np) {
#pragma unroll
        for (int j = 0; j < 7; ++j) lds[p * 7 + j] = x[j];
    }
    __syncthreads();
    tile_store<double, TP>(x_out + n0 * 7, lds, np * 7, out_vec_ok != 0, true);
}

}  // namespace

// A run of elements tracked with their second-order maps (element.py:195-228) on ONE beam. float32: the particles stay in
// registers for the whole run (so_chain_kernel, two launches; x_tmp holds the folded coefficients). Otherwise E launches of
// chx_apply_second_order, rows ping-ponging between x_out and x_tmp so that the last element writes x_out; the path length like
// chx_dkd_chain (lengths[E]: device pointers to the elements' length scalars). Same results as E separate calls, bit for bit.
extern "C" int chx_second_order_chain(const void* const* T_maps, const void* const* lengths, int64_t E, const void* x_in, int64_t N,
                                      int dtype, void* x_out, void* x_tmp, const void* s_in, void* s_out, void* stream) {
    return chx_second_order_chain_mixed(T_maps, nullptr, lengths, E, x_in, N, dtype, x_out, x_tmp, s_in, s_out, stream);
}

// The same run with first-order maps in between: linear[e] != 0 marks T_maps[e] as a [7][7] map (a merged run of linear elements
// that stands between second-order elements, applied like chx_apply_affine7) — a lattice whose drifts are tracked linearly and
// whose magnets to second order is still ONE pass over the beam.
extern "C" int chx_second_order_chain_mixed(const void* const* T_maps, const int32_t* linear, const void* const* lengths, int64_t E,
                                            const void* x_in, int64_t N, int dtype, void* x_out, void* x_tmp, const void* s_in,
                                            void* s_out, void* stream) {
    if (!T_maps || E < 1 || E > 65535 || !x_in || !x_out || (E > 1 && !x_tmp) || N < 1) return CHX_ERR_INVALID_ARG;
    if (dtype != CHX_F32 && dtype != CHX_F64) return CHX_ERR_DTYPE;
    if ((s_in == nullptr) != (s_out == nullptr) || (s_out && !lengths)) return CHX_ERR_INVALID_ARG;
    if (x_out == x_in || x_tmp == x_in || x_tmp == x_out) return CHX_ERR_INVALID_ARG;
    for (int64_t e = 0; e < E; ++e)
        if (!T_maps[e]) return CHX_ERR_INVALID_ARG;
    // a scratch of 256 coefficients per element inside x_tmp: the particles stay in registers for the whole run (so_chain_kernel,
    // so_chain_kernel_f64) — two launches per 224 elements instead of E passes over HBM, the same values
    static const bool fused_off = [] { const char* v = getenv("CHX_SO_CHAIN_FUSED"); return v && v[0] == '0'; }();
    // (a longer run takes several such pairs, the later ones in place on x_out: a workgroup holds its whole tile in registers
    // before it writes)
    const int64_t per_pass = std::min<int64_t>(kSoChainMax, N * 7 / kSoCoefStride);
    const bool fuse = !fused_off && E >= 2 && per_pass >= 2;
    if (fuse) {
        hipStream_t s = (hipStream_t)stream;
        const int per_tile = dtype == CHX_F32 ? 2 * CHX_BLOCK : CHX_BLOCK;
        const int64_t tiles = (N + per_tile - 1) / per_tile;
        if (tiles > 0x7fffffffLL) return CHX_ERR_INVALID_ARG;
        for (int64_t done = 0; done < E; done += per_pass) {
            const int n = (int)std::min<int64_t>(per_pass, E - done);
            SoChainPtrs maps;
            for (int e = 0; e < kSoChainMax; ++e) {
                maps.T[e] = e < n ? T_maps[done + e] : nullptr;
                maps.length[e] = e < n && lengths ? lengths[done + e] : nullptr;
                maps.linear[e] = (e < n && linear && linear[done + e]) ? 1 : 0;
                if (s_out && e < n && !maps.length[e]) return CHX_ERR_INVALID_ARG;
            }
            const void* src = done == 0 ? x_in : x_out;
            const void* s_from = done == 0 ? s_in : s_out;
            const unsigned blocks = (unsigned)(n + (s_out ? 1 : 0));
            if (dtype == CHX_F32) {
                hipLaunchKernelGGL(so_chain_coeff_kernel<float>, dim3(blocks), dim3(CHX_BLOCK), 0, s, maps, n, (float*)x_tmp,
                                   (const float*)s_from, (float*)s_out);
                CHX_CHECK_LAUNCH();
                hipLaunchKernelGGL(so_chain_kernel, dim3((unsigned)tiles), dim3(CHX_BLOCK), 0, s, (const float*)src, (const float*)x_tmp, n,
                                   (float*)x_out, N, (int)chx_aligned16(src), (int)chx_aligned16(x_out));
            } else {
                hipLaunchKernelGGL(so_chain_coeff_kernel<double>, dim3(blocks), dim3(CHX_BLOCK), 0, s, maps, n, (double*)x_tmp,
                                   (const double*)s_from, (double*)s_out);
                CHX_CHECK_LAUNCH();
                hipLaunchKernelGGL(so_chain_kernel_f64, dim3((unsigned)tiles), dim3(CHX_BLOCK), 0, s, (const double*)src,
                                   (const double*)x_tmp, n, (double*)x_out, N, (int)chx_aligned16(src), (int)chx_aligned16(x_out));
            }
            CHX_CHECK_LAUNCH();
        }
    } else {
        const void* src = x_in;
        for (int64_t e = 0; e < E; ++e) {
            void* dst = ((E - 1 - e) & 1) ? x_tmp : x_out;
            const int st = (linear && linear[e]) ? chx_apply_affine7(src, T_maps[e], dst, 1, 1, 1, N, dtype, stream)
                                                 : chx_apply_second_order(src, T_maps[e], dst, 1, 1, 1, N, dtype, stream);
            if (st != CHX_OK) return st;
            src = dst;
        }
    }
    for (int64_t done = 0; s_out && !fuse && done < E; done += kDkdSChunk) {      // (the fused form has s from its first launch)
        DkdLengthPtrs a;
        const int n = (int)((E - done < kDkdSChunk) ? (E - done) : kDkdSChunk);
        for (int e = 0; e < kDkdSChunk; ++e) a.p[e] = e < n ? lengths[done + e] : nullptr;
        for (int e = 0; e < n; ++e)
            if (!a.p[e]) return CHX_ERR_INVALID_ARG;
        const void* from = done == 0 ? s_in : s_out;
        if (dtype == CHX_F32)
            hipLaunchKernelGGL(dkd_path_length_kernel<float>, dim3(1), dim3(1), 0, (hipStream_t)stream, a, n, (const float*)from, (float*)s_out);
        else
            hipLaunchKernelGGL(dkd_path_length_kernel<double>, dim3(1), dim3(1), 0, (hipStream_t)stream, a, n, (const double*)from,
                               (double*)s_out);
        CHX_CHECK_LAUNCH();
    }
    return CHX_OK;
}

namespace {
template <typename T, int KIND>
int launch_dkd_bwd(const void* x_in, const void* params, const void* energy, const void* dY, double mc2, double nq,
                   int num_steps, int fringe, int P, int64_t B, int64_t Bx, int64_t Bp, int64_t Be, int64_t N, void* dx,
                   double* partials, hipStream_t s) {
    const int64_t tiles = ((N + CHX_BLOCK - 1) / CHX_BLOCK) * B;
    if (tiles > 0x7fffffffLL) return CHX_ERR_INVALID_ARG;
    hipLaunchKernelGGL((dkd_bwd_kernel<T, KIND>), dim3((unsigned)tiles), dim3(CHX_BLOCK), 0, s, (const T*)x_in,
                       (const T*)params, (const T*)energy, (const T*)dY, mc2, nq, num_steps, fringe, P, B, Bx, Bp, Be, N,
                       (T*)dx, partials);
    CHX_CHECK_LAUNCH();
    return CHX_OK;
}

template <typename T>
int dispatch_dkd_bwd(int kind, const void* x_in, const void* params, const void* energy, const void* dY, double mc2,
                     double nq, int num_steps, int fringe, int64_t B, int64_t Bx, int64_t Bp, int64_t Be, int64_t N,
                     void* dx, double* partials, hipStream_t s) {
    const int P = chx_dkd_num_params(kind);
    switch (kind) {
        case CHX_DKD_DRIFT:
            return launch_dkd_bwd<T, CHX_DKD_DRIFT>(x_in, params, energy, dY, mc2, nq, num_steps, fringe, P, B, Bx, Bp,
                                                    Be, N, dx, partials, s);
        case CHX_DKD_QUADRUPOLE:
            return launch_dkd_bwd<T, CHX_DKD_QUADRUPOLE>(x_in, params, energy, dY, mc2, nq, num_steps, fringe, P, B, Bx,
                                                         Bp, Be, N, dx, partials, s);
        case CHX_DKD_DIPOLE:
            return launch_dkd_bwd<T, CHX_DKD_DIPOLE>(x_in, params, energy, dY, mc2, nq, num_steps, fringe, P, B, Bx, Bp,
                                                     Be, N, dx, partials, s);
        case CHX_DKD_TDC:
            return launch_dkd_bwd<T, CHX_DKD_TDC>(x_in, params, energy, dY, mc2, nq, num_steps, fringe, P, B, Bx, Bp, Be,
                                                  N, dx, partials, s);
    }
    return CHX_ERR_INVALID_ARG;
}
}  // namespace

extern "C" int64_t chx_dkd_bwd_partials_count(int kind, int64_t B, int64_t N) {
    const int P = chx_dkd_num_params(kind);
    if (P < 0 || B < 0 || N < 0) return 0;
    return B * ((N + CHX_BLOCK - 1) / CHX_BLOCK) * (P + 1);
}

extern "C" int chx_dkd_track_bwd(int kind, const void* x_in, const void* params, const void* energy, const void* dY,
                                 double mass_eV, double n_charges, int32_t num_steps, int32_t fringe_at, int64_t B,
                                 int64_t Bx, int64_t Bp, int64_t Be, int64_t N, int dtype, void* dx, double* partials,
                                 void* stream) {
    if (chx_dkd_num_params(kind) < 0) return CHX_ERR_INVALID_ARG;
    if (dtype != CHX_F32 && dtype != CHX_F64) return CHX_ERR_DTYPE;
    if (B < 0 || N < 0) return CHX_ERR_INVALID_ARG;
    if (B == 0 || N == 0) return CHX_OK;
    if (!x_in || !params || !energy || !dY || (!dx && !partials)) return CHX_ERR_INVALID_ARG;
    if (!chx_bcast_ok(Bx, B) || !chx_bcast_ok(Bp, B) || !chx_bcast_ok(Be, B)) return CHX_ERR_INVALID_ARG;
    if (kind == CHX_DKD_QUADRUPOLE && num_steps < 1) return CHX_ERR_INVALID_ARG;
    if (fringe_at < 0 || fringe_at > 3) return CHX_ERR_INVALID_ARG;
    hipStream_t s = (hipStream_t)stream;
    return dtype == CHX_F32 ? dispatch_dkd_bwd<float>(kind, x_in, params, energy, dY, mass_eV, n_charges, num_steps,
                                                      fringe_at, B, Bx, Bp, Be, N, dx, partials, s)
                            : dispatch_dkd_bwd<double>(kind, x_in, params, energy, dY, mass_eV, n_charges, num_steps,
                                                       fringe_at, B, Bx, Bp, Be, N, dx, partials, s);
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
    tile_load<T, TP>(x_in + (in_row * N + n0) * 7, lds, np * 7, in_vec, !(Bx == 1 && B > 1));
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
    tile_store<T, TP>(x_out + (b * N + n0) * 7, lds, np * 7, out_vec, true);
}


// float32: TWO particles per lane in packed arithmetic (v_pk_mul_f32 / v_pk_fma_f32 on {particle p, particle p + 256}): every
// coefficient fetched from LDS serves two particles and every instruction two multiply-adds; per-lane IEEE fma on both halves,
// the same products in the same order as the one-particle kernel above.
// Round 4 (profiles/r04_second_order.md; kernel trace at 1e6 particles, the linear apply = 9.1 us): the round-3 form of this
// kernel took 17.9 us — 6 of them arithmetic, 1.8 the coefficient fill in FRONT of the tile loads (its two dependent global
// loads stalled every workgroup before it had requested a single row), ~2.6 the lower occupancy (74 VGPRs: the 28 products of
// both particles were kept in registers) and a third barrier. Now: the rows are requested first and the coefficients folded
// while they fly; the products x_j x_k are formed where they are used (same rounding: a product, then an fma), which brings the
// kernel to the apply kernel's 8 waves per SIMD; groups of four coefficients that are all zero are skipped (a drift's tensor has
// 15 non-zero entries of 343, a quadrupole's 27: track_methods.py:80-296) — wave-uniform jumps over a ds_read_b128, four packed
// products and four packed FMAs; a lane overwrites the rows it has read itself, so two barriers suffice. (Wave-private staging
// without any workgroup barrier was measured slower, 19.4 us: as for the linear apply at this size.)
__global__ __launch_bounds__(CHX_BLOCK) __attribute__((amdgpu_waves_per_eu(8, 8)))
void second_order_pk_kernel(const float* __restrict__ x_in, const float* __restrict__ Tt, float* __restrict__ x_out, int64_t B,
                            int64_t Bx, int64_t BT, int64_t N, int in_vec_ok, int out_vec_ok) {
    constexpr int TP = 2 * CHX_BLOCK;
    __shared__ __attribute__((aligned(16))) float lds[TP * 7];
    __shared__ __attribute__((aligned(16))) float U[7 * 28];   // rows of 28 = seven 16-byte groups: read as ds_read_b128
    const int64_t tiles_per_row = (N + TP - 1) / TP;
    const int64_t b = blockIdx.x / tiles_per_row;
    const int64_t t = blockIdx.x - b * tiles_per_row;
    const int64_t n0 = t * TP;
    const int np = (int)((N - n0 < TP) ? (N - n0) : TP);
    const int64_t in_row = (Bx == 1) ? 0 : b;
    const bool in_vec = in_vec_ok && (((in_row * N * 7 * (int64_t)sizeof(float)) & 15) == 0);
    const bool out_vec = out_vec_ok && (((b * N * 7 * (int64_t)sizeof(float)) & 15) == 0);
    tile_load<float, TP>(x_in + (in_row * N + n0) * 7, lds, np * 7, in_vec, !(Bx == 1 && B > 1));
    if (threadIdx.x < 7 * 28) {
        // U[i][(j,k)], j <= k: T_ijk + T_ikj folded (element.py:207-217 sums over all j, k)
        const int i = threadIdx.x / 28;
        int r = threadIdx.x - i * 28, j = 0;
        while (r >= 7 - j) { r -= 7 - j; ++j; }
        const int k = j + r;
        const float* Tb = Tt + ((BT == 1) ? 0 : b) * 343 + i * 49;
        U[threadIdx.x] = (j == k) ? Tb[j * 7 + k] : Tb[j * 7 + k] + Tb[k * 7 + j];
    }
    __syncthreads();
    unsigned long long groups;
    {
        const int lane = threadIdx.x & 63;
        bool nz = false;
        if (lane < 49) {
            const chx_v4f u = *reinterpret_cast<const chx_v4f*>(&U[4 * lane]);
            nz = u[0] != 0.0f || u[1] != 0.0f || u[2] != 0.0f || u[3] != 0.0f;
        }
        groups = __ballot(nz);
    }
    const int p0 = threadIdx.x, p1 = threadIdx.x + CHX_BLOCK;
    const bool on0 = p0 < np, on1 = p1 < np;
    chx_v2f x[7], y[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) x[j] = chx_v2f{on0 ? lds[p0 * 7 + j] : 0.0f, on1 ? lds[p1 * 7 + j] : 0.0f};
    // The products that are skipped are exact zeros for finite coordinates; a non-finite coordinate makes EVERY output NaN in the
    // dense sum (0 * inf, 0 * nan in each row): restored explicitly. (An overflow of x_j x_k at finite coordinates of ~1e19 is not.)
    bool bad0 = false, bad1 = false;
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        bad0 = bad0 || !isfinite(x[j].x);
        bad1 = bad1 || !isfinite(x[j].y);
    }
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        chx_v2f acc = chx_v2f{0.0f, 0.0f};
#pragma unroll
        for (int m = 0; m < 7; ++m) {
            if (!((groups >> (i * 7 + m)) & 1ull)) continue;           // wave-uniform
            const chx_v4f u = *reinterpret_cast<const chx_v4f*>(&U[i * 28 + 4 * m]);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                constexpr int kJ[28] = {0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 5, 5, 6};
                constexpr int kK[28] = {0, 1, 2, 3, 4, 5, 6, 1, 2, 3, 4, 5, 6, 2, 3, 4, 5, 6, 3, 4, 5, 6, 4, 5, 6, 5, 6, 6};
                const int c = 4 * m + e;
                const chx_v2f uu = {u[e], u[e]};
                const chx_v2f q = x[kJ[c]] * x[kK[c]];
                acc = __builtin_elementwise_fma(uu, q, acc);
            }
        }
        y[i] = chx_v2f{bad0 ? __builtin_nanf("") : acc.x, bad1 ? __builtin_nanf("") : acc.y};
    }
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        if (on0) lds[p0 * 7 + j] = y[j].x;
        if (on1) lds[p1 * 7 + j] = y[j].y;
    }
    __syncthreads();
    tile_store<float, TP>(x_out + (b * N + n0) * 7, lds, np * 7, out_vec, true);
}

// Backward of second_order_kernel (arithmetic in fp64). With h_c = sum_i dY_i U_ic (28 values per particle)
//   dx_m = sum_{k >= m} h_(m,k) x_k + sum_{j <= m} h_(j,m) x_j
// and dU_ic = sum_n dY_i x_j x_k: 196 lanes each own one (i, c) and walk the tile's particles in LDS; a workgroup
// strides over the tiles of its batch row and writes one 196-vector of partial sums (the caller adds them up and
// unfolds dT_ijk = dT_ikj = dU_i(jk)).
constexpr int kSoBwdBlocks = 256;

template <typename T>
__global__ __launch_bounds__(CHX_BLOCK) void second_order_bwd_kernel(const T* __restrict__ x_in, const T* __restrict__ Tt,
                                                                     const T* __restrict__ dY, T* __restrict__ dx,
                                                                     double* __restrict__ dU_partials, int64_t B,
                                                                     int64_t Bx, int64_t BT, int64_t N) {
    constexpr int TP = CHX_BLOCK;
    __shared__ double xs[TP * 7], gs[TP * 7];
    __shared__ double U[7 * 28];
    const int64_t b = blockIdx.y;
    const int64_t in_row = (Bx == 1) ? 0 : b;
    int ui = 0, uj = 0, uk = 0;
    if (threadIdx.x < 7 * 28) {
        ui = threadIdx.x / 28;
        int r = threadIdx.x - ui * 28;
        while (r >= 7 - uj) { r -= 7 - uj; ++uj; }
        uk = uj + r;
        const T* Tb = Tt + ((BT == 1) ? 0 : b) * 343 + ui * 49;
        U[threadIdx.x] = (uj == uk) ? (double)Tb[uj * 7 + uk] : (double)Tb[uj * 7 + uk] + (double)Tb[uk * 7 + uj];
    }
    double dU = 0.0;
    const int64_t tiles = (N + TP - 1) / TP;
    for (int64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
        const int64_t n0 = t * TP;
        const int np = (int)((N - n0 < TP) ? (N - n0) : TP);
        __syncthreads();
        for (int e = threadIdx.x; e < np * 7; e += TP) {
            xs[e] = (double)x_in[(in_row * N + n0) * 7 + e];
            gs[e] = (double)dY[(b * N + n0) * 7 + e];
        }
        __syncthreads();
        const int p = threadIdx.x;
        if (dx && p < np) {
            double x[7], g[7], h[28];
#pragma unroll
            for (int j = 0; j < 7; ++j) { x[j] = xs[p * 7 + j]; g[j] = gs[p * 7 + j]; }
#pragma unroll
            for (int c = 0; c < 28; ++c) {
                double acc = 0.0;
#pragma unroll
                for (int i = 0; i < 7; ++i) acc = fma(g[i], U[i * 28 + c], acc);
                h[c] = acc;
            }
            double d[7] = {0, 0, 0, 0, 0, 0, 0};
            {
                int c = 0;
#pragma unroll
                for (int j = 0; j < 7; ++j)
#pragma unroll
                    for (int k = j; k < 7; ++k) {
                        d[j] = fma(h[c], x[k], d[j]);
                        d[k] = fma(h[c], x[j], d[k]);
                        ++c;
                    }
            }
#pragma unroll
            for (int j = 0; j < 7; ++j) dx[(b * N + n0 + p) * 7 + j] = (T)d[j];
        }
        if (dU_partials && threadIdx.x < 7 * 28) {
            for (int q = 0; q < np; ++q) dU = fma(gs[q * 7 + ui], xs[q * 7 + uj] * xs[q * 7 + uk], dU);
        }
    }
    if (dU_partials && threadIdx.x < 7 * 28) dU_partials[(b * gridDim.x + blockIdx.x) * 196 + threadIdx.x] = dU;
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
    if (dtype == CHX_F32) {
        const int64_t tiles2 = ((N + 2 * CHX_BLOCK - 1) / (2 * CHX_BLOCK)) * B;
        hipLaunchKernelGGL(second_order_pk_kernel, dim3((unsigned)tiles2), dim3(CHX_BLOCK), 0, s, (const float*)x_in, (const float*)T,
                           (float*)x_out, B, Bx, BT, N, (int)chx_aligned16(x_in), (int)chx_aligned16(x_out));
    } else
        hipLaunchKernelGGL(second_order_kernel<double>, dim3((unsigned)tiles), dim3(CHX_BLOCK), 0, s,
                           (const double*)x_in, (const double*)T, (double*)x_out, B, Bx, BT, N,
                           (int)chx_aligned16(x_in), (int)chx_aligned16(x_out));
    CHX_CHECK_LAUNCH();
    return CHX_OK;
}

extern "C" int64_t chx_second_order_bwd_partials_count(int64_t B) { return B * kSoBwdBlocks * 196; }

extern "C" int chx_apply_second_order_bwd(const void* x_in, const void* T, const void* dY, void* dx, double* dU_partials,
                                          int64_t B, int64_t Bx, int64_t BT, int64_t N, int dtype, void* stream) {
    if (dtype != CHX_F32 && dtype != CHX_F64) return CHX_ERR_DTYPE;
    if (B < 0 || N < 0 || B > 65535) return CHX_ERR_INVALID_ARG;
    if (B == 0 || N == 0) return CHX_OK;
    if (!x_in || !T || !dY || (!dx && !dU_partials)) return CHX_ERR_INVALID_ARG;
    if (!chx_bcast_ok(Bx, B) || !chx_bcast_ok(BT, B)) return CHX_ERR_INVALID_ARG;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(kSoBwdBlocks, (unsigned)B);
    if (dtype == CHX_F32)
        hipLaunchKernelGGL(second_order_bwd_kernel<float>, grid, dim3(CHX_BLOCK), 0, s, (const float*)x_in, (const float*)T,
                           (const float*)dY, (float*)dx, dU_partials, B, Bx, BT, N);
    else
        hipLaunchKernelGGL(second_order_bwd_kernel<double>, grid, dim3(CHX_BLOCK), 0, s, (const double*)x_in,
                           (const double*)T, (const double*)dY, (double*)dx, dU_partials, B, Bx, BT, N);
    CHX_CHECK_LAUNCH();
    return CHX_OK;
}

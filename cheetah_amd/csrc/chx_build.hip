// chx_build.hip — per-element 7x7 first-order transfer maps, segment composition and cavity
// coefficients, built on device in fp64 (one thread per batch row; B is small, this is latency-
// not bandwidth-work) so that a parameter change never costs a host round trip.
//
// Reference arithmetic being replaced (all under cheetah/):
//   drift_matrix            track_methods.py:284-299
//   base_rmatrix            track_methods.py:17-77   (complex sqrt/cos/sinc -> real branches here)
//   rotation / misalignment track_methods.py:302-382
//   Quadrupole              accelerator/quadrupole.py:93-110
//   Dipole + edge maps      accelerator/dipole.py:372-394, 430-466
//   correctors              accelerator/{horizontal,vertical,combined}_corrector.py
//   Cavity R                accelerator/cavity.py:253-358 ; track coefficients cavity.py:100-226
//   si1mdiv / log1pdiv      utils/autograd.py:77-146
//   Segment composition     accelerator/segment.py:534-543
//
// All builders are templates over a scalar S that is either `double` or a forward-mode dual
// number; chx_build_rmatrix_vjp seeds one input at a time and contracts dR/dtheta with the
// incoming cotangent, which replaces torch autograd through ~40 tiny ops per element.
#include "chx_common.h"
#include "chx_dual.h"
#include "chx_moments_dev.h"

namespace {

constexpr double kSpeedOfLight = 299792458.0;  // scipy.constants.speed_of_light (cavity.py:7)
constexpr double kPi = 3.14159265358979323846;

// ---- singularity-free "sinc family" as functions of u = k^2 L^2 (any sign) --------------------
// C(u) = cos(sqrt u), S(u) = sin(sqrt u)/sqrt u, G(u) = (1-cos sqrt u)/u, F(u) = (1-S(u))/u.
// |u| < kSeries uses the Taylor series (exact limits 1, 1, 1/2, 1/6 at u = 0 and correct
// derivatives there; F is utils/autograd.py:108-146 `si1mdiv`), otherwise trig (u>0) or
// hyperbolic (u<0) closed forms — the real restatement of the reference's complex sqrt.
constexpr double kSeries = 0.05;

template <typename S>
__device__ __forceinline__ void sinc_family(S u, S& C, S& Sn, S& G, S& F) {
    const double uv = val(u);
    if (fabs(uv) < kSeries) {
        // Horner in u; coefficients (-1)^n / (2n+m)!
        C = 1.0 + u * (-1.0 / 2 + u * (1.0 / 24 + u * (-1.0 / 720 + u * (1.0 / 40320 + u * (-1.0 / 3628800 + u * (1.0 / 479001600))))));
        Sn = 1.0 + u * (-1.0 / 6 + u * (1.0 / 120 + u * (-1.0 / 5040 + u * (1.0 / 362880 + u * (-1.0 / 39916800 + u * (1.0 / 6227020800.0))))));
        G = 1.0 / 2 + u * (-1.0 / 24 + u * (1.0 / 720 + u * (-1.0 / 40320 + u * (1.0 / 3628800 + u * (-1.0 / 479001600 + u * (1.0 / 87178291200.0))))));
        F = 1.0 / 6 + u * (-1.0 / 120 + u * (1.0 / 5040 + u * (-1.0 / 362880 + u * (1.0 / 39916800 + u * (-1.0 / 6227020800.0 + u * (1.0 / 1307674368000.0))))));
    } else if (uv > 0) {
        const S a = m_sqrt(u);
        C = m_cos(a);
        Sn = m_sin(a) / a;
        const S sh = m_sin(0.5 * a);
        G = 2.0 * sh * sh / u;
        F = (1.0 - Sn) / u;
    } else {
        const S a = m_sqrt(-u);
        C = m_cosh(a);
        Sn = m_sinh(a) / a;
        const S sh = m_sinh(0.5 * a);
        G = 2.0 * sh * sh / (-u);  // (1 - cosh a)/u = (cosh a - 1)/a^2
        F = (1.0 - Sn) / u;
    }
}

// log(1+x)/x with its limit (utils/autograd.py:77-105)
template <typename S>
__device__ __forceinline__ S log1pdiv(S x) {
    const double xv = val(x);
    if (fabs(xv) < 1e-4) return 1.0 + x * (-1.0 / 2 + x * (1.0 / 3 + x * (-1.0 / 4 + x * (1.0 / 5))));
    return m_log1p(x) / x;
}

// ---------------------------------------------------------------------------------------------
template <typename S>
struct Mat7 {
    S m[49];
    __device__ __forceinline__ S& operator()(int i, int j) { return m[i * 7 + j]; }
    __device__ __forceinline__ const S& operator()(int i, int j) const { return m[i * 7 + j]; }
};

// The dual-number matrices of the builders' VJP kernels live in LDS and are SHARED by the lanes of a wave: a VJP kernel runs one
// wave per (element, input) pair, every lane executes the builder's scalar code redundantly (identical values: the redundant
// stores to a shared matrix all carry the same value), and the dense 7x7 products are split over the lanes (matmul7<Dual>
// below: lane 7 i + j forms entry (i, j)). One lane doing everything — matrices in its stack, i.e. scratch memory, 2.7 KB of it,
// or in a private LDS slab — took 32-35 us for ONE quadrupole: ~700 dependent dual-number multiply-adds in a single thread.
// The matrices are handed out from the workgroup's slab in stack order as the builders' locals come and go (every lane keeps its
// own, identical, depth counter). ttensor_vjp_kernel fills its tensors from thread 0 alone: it never calls matmul7.
constexpr int kDualArenaLanes = 64;
constexpr int kDualArenaMats = 8;     // deepest nesting: dipole_map's seven locals + the caller's result

__device__ __forceinline__ Dual* dual_arena_slab(int*& depth) {
    __shared__ Dual arena[kDualArenaMats * 49];
    __shared__ int depths[kDualArenaLanes];
    depth = &depths[threadIdx.x];
    return arena;
}
// at the entry of a kernel whose threads build dual-number maps (threadIdx.x < kDualArenaLanes)
__device__ __forceinline__ void dual_arena_reset() {
    int* depth;
    dual_arena_slab(depth);
    *depth = 0;
}

template <>
struct Mat7<Dual> {
    Dual* m;
    __device__ __forceinline__ Mat7() {
        int* depth;
        Dual* slab = dual_arena_slab(depth);
        if (threadIdx.x >= kDualArenaLanes || *depth >= kDualArenaMats) __builtin_trap();
        m = slab + *depth * 49;
        *depth += 1;
    }
    __device__ __forceinline__ ~Mat7() {
        int* depth;
        dual_arena_slab(depth);
        *depth -= 1;
    }
    Mat7(const Mat7&) = delete;
    Mat7& operator=(const Mat7&) = delete;
    __device__ __forceinline__ Dual& operator()(int i, int j) { return m[i * 7 + j]; }
    __device__ __forceinline__ const Dual& operator()(int i, int j) const { return m[i * 7 + j]; }
};

template <typename S>
__device__ void eye7(Mat7<S>& M) {
#pragma unroll
    for (int k = 0; k < 49; ++k) M.m[k] = cst<S>(0.0);
#pragma unroll
    for (int i = 0; i < 7; ++i) M(i, i) = cst<S>(1.0);
}

// fully unrolled: every index is a constant, so the matrices of the dual-number instantiation (98 doubles each: they live in
// scratch) are read in batches instead of one dependent load per multiply-add
template <typename S>
__device__ void matmul7(const Mat7<S>& A, const Mat7<S>& Bm, Mat7<S>& Cm) {
#pragma unroll
    for (int i = 0; i < 7; ++i)
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            S acc = A(i, 0) * Bm(0, j);
#pragma unroll
            for (int k = 1; k < 7; ++k) acc = acc + A(i, k) * Bm(k, j);
            Cm(i, j) = acc;
        }
}

// C = A B for the wave-shared dual-number matrices (see Mat7<Dual>): lane 7 i + j forms entry (i, j); A, B, C are distinct
template <>
__device__ __forceinline__ void matmul7<Dual>(const Mat7<Dual>& A, const Mat7<Dual>& Bm, Mat7<Dual>& Cm) {
    const int lane = threadIdx.x & 63;
    if (lane < 49) {
        const int i = lane / 7, j = lane - 7 * i;
        Dual acc = A(i, 0) * Bm(0, j);
#pragma unroll
        for (int k = 1; k < 7; ++k) acc = acc + A(i, k) * Bm(k, j);
        Cm.m[lane] = acc;
    }
    chx_wave_sync();          // the other lanes read these entries next
}

template <typename S>
__device__ __forceinline__ void rel_factors(S energy, double mass, S& gamma, S& igamma2, S& beta) {
    gamma = energy / mass;                 // utils/physics.py:15-17
    igamma2 = 1.0 / (gamma * gamma);
    beta = m_sqrt(1.0 - igamma2);
}

// track_methods.py:284-299 (+ corrector kicks in column 6)
template <typename S>
__device__ void drift_map(S L, S energy, double mass, Mat7<S>& R) {
    S g, ig2, beta;
    rel_factors(energy, mass, g, ig2, beta);
    eye7(R);
    R(0, 1) = L;
    R(2, 3) = L;
    R(4, 5) = -L / (beta * beta) * ig2;
}

// track_methods.py:17-77
template <typename S>
__device__ void base_rmatrix(S L, S k1, S hx, S energy, double mass, Mat7<S>& R) {
    S g, ig2, beta;
    rel_factors(energy, mass, g, ig2, beta);
    const S kx2 = k1 + hx * hx;
    const S ky2 = -k1;
    const S L2 = L * L;
    S Cx, Sx, Gx, Fx, Cy, Sy, Gy, Fy;
    sinc_family<S>(kx2 * L2, Cx, Sx, Gx, Fx);
    sinc_family<S>(ky2 * L2, Cy, Sy, Gy, Fy);
    const S sx = Sx * L, sy = Sy * L;
    const S dx = hx * L2 * Gx;  // = hx * 0.5 L^2 sinc^2(kx L / 2)
    const S b2 = beta * beta;
    const S r56 = hx * hx * L2 * L * Fx / b2 - L / b2 * ig2;
    eye7(R);
    R(0, 0) = Cx;
    R(0, 1) = sx;
    R(0, 5) = dx / beta;
    R(1, 0) = -kx2 * sx;
    R(1, 1) = Cx;
    R(1, 5) = sx * hx / beta;
    R(2, 2) = Cy;
    R(2, 3) = sy;
    R(3, 2) = -ky2 * sy;
    R(3, 3) = Cy;
    R(4, 0) = sx * hx / beta;
    R(4, 1) = dx / beta;
    R(4, 5) = r56;
}

// track_methods.py:302-323
template <typename S>
__device__ void rotation_map(S angle, Mat7<S>& R) {
    const S cs = m_cos(angle), sn = m_sin(angle);
    eye7(R);
    R(0, 0) = cs; R(0, 2) = sn; R(1, 1) = cs; R(1, 3) = sn;
    R(2, 0) = -sn; R(2, 2) = cs; R(3, 1) = -sn; R(3, 3) = cs;
}

template <typename S>
__device__ void transpose7(const Mat7<S>& A, Mat7<S>& At) {
#pragma unroll
    for (int i = 0; i < 7; ++i)
#pragma unroll
        for (int j = 0; j < 7; ++j) At(i, j) = A(j, i);
}

// exactly zero, and constant with respect to the input a dual-number build differentiates
__device__ __forceinline__ bool plain_zero(double x) { return x == 0.0; }
__device__ __forceinline__ bool plain_zero(Dual x) { return x.v == 0.0 && x.d == 0.0; }
__device__ __forceinline__ bool finite_entry(double x) { return isfinite(x); }
__device__ __forceinline__ bool finite_entry(Dual x) { return isfinite(x.v) && isfinite(x.d); }

template <typename S>
__device__ bool all_finite7(const Mat7<S>& A) {
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 49; ++k) ok = ok && finite_entry(A.m[k]);
    return ok;
}
template <>
__device__ __forceinline__ bool all_finite7<Dual>(const Mat7<Dual>& A) {     // (wave-shared matrix: a lane per entry)
    const int lane = threadIdx.x & 63;
    return !__any(lane < 49 && !finite_entry(A.m[lane]));
}
template <typename S>
__device__ void copy7(const Mat7<S>& A, Mat7<S>& B) {
#pragma unroll
    for (int k = 0; k < 49; ++k) B.m[k] = A.m[k];
}
template <>
__device__ __forceinline__ void copy7<Dual>(const Mat7<Dual>& A, Mat7<Dual>& B) {
    const int lane = threadIdx.x & 63;
    if (lane < 49) B.m[lane] = A.m[lane];
    chx_wave_sync();
}

// quadrupole.py:93-110 with track_methods.py:345-382
template <typename S>
__device__ void quadrupole_map(const S* p, S energy, double mass, Mat7<S>& R) {
    const S L = p[0], k1 = p[1], tilt = p[2], mx = p[3], my = p[4];
    Mat7<S> base, entry, exitm, tmp;
    base_rmatrix<S>(L, k1, cst<S>(0.0), energy, mass, base);
    // An upright, centred quadrupole — nearly every one of a lattice: the rotation in front is the identity with a column of zeros,
    // the one behind its transpose, and exit (base entry) returns base's entries as they are (products with 1 and sums with 0 are
    // exact; a zero may change its sign) when all of them are finite. The two dense products and the four sines and cosines are
    // two thirds of this builder's time — 4.7 of the 10.9 us a wave spends on the dual-number form in the backward pass of an
    // optimisation step, 700 dependent multiply-adds on the one lane that builds the map in the forward pass.
    if (plain_zero(tilt) && plain_zero(mx) && plain_zero(my) && all_finite7(base)) {
        copy7(base, R);
        return;
    }
    rotation_map<S>(tilt, entry);
    transpose7(entry, exitm);  // tm_exit = tm_entry.clone().mT (before the misalignment is added)
    const S cs = m_cos(tilt), sn = m_sin(tilt);
    entry(0, 6) = -mx * cs - my * sn;
    entry(2, 6) = mx * sn - my * cs;
    exitm(0, 6) = mx;
    exitm(2, 6) = my;
    matmul7(base, entry, tmp);
    matmul7(exitm, tmp, R);
}

// dipole.py:372-394, 430-466 ; hx = angle / length (dipole.py:133-135)
template <typename S>
__device__ void dipole_map(const S* p, S energy, double mass, Mat7<S>& R) {
    const S L = p[0], angle = p[1], k1 = p[2], e1 = p[3], e2 = p[4], tilt = p[5], fint = p[6],
            fint_exit = p[7], gap = p[8];
    const S hx = angle / L;
    Mat7<S> base, enter, exitm, rot, rotT, t1, t2;
    base_rmatrix<S>(L, k1, hx, energy, mass, base);
    {
        const S sec = 1.0 / m_cos(e1), s1 = m_sin(e1);
        const S phi = fint * hx * gap * sec * (1.0 + s1 * s1);
        eye7(enter);
        enter(1, 0) = hx * m_tan(e1);
        enter(3, 2) = -hx * m_tan(e1 - phi);
    }
    {
        const S sec = 1.0 / m_cos(e2), s2 = m_sin(e2);
        const S phi = fint_exit * hx * gap * sec * (1.0 + s2 * s2);  // NB: reference uses `gap` here too
        eye7(exitm);
        exitm(1, 0) = hx * m_tan(e2);
        exitm(3, 2) = -hx * m_tan(e2 - phi);
    }
    matmul7(base, enter, t1);
    matmul7(exitm, t1, t2);
    if (plain_zero(tilt) && all_finite7(t2)) {      // an upright dipole: rot^T (t2 rot) returns t2's entries (see quadrupole_map)
        copy7(t2, R);
        return;
    }
    rotation_map<S>(tilt, rot);
    transpose7(rot, rotT);
    matmul7(t2, rot, t1);
    matmul7(rotT, t1, R);
}

// cavity.py:253-358
template <typename S>
__device__ void cavity_map(const S* p, S energy, double mass, double nq, bool standing, Mat7<S>& R) {
    const S L = p[0], V = p[1], phase = p[2], freq = p[3];
    const S phi = phase * (kPi / 180.0);
    const S veff = -V * nq;
    const S cphi = m_cos(phi), sphi = m_sin(phi);
    const S dEn = veff * cphi;  // delta_energy
    const S Ei = energy / mass;
    const S dE = dEn / mass;
    const S Ef = Ei + dE;
    const S k = 2.0 * kPi * freq / kSpeedOfLight;
    S r11, r12, r21, r22, r55, r56, r65, r66;
    if (standing) {
        const S l1p = log1pdiv<S>(dEn / energy);
        const S alpha = 0.35355339059327378 /* sqrt(1/8) */ * veff / energy * l1p;
        const S beta0 = m_sqrt(1.0 - 1.0 / (Ei * Ei));
        const S beta1 = m_sqrt(1.0 - 1.0 / (Ef * Ef));
        const S ca = m_cos(alpha), sa = m_sin(alpha);
        S Ca, Sa, Ga, Fa;
        sinc_family<S>(alpha * alpha, Ca, Sa, Ga, Fa);  // Sa = sin(alpha)/alpha
        const double rt2 = 1.4142135623730951;
        r11 = ca - rt2 * cphi * sa;
        r12 = Sa * l1p * L;
        r21 = -(veff / ((energy + dEn) * rt2 * L) * (0.5 + cphi * cphi) * sa);
        r22 = Ei / Ef * (ca + rt2 * cphi * sa);
        if (val(dE) != 0.0) {
            r55 = 1.0 + k * L * beta0 * (sphi / cphi) * (Ei * Ef * (beta0 * beta1 - 1.0) + 1.0) / (beta1 * Ef * dE);
        } else {
            r55 = cst<S>(1.0);
        }
        r56 = -L / (Ef * Ef * Ei * beta1) * (Ef + Ei) / (beta1 + beta0);
        r65 = k * sphi * veff / (beta1 * (energy + dEn));
        r66 = Ei / Ef * beta0 / beta1;
    } else {
        const S Ep = dE / L;
        // M = F_exit * Body * F_entry (2x2)
        const S b01 = L * log1pdiv<S>(dE / Ei), b11 = Ei / Ef;
        const S fe = -Ep / (2.0 * Ei), fx = Ep / (2.0 * Ef);
        // Body * F_entry
        const S m00 = 1.0 + b01 * fe, m01 = b01, m10 = b11 * fe, m11 = b11;
        r11 = m00;
        r12 = m01;
        r21 = fx * m00 + m10;
        r22 = fx * m01 + m11;
        r55 = cst<S>(1.0);
        r56 = cst<S>(0.0);
        r65 = k * sphi * veff / (energy + dEn);
        r66 = r22;
    }
    eye7(R);
    R(0, 0) = r11; R(0, 1) = r12; R(1, 0) = r21; R(1, 1) = r22;
    R(2, 2) = r11; R(2, 3) = r12; R(3, 2) = r21; R(3, 3) = r22;
    R(4, 4) = r55; R(4, 5) = r56; R(5, 4) = r65; R(5, 5) = r66;
}

// solenoid.py:75-116 (row f3): thin-lens-free hard-edge solenoid, then the (un-rotated) misalignment shift
template <typename S>
__device__ void solenoid_map(const S* p, S energy, double mass, Mat7<S>& R) {
    const S L = p[0], k = p[1], mx = p[2], my = p[3];
    const S gamma = energy / mass;
    const S c = m_cos(L * k), s = m_sin(L * k);
    S Cu, Su, Gu, Fu;
    const S lk = L * k;
    sinc_family<S>(lk * lk, Cu, Su, Gu, Fu);  // Su = sin(Lk)/(Lk) with its limit 1
    const S s_k = Su * L;
    Mat7<S> body, entry, exitm, tmp;
    eye7(body);
    body(0, 0) = c * c;      body(0, 1) = c * s_k;  body(0, 2) = s * c;       body(0, 3) = s * s_k;
    body(1, 0) = -k * s * c; body(1, 1) = c * c;    body(1, 2) = -k * s * s;  body(1, 3) = s * c;
    body(2, 0) = -s * c;     body(2, 1) = -s * s_k; body(2, 2) = c * c;       body(2, 3) = c * s_k;
    body(3, 0) = k * s * s;  body(3, 1) = -s * c;   body(3, 2) = -k * s * c;  body(3, 3) = c * c;
    body(4, 5) = L / (1.0 - gamma * gamma);
    eye7(entry);
    eye7(exitm);
    entry(0, 6) = -mx; entry(2, 6) = -my;  // track_methods.py:326-342
    exitm(0, 6) = mx;  exitm(2, 6) = my;
    matmul7(body, entry, tmp);
    matmul7(exitm, tmp, R);
}

// undulator.py:79-125 (row f3)
template <typename S>
__device__ void undulator_map(const S* p, S energy, double mass, Mat7<S>& R) {
    const S L = p[0], kx = p[1], ky = p[2], period = p[3];
    S g, ig2, beta;
    rel_factors(energy, mass, g, ig2, beta);
    eye7(R);
    R(4, 5) = -L * ig2 * (1.0 / (beta * beta) + 0.5 * (kx * kx + ky * ky));
    S sf = cst<S>(0.0);
    if (val(period) > 0.0) sf = (1.4142135623730951 * kPi) / (period * g * beta);
    {
        const S w = sf * kx, wl = w * L;
        S Cu, Su, Gu, Fu;
        sinc_family<S>(wl * wl, Cu, Su, Gu, Fu);
        R(2, 2) = m_cos(wl); R(2, 3) = Su * L; R(3, 2) = -m_sin(wl) * w; R(3, 3) = m_cos(wl);
    }
    {
        const S w = sf * ky, wl = w * L;
        S Cu, Su, Gu, Fu;
        sinc_family<S>(wl * wl, Cu, Su, Gu, Fu);
        R(0, 0) = m_cos(wl); R(0, 1) = Su * L; R(1, 0) = -m_sin(wl) * w; R(1, 1) = m_cos(wl);
    }
}

template <typename S>
__device__ void build_kind(int kind, const S* p, S energy, double mass, double nq, Mat7<S>& R) {
    switch (kind) {
        case CHX_IDENTITY: eye7(R); break;
        case CHX_DRIFT: drift_map<S>(p[0], energy, mass, R); break;
        case CHX_QUADRUPOLE: quadrupole_map<S>(p, energy, mass, R); break;
        case CHX_DIPOLE: dipole_map<S>(p, energy, mass, R); break;
        case CHX_HCOR: drift_map<S>(p[0], energy, mass, R); R(1, 6) = p[1]; break;
        case CHX_VCOR: drift_map<S>(p[0], energy, mass, R); R(3, 6) = p[1]; break;
        case CHX_CCOR: drift_map<S>(p[0], energy, mass, R); R(1, 6) = p[1]; R(3, 6) = p[2]; break;
        case CHX_CAVITY_SW: cavity_map<S>(p, energy, mass, nq, true, R); break;
        case CHX_CAVITY_TW: cavity_map<S>(p, energy, mass, nq, false, R); break;
        case CHX_SOLENOID: solenoid_map<S>(p, energy, mass, R); break;
        case CHX_UNDULATOR: undulator_map<S>(p, energy, mass, R); break;
        default: eye7(R); break;
    }
}

__host__ __device__ inline int kind_num_params(int kind) {
    switch (kind) {
        case CHX_IDENTITY: return 0;
        case CHX_DRIFT: return 1;
        case CHX_QUADRUPOLE: return 5;
        case CHX_DIPOLE: return 9;
        case CHX_HCOR: return 2;
        case CHX_VCOR: return 2;
        case CHX_CCOR: return 3;
        case CHX_CAVITY_SW: return 4;
        case CHX_CAVITY_TW: return 4;
        case CHX_SOLENOID: return 4;
        case CHX_UNDULATOR: return 4;
        default: return -1;
    }
}

template <typename T>
__global__ void build_kernel(int kind, const T* __restrict__ params, const T* __restrict__ energy,
                             double mass, double nq, int64_t B, int64_t Bp, int64_t Be, int P,
                             T* __restrict__ R_out) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double p[CHX_MAX_PARAMS];
    for (int k = 0; k < P; ++k) p[k] = (double)params[(Bp == 1 ? 0 : b) * P + k];
    const double en = (double)energy[Be == 1 ? 0 : b];
    Mat7<double> R;
    build_kind<double>(kind, p, en, mass, nq, R);
    for (int k = 0; k < 49; ++k) R_out[b * 49 + k] = (T)R.m[k];
}

// one WAVE per (b, input k) with k in [0, P] (k == P is the energy): see Mat7<Dual>
template <typename T>
__global__ __launch_bounds__(64) void build_vjp_kernel(int kind, const T* __restrict__ params, const T* __restrict__ energy,
                                                       double mass, double nq, const T* __restrict__ dR, int64_t B,
                                                       int64_t Bp, int64_t Be, int P, T* __restrict__ dparams,
                                                       T* __restrict__ denergy) {
    const int64_t idx = blockIdx.x;
    dual_arena_reset();
    const int64_t b = idx / (P + 1);
    const int k = (int)(idx - b * (P + 1));
    Dual p[CHX_MAX_PARAMS];
    for (int q = 0; q < P; ++q) p[q] = mk((double)params[(Bp == 1 ? 0 : b) * P + q], q == k ? 1.0 : 0.0);
    const Dual en = mk((double)energy[Be == 1 ? 0 : b], k == P ? 1.0 : 0.0);
    Mat7<Dual> R;
    build_kind<Dual>(kind, p, en, mass, nq, R);
    chx_wave_sync();
    const int lane = threadIdx.x;
    double acc = lane < 49 ? (double)dR[b * 49 + lane] * R.m[lane].d : 0.0;
    acc = chx_wave_sum(acc);
    if (lane == 0) {
        if (k < P) dparams[b * P + k] = (T)acc;
        else denergy[b] = (T)acc;
    }
}

// ---- composition (segment.py:534-543): tm = R_e @ tm for e = 0..E-1 -------------------------
// One workgroup (4 waves) per batch row. Wave w composes its contiguous chunk of elements
// sequentially (49 lanes = 49 matrix entries, operands exchanged through LDS), then wave 0
// multiplies the four partial products together: sequential depth E/4 + 4 instead of E.
// The per-element device pointers travel BY VALUE in the kernel arguments (<= 192 per launch,
// longer lattices chain launches through R_out): no pointer table in device memory, no H2D copy.
constexpr int kComposeChunk = 192;
struct ComposeArgs {
    const void* ptr[kComposeChunk];
    uint8_t bcast[kComposeChunk];
};

// The composition of one batch row by one 256-thread workgroup; `map_of(e)` returns element e's 49 entries (row-major).
// Shared by compose_kernel and run_map_kernel so that both associate the products identically (bit-identical results).
template <typename T, typename MapOf>
__device__ __forceinline__ void compose_block(MapOf map_of, int E, int has_init, T* __restrict__ R_row) {
    __shared__ double cur[4][49];
    __shared__ double nxt[4][49];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane / 7, j = lane - 7 * i;
    const int per = (E + 3) / 4;
    const int e0 = wave * per, e1 = (e0 + per < E) ? e0 + per : E;
    double acc = (lane < 49) ? ((i == j) ? 1.0 : 0.0) : 0.0;
    // a follow-up launch continues from the product accumulated so far (wave 0 only)
    if (has_init && wave == 0 && lane < 49) acc = (double)R_row[lane];
    if (lane < 49) cur[wave][lane] = acc;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int e = e0; e < e1; ++e) {
        const T* Re = map_of(e);
        if (lane < 49) nxt[wave][lane] = (double)Re[lane];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (lane < 49) {
            double s = nxt[wave][i * 7] * cur[wave][j];
            for (int k = 1; k < 7; ++k) s = fma(nxt[wave][i * 7 + k], cur[wave][k * 7 + j], s);
            acc = s;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (lane < 49) cur[wave][lane] = acc;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    __syncthreads();
    if (wave == 0) {
        // total = P3 P2 P1 P0 ; start from P0 held in cur[0]
        for (int w = 1; w < 4; ++w) {
            if (lane < 49) {
                double s = cur[w][i * 7] * cur[0][j];
                for (int k = 1; k < 7; ++k) s = fma(cur[w][i * 7 + k], cur[0][k * 7 + j], s);
                acc = s;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (lane < 49) cur[0][lane] = acc;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        if (lane < 49) R_row[lane] = (T)acc;
    }
}

template <typename T>
__global__ __launch_bounds__(CHX_BLOCK) void compose_kernel(ComposeArgs args, int E, int has_init,
                                                           T* __restrict__ R_out) {
    const int64_t b = blockIdx.x;
    compose_block<T>([&](int e) { return (const T*)args.ptr[e] + (args.bcast[e] ? 0 : b * 49); }, E, has_init,
                     R_out + b * 49);
}

// ---- cavity track coefficients (cavity.py:100-226) -------------------------------------------
// One row: [a, b, k*beta0, phi, cos phi, T566, T556, T555] and the outgoing energy; `gain` = any(delta_energy > 0) over the
// batch (cavity.py:157 switches the second-order path-length terms for the whole batch at once).
__device__ __forceinline__ double cavity_coeff_row(double L, double V, double phi_deg, double freq, double E0, double mass, double nq,
                                                   bool gain, double* __restrict__ c) {
    const double phi = phi_deg * (kPi / 180.0);
    const double g0 = E0 / mass, ig2 = 1.0 / (g0 * g0), b0 = sqrt(1.0 - ig2);
    const double cphi = cos(phi), sphi = sin(phi);
    const double dEn = V * cphi * nq * -1.0;
    const double E1 = E0 + dEn;
    const double g1 = E1 / mass, b1 = sqrt(1.0 - 1.0 / (g1 * g1));
    const double k = 2.0 * kPi * freq / kSpeedOfLight;
    double T566 = 1.5 * L * ig2 / (b0 * b0 * b0), T556 = 0.0, T555 = 0.0;
    if (gain) {
        const double dg = V / mass;
        const double b03 = b0 * b0 * b0, b13 = b1 * b1 * b1, g03 = g0 * g0 * g0, g13 = g1 * g1 * g1;
        const double gd = g0 - g1;
        T566 = L * (b03 * g03 - b13 * g13) / (2.0 * b0 * b13 * g0 * gd * g13);
        T556 = b0 * k * L * dg * g0 * (b13 * g13 + b0 * (g0 - g13)) * sphi / (b13 * g13 * gd * gd);
        T555 = b0 * b0 * k * k * L * dg / 2.0 *
               (dg * (2.0 * g0 * g13 * (b0 * b13 - 1.0) + g0 * g0 + 3.0 * g1 * g1 - 2.0) /
                    (b13 * g13 * gd * gd * gd) * sphi * sphi -
                (g1 * g0 * (b1 * b0 - 1.0) + 1.0) / (b1 * g1 * gd * gd) * cphi);
    }
    c[0] = E0 * b0 / (E1 * b1);
    c[1] = V * b0 / (E1 * b1);
    c[2] = b0 * k;
    c[3] = phi;
    c[4] = cphi;
    c[5] = T566;
    c[6] = T556;
    c[7] = T555;
    return E1;
}

// Single workgroup; pass 1 evaluates `any(delta_energy > 0)` over the batch, pass 2 writes the rows.
template <typename T>
__global__ __launch_bounds__(CHX_BLOCK) void cavity_coeff_kernel(const T* __restrict__ params,
                                                                const T* __restrict__ energy,
                                                                double mass, double nq, int64_t B,
                                                                int64_t Bp, int64_t Be,
                                                                double* __restrict__ coeffs,
                                                                T* __restrict__ energy_out) {
    __shared__ int any_gain;
    if (threadIdx.x == 0) any_gain = 0;
    __syncthreads();
    int local = 0;
    for (int64_t b = threadIdx.x; b < B; b += blockDim.x) {
        const T* p = params + (Bp == 1 ? 0 : b) * 4;
        const double V = (double)p[1], phi = (double)p[2] * (kPi / 180.0);
        const double dEn = V * cos(phi) * nq * -1.0;
        if (dEn > 0.0) local = 1;
    }
    if (local) atomicOr(&any_gain, 1);
    __syncthreads();
    const bool gain = any_gain != 0;
    for (int64_t b = threadIdx.x; b < B; b += blockDim.x) {
        const T* p = params + (Bp == 1 ? 0 : b) * 4;
        const double E1 = cavity_coeff_row((double)p[0], (double)p[1], (double)p[2], (double)p[3], (double)energy[Be == 1 ? 0 : b],
                                           mass, nq, gain, coeffs + b * CHX_CAV_NCOEF);
        energy_out[b] = (T)E1;
    }
}

// A cavity whose four settings are device scalars, for one beam: its map (chx_build_rmatrix), its coefficient row and the
// outgoing energy by ONE thread of one launch — what Cavity.track needs before the particle pass.
template <typename T>
__global__ void cavity_prepare_scalars_kernel(const T* __restrict__ L, const T* __restrict__ V, const T* __restrict__ ph,
                                              const T* __restrict__ fr, const T* __restrict__ energy, int kind, double mass,
                                              double nq, T* __restrict__ R_out, double* __restrict__ coeffs,
                                              T* __restrict__ energy_out, const T* __restrict__ s_in, T* __restrict__ s_out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (s_out) *s_out = *s_in + *L;          // incoming.s + length (cavity.py:228-251), in T
    const double p[4] = {(double)*L, (double)*V, (double)*ph, (double)*fr};
    const double E0 = (double)*energy;
    Mat7<double> R;
    build_kind<double>(kind, p, E0, mass, nq, R);
    for (int q = 0; q < 49; ++q) R_out[q] = (T)R.m[q];
    const double dEn = p[1] * cos(p[2] * (kPi / 180.0)) * nq * -1.0;
    *energy_out = (T)cavity_coeff_row(p[0], p[1], p[2], p[3], E0, mass, nq, dEn > 0.0, coeffs);
}

// ---- second-order transfer tensors (track_methods.py:80-296, MAD convention) ------------------
// E(u) = (3 - 4 S + S C) / (2u)                              (utils/autograd.py:227-234, `sipsicos3mdiv`)
// H(u) = (15 - 22.5 S + 9 S C - 1.5 S C^2 + u S^3) / u^3     (j3 of track_methods.py:134-143, = 6 j3 / L^7)
// with S = sin(sqrt u)/sqrt u, C = cos(sqrt u). The closed form of E cancels like u^2 near u = 0, so
// |u| < 1 is evaluated from its Taylor series (Horner, 12 terms, truncation < 1e-17).
template <typename S>
__device__ S ttensor_E(S u, S C, S Sn) {
    if (fabs(val(u)) < 1.0) {
        const double c[12] = {0.0, 1.0 / 20, -1.0 / 168, 1.0 / 2880, -17.0 / 1330560, 31.0 / 94348800, -1.0 / 159667200,
                              5461.0 / 59281238016000.0, -257.0 / 238519804723200.0, 73.0 / 7113748561920000.0,
                              -1271.0 / 15667888932657561600.0, 60787.0 / 112400072777760768000000.0};
        S r = cst<S>(c[11]);
        for (int k = 10; k >= 0; --k) r = r * u + c[k];
        return r;
    }
    return (3.0 - 4.0 * Sn + Sn * C) / (2.0 * u);
}
// NB: the numerator of H does NOT vanish like u^3 (it starts at order u), so j3 has no finite limit at u = 0 —
// the reference says so itself (track_methods.py:130-133) and substitutes L^7/56 at exactly kx2 == 0. Parity with
// the reference means evaluating its closed form as written; there is no cancellation to protect against.
template <typename S>
__device__ S ttensor_H(S u, S C, S Sn) {
    return (15.0 - 22.5 * Sn + 9.0 * Sn * C - 1.5 * Sn * C * C + u * Sn * Sn * Sn) / (u * u * u);
}

#define TT(i, j, k) T[((i) * 7 + (j)) * 7 + (k)]
// T must be zero-filled by the caller. Every jc/js/jd/jf term below carries a factor k2; the reference's own elements only
// combine k2 != 0 with k1 = hx = 0 (Sextupole), custom elements written against track_methods.base_ttensor may combine
// all three (CHX_T_GENERAL).
template <typename S>
__device__ void base_ttensor(S L, S k1, S k2, S hx, S energy, double mass, S* T) {
    S g, ig2, beta;
    rel_factors<S>(energy, mass, g, ig2, beta);
    const S kx2 = k1 + hx * hx, ky2 = -k1;
    const S L2 = L * L, L3 = L2 * L;
    const S a = kx2 * L2, b = ky2 * L2;
    S cx, Sx, Gx, Fx, cy, Sy, Gy, Fy, C4, S4, G4, F4;
    sinc_family<S>(a, cx, Sx, Gx, Fx);
    sinc_family<S>(b, cy, Sy, Gy, Fy);
    sinc_family<S>(4.0 * b, C4, S4, G4, F4);
    const S sx = Sx * L, sy = Sy * L;
    const S dx = L2 * Gx;                 // 0.5 L^2 sinc^2(kx L / 2), no hx factor here (track_methods.py:120)
    const S fx = L3 * Fx;                 // si1mdiv
    // sicos1mdiv(v) = (1 - S(v) C(v)) / v = 4 F(4v). At exactly v == 0 the reference substitutes 1/6 (utils/autograd.py
    // `SiCos1MDiv.forward`: `.where(x != 0, 1.0 / 6.0)`) although the limit is 2/3; parity means taking what it takes. The
    // value only reaches a tensor entry through jf when k1 == 0 exactly while k2 != 0 and hx != 0 (CHX_T_GENERAL).
    const S f2y = (val(b) != 0.0) ? L3 * 4.0 * F4 : L3 * (1.0 / 6.0);
    const S j1 = fx;
    const S j2 = L3 * ttensor_E<S>(a, cx, Sx);
    const S L7 = L3 * L3 * L;
    const S j3 = (val(kx2) != 0.0) ? L7 * ttensor_H<S>(a, cx, Sx) / 6.0 : L7 / 56.0;
    const S jden = kx2 - 4.0 * ky2;
    const bool same = val(a) == val(b), bzero = val(b) == 0.0;
    // The divided differences (utils/autograd.py `cossqrtmcosdivdiff`, `simsidivdiff`, `si2msi2divdiff`). On the diagonal
    // a == b the reference substitutes the limit and gives BOTH arguments the same partial derivative (its backward passes,
    // autograd.py:395-405, 452-470, 560-585): the value below is that limit, and its tangent is limit' * (da + db) — a
    // one-variable expression in `a` alone would have the wrong tangent whenever da != db (general k1, k2, hx).
    const double vb = val(b), vcy = val(cy), vSy = val(Sy);
    const S dab = (a + b) - (val(a) + val(b));      // value 0, tangent da + db
    const S jc = L2 * (!same ? (cy - cx) / (a - b)
                              : 0.5 * val(Sx) + dab * (!bzero ? (vcy - vSy) / (8.0 * vb) : -1.0 / 24.0));
    const S js = L3 * (!same ? (Sx - Sy) / (b - a)
                              : (!bzero ? 0.5 * (vSy - vcy) / vb : 1.0 / 6.0) +
                                    dab * (!bzero ? (3.0 * vcy + (vb - 3.0) * vSy) / (8.0 * vb * vb) : -1.0 / 120.0));
    const S jd = L2 * L2 *
                 (!same ? (Sy * Sy - Sx * Sx) / (a - b)
                        : (!bzero ? (1.0 - vcy * vcy - vb * vSy * vcy) / (vb * vb) : 1.0 / 3.0) +
                              dab * (!bzero ? (5.0 * vb * vSy * vcy - (vb - 2.0) * (2.0 * vcy * vcy - 1.0) - 2.0) / (4.0 * vb * vb * vb)
                                            : -2.0 / 45.0));
    const S jf = (val(jden) != 0.0) ? (f2y - fx) / jden : L2 * L3 / 120.0;
    const S khk = k2 + 2.0 * hx * k1;
    const S b2 = beta * beta, b3 = b2 * beta, hx2 = hx * hx, hx3 = hx2 * hx, dx2 = dx * dx;

    TT(0, 0, 0) = -khk * (sx * sx + dx) / 6.0 - 0.5 * hx * kx2 * sx * sx;
    TT(0, 0, 1) = 2.0 * (-khk * sx * dx / 6.0 + 0.5 * hx * sx * cx);
    TT(0, 1, 1) = -khk * dx2 / 6.0 + 0.5 * hx * dx * cx;
    TT(0, 0, 5) = 2.0 * (-hx / 12.0 / beta * khk * (3.0 * sx * j1 - dx2) + 0.5 * hx2 / beta * sx * sx +
                         0.25 / beta * k1 * L * sx);
    TT(0, 1, 5) = 2.0 * (-hx / 12.0 / beta * khk * (sx * dx2 - 2.0 * cx * j2) + 0.25 * hx2 / beta * (sx * dx + cx * j1) -
                         0.25 / beta * (sx + L * cx));
    TT(0, 5, 5) = -hx2 / 6.0 / b2 * khk * (dx2 * dx - 2.0 * sx * j2) + 0.5 * hx3 / b2 * sx * j1 - 0.5 * hx / b2 * L * sx -
                  0.5 * hx / b2 * ig2 * dx;
    TT(0, 2, 2) = k1 * k2 * jd + 0.5 * (k2 + hx * k1) * dx;
    TT(0, 2, 3) = 2.0 * (0.5 * k2 * js);
    TT(0, 3, 3) = k2 * jd - 0.5 * hx * dx;
    TT(1, 0, 0) = -khk * sx * (1.0 + 2.0 * cx) / 6.0;
    TT(1, 0, 1) = -2.0 * khk * dx * (1.0 + 2.0 * cx) / 6.0;
    TT(1, 1, 1) = -khk * sx * dx / 3.0 - 0.5 * hx * sx;
    TT(1, 0, 5) = 2.0 * (-hx / 12.0 / beta * khk * (3.0 * cx * j1 + sx * dx) - 0.25 / beta * k1 * (sx - L * cx));
    TT(1, 1, 5) = 2.0 * (-hx / 12.0 / beta * khk * (3.0 * sx * j1 + dx2) + 0.25 / beta * k1 * L * sx);
    TT(1, 5, 5) = -hx2 / 6.0 / b2 * khk * (sx * dx2 - 2.0 * cx * j2) - 0.5 * hx / b2 * k1 * (cx * j1 - sx * dx) -
                  0.5 * hx / b2 * ig2 * sx;
    TT(1, 2, 2) = k1 * k2 * js + 0.5 * (k2 + hx * k1) * sx;
    TT(1, 2, 3) = 2.0 * (0.5 * k2 * jc);
    TT(1, 3, 3) = k2 * js - 0.5 * hx * sx;
    TT(2, 0, 2) = 2.0 * (0.5 * k2 * (cy * jc - 2.0 * k1 * sy * js) + 0.5 * hx * k1 * sx * sy);
    TT(2, 0, 3) = 2.0 * (0.5 * k2 * (sy * jc - 2.0 * cy * js) + 0.5 * hx * sx * cy);
    TT(2, 1, 2) = 2.0 * (0.5 * k2 * (cy * js - 2.0 * k1 * sy * jd) + 0.5 * hx * k1 * dx * sy);
    TT(2, 1, 3) = 2.0 * (0.5 * k2 * (sy * js - 2.0 * cy * jd) + 0.5 * hx * dx * cy);
    TT(2, 2, 5) = 2.0 * (0.5 * hx / beta * k2 * (cy * jd - 2.0 * k1 * sy * jf) + 0.5 * hx2 / beta * k1 * j1 * sy -
                         0.25 / beta * k1 * L * sy);
    TT(2, 3, 5) = 2.0 * (0.5 * hx / beta * k2 * (sy * jd - 2.0 * cy * jf) + 0.5 * hx2 / beta * j1 * cy -
                         0.25 / beta * (sy + L * cy));
    TT(3, 0, 2) = 2.0 * (0.5 * k1 * k2 * (2.0 * cy * js - sy * jc) + 0.5 * (k2 + hx * k1) * sx * cy);
    TT(3, 0, 3) = 2.0 * (0.5 * k2 * (2.0 * k1 * sy * js - cy * jc) + 0.5 * (k2 + hx * k1) * sx * sy);
    TT(3, 1, 2) = 2.0 * (0.5 * k1 * k2 * (2.0 * cy * jd - sy * js) + 0.5 * (k2 + hx * k1) * dx * cy);
    TT(3, 1, 3) = 2.0 * (0.5 * k2 * (2.0 * k1 * sy * jd - cy * js) + 0.5 * (k2 + hx * k1) * dx * sy);
    TT(3, 2, 5) = 2.0 * (0.5 * hx / beta * k1 * k2 * (2.0 * cy * jf - sy * jd) + 0.5 * hx / beta * (k2 + hx * k1) * j1 * cy +
                         0.25 / beta * k1 * (sy - L * cy));
    TT(3, 3, 5) = 2.0 * (0.5 * hx / beta * k2 * (2.0 * k1 * sy * jf - cy * jd) + 0.5 * hx / beta * (k2 + hx * k1) * j1 * sy -
                         0.25 / beta * k1 * L * sy);
    TT(4, 0, 0) = -(hx / 12.0 / beta * khk * (sx * dx + 3.0 * j1) - 0.25 / beta * k1 * (L - sx * cx));
    TT(4, 0, 1) = -2.0 * (hx / 12.0 / beta * khk * dx2 + 0.25 / beta * k1 * sx * sx);
    TT(4, 1, 1) = -(hx / 6.0 / beta * khk * j2 - 0.5 / beta * sx - 0.25 / beta * k1 * (j1 - sx * dx));
    TT(4, 0, 5) = -2.0 * (hx2 / 12.0 / b2 * khk * (3.0 * dx * j1 - 4.0 * j2) + 0.25 * hx / b2 * k1 * j1 * (1.0 + cx) +
                          0.5 * hx / b2 * ig2 * sx);
    TT(4, 1, 5) = -2.0 * (hx2 / 12.0 / b2 * khk * (dx * dx2 - 2.0 * sx * j2) + 0.25 * hx / b2 * k1 * sx * j1 +
                          0.5 * hx / b2 * ig2 * dx);
    TT(4, 5, 5) = -(hx3 / 6.0 / b3 * khk * (3.0 * j3 - 2.0 * dx * j2) +
                    hx2 / 6.0 / b3 * k1 * (sx * dx2 - j2 * (1.0 + 2.0 * cx)) + 1.5 / b3 * ig2 * (hx2 * j1 - L));
    TT(4, 2, 2) = -(-hx / beta * k1 * k2 * jf - 0.5 * hx / beta * (k2 + hx * k1) * j1 + 0.25 / beta * k1 * (L - cy * sy));
    TT(4, 2, 3) = -2.0 * (-0.5 * hx / beta * k2 * jd - 0.25 / beta * k1 * sy * sy);
    TT(4, 3, 3) = -(-hx / beta * k2 * jf + 0.5 * hx2 / beta * j1 - 0.25 / beta * (L + cy * sy));
}
#undef TT

// One workgroup per batch row: lane 0 evaluates the closed forms into LDS, then 343 lanes carry out
// T'_inm = sum_jkl X_ij T_jkl E_kn E_lm (quadrupole.py:140-144) as three 7-term contractions.
constexpr int kTBlock = 384;

template <typename S>
__device__ __forceinline__ S s_zero() { return cst<S>(0.0); }

template <typename S>
__device__ void ttensor_contract(S* Ts, S* As, const S* X, const S* En) {
    const int id = threadIdx.x;
    const int i = id / 49, n = (id / 7) % 7, m = id % 7;
    __syncthreads();
    if (id < 343) {  // As[j=i][k=n][m] = sum_l Ts[j][k][l] En[l][m]
        S acc = s_zero<S>();
        for (int l = 0; l < 7; ++l) acc = acc + Ts[(i * 7 + n) * 7 + l] * En[l * 7 + m];
        As[id] = acc;
    }
    __syncthreads();
    if (id < 343) {  // Ts[j=i][n][m] = sum_k As[j][k][m] En[k][n]
        S acc = s_zero<S>();
        for (int k = 0; k < 7; ++k) acc = acc + As[(i * 7 + k) * 7 + m] * En[k * 7 + n];
        Ts[id] = acc;
    }
    __syncthreads();
    if (id < 343) {  // As[i][n][m] = sum_j X[i][j] Ts[j][n][m]
        S acc = s_zero<S>();
        for (int j = 0; j < 7; ++j) acc = acc + X[i * 7 + j] * Ts[(j * 7 + n) * 7 + m];
        As[id] = acc;
    }
    __syncthreads();
    if (id < 343) Ts[id] = As[id];
    __syncthreads();
}

// closed forms of one batch row (a single lane): the bare tensor + first-order map in Ts, the dressing matrices in
// X1/E1 (misalignment or dipole faces) and X2/E2 (dipole tilt)
template <typename S>
__device__ void ttensor_fill(int kind, const S* p, S en, double mass, S* Ts, S* X1, S* E1, S* X2, S* E2) {
    const S zero = s_zero<S>();
    Mat7<S> R;
    if (kind == CHX_T_DRIFT) {
        base_ttensor<S>(p[0], zero, zero, zero, en, mass, Ts);
        drift_map<S>(p[0], en, mass, R);
    } else if (kind == CHX_T_QUADRUPOLE) {
        base_ttensor<S>(p[0], p[1], zero, zero, en, mass, Ts);
        base_rmatrix<S>(p[0], p[1], zero, en, mass, R);
    } else if (kind == CHX_T_SEXTUPOLE) {
        base_ttensor<S>(p[0], zero, p[1], zero, en, mass, Ts);
        drift_map<S>(p[0], en, mass, R);
    } else if (kind == CHX_T_GENERAL) {  // [L, k1, k2, hx]: the bare tensor of track_methods.base_ttensor, no first-order block
        base_ttensor<S>(p[0], p[1], p[2], p[3], en, mass, Ts);
        return;
    } else {  // CHX_T_DIPOLE: [L, angle, k1, e1, e2, tilt, fint, fint_exit, gap]
        const S hx = p[1] / p[0];
        base_ttensor<S>(p[0], p[2], zero, hx, en, mass, Ts);
        base_rmatrix<S>(p[0], p[2], hx, en, mass, R);
    }
    // first-order map into T[:, 6, :] (drift.py:79-82)
    for (int i = 0; i < 7; ++i)
        for (int k = 0; k < 7; ++k) Ts[(i * 7 + 6) * 7 + k] = R(i, k);
    if (kind == CHX_T_QUADRUPOLE || kind == CHX_T_SEXTUPOLE) {
        // combined_rotation_misalignment_matrix (track_methods.py:345-382)
        const S tilt = p[2], mx = p[3], my = p[4];
        Mat7<S> entry, exitm;
        rotation_map<S>(tilt, entry);
        transpose7(entry, exitm);
        const S cs = m_cos(tilt), sn = m_sin(tilt);
        entry(0, 6) = -mx * cs - my * sn;
        entry(2, 6) = mx * sn - my * cs;
        exitm(0, 6) = mx;
        exitm(2, 6) = my;
        for (int k = 0; k < 49; ++k) { E1[k] = entry.m[k]; X1[k] = exitm.m[k]; }
    } else if (kind == CHX_T_DIPOLE) {
        const S hx = p[1] / p[0], e1 = p[3], e2 = p[4], tilt = p[5], fint = p[6], fint_exit = p[7], gap = p[8];
        Mat7<S> enter, exitm, rot, rotT;
        {
            const S sec = 1.0 / m_cos(e1), s1 = m_sin(e1);
            const S phi = fint * hx * gap * sec * (1.0 + s1 * s1);
            eye7(enter);
            enter(1, 0) = hx * m_tan(e1);
            enter(3, 2) = -hx * m_tan(e1 - phi);
        }
        {
            const S sec = 1.0 / m_cos(e2), s2 = m_sin(e2);
            const S phi = fint_exit * hx * gap * sec * (1.0 + s2 * s2);
            eye7(exitm);
            exitm(1, 0) = hx * m_tan(e2);
            exitm(3, 2) = -hx * m_tan(e2 - phi);
        }
        rotation_map<S>(tilt, rot);
        transpose7(rot, rotT);
        for (int k = 0; k < 49; ++k) { E1[k] = enter.m[k]; X1[k] = exitm.m[k]; E2[k] = rot.m[k]; X2[k] = rotT.m[k]; }
    }
}

template <typename T>
__global__ __launch_bounds__(kTBlock) void ttensor_kernel(int kind, const T* __restrict__ params,
                                                          const T* __restrict__ energy, double mass, int64_t Bp,
                                                          int64_t Be, int P, T* __restrict__ T_out) {
    __shared__ double Ts[343], As[343], X1[49], E1[49], X2[49], E2[49];
    const int64_t b = blockIdx.x;
    const int id = threadIdx.x;
    if (id < 343) Ts[id] = 0.0;
    __syncthreads();
    if (id == 0) {
        double p[CHX_MAX_PARAMS];
        for (int k = 0; k < P; ++k) p[k] = (double)params[(Bp == 1 ? 0 : b) * P + k];
        ttensor_fill<double>(kind, p, (double)energy[Be == 1 ? 0 : b], mass, Ts, X1, E1, X2, E2);
    }
    if (kind != CHX_T_DRIFT && kind != CHX_T_GENERAL) ttensor_contract<double>(Ts, As, X1, E1);
    if (kind == CHX_T_DIPOLE) ttensor_contract<double>(Ts, As, X2, E2);
    __syncthreads();
    if (id < 343) T_out[b * 343 + id] = (T)Ts[id];
}

// vector-Jacobian product of the builder: workgroup (b, k) seeds input k (the parameters, then the energy), pushes
// the tangent through the closed forms and the dressing, and contracts dT/dtheta_k with the incoming cotangent
template <typename T>
__global__ __launch_bounds__(kTBlock) void ttensor_vjp_kernel(int kind, const T* __restrict__ params,
                                                              const T* __restrict__ energy, double mass,
                                                              const T* __restrict__ dT, int64_t Bp, int64_t Be, int P,
                                                              T* __restrict__ dparams, T* __restrict__ denergy) {
    __shared__ Dual Ts[343], As[343], X1[49], E1[49], X2[49], E2[49];
    __shared__ double red[kTBlock / 64];
    const int64_t b = blockIdx.x;
    const int k = blockIdx.y;  // 0..P-1: parameter, P: energy
    const int id = threadIdx.x;
    if (id < 343) Ts[id] = mk(0.0, 0.0);
    __syncthreads();
    if (id == 0) {
        dual_arena_reset();
        Dual p[CHX_MAX_PARAMS];
        for (int j = 0; j < P; ++j) p[j] = mk((double)params[(Bp == 1 ? 0 : b) * P + j], j == k ? 1.0 : 0.0);
        const Dual en = mk((double)energy[Be == 1 ? 0 : b], k == P ? 1.0 : 0.0);
        ttensor_fill<Dual>(kind, p, en, mass, Ts, X1, E1, X2, E2);
    }
    if (kind != CHX_T_DRIFT && kind != CHX_T_GENERAL) ttensor_contract<Dual>(Ts, As, X1, E1);
    if (kind == CHX_T_DIPOLE) ttensor_contract<Dual>(Ts, As, X2, E2);
    __syncthreads();
    double acc = (id < 343) ? (double)dT[b * 343 + id] * Ts[id].d : 0.0;
    acc = chx_wave_sum(acc);
    if ((id & 63) == 0) red[id >> 6] = acc;
    __syncthreads();
    if (id == 0) {
        double tot = 0.0;
        for (int w = 0; w < kTBlock / 64; ++w) tot += red[w];
        if (k < P) dparams[b * P + k] = (T)tot;
        else denergy[b] = (T)tot;
    }
}

// ---- the reference's singularity-free special functions as element-wise ops (utils/autograd.py:4-74) -----------------
// The map builders above use these inline; chx_special evaluates one of them over an array together with its partial
// derivatives (what the reference's torch.autograd.Function pairs provide: forward :77-700, backward/jvp ibid.).
// The reference takes the closed form everywhere except exactly at the singular point; the series used here near it
// agree to rounding.
template <typename S>
__device__ S special_one(int kind, S x) {
    if (kind == CHX_SP_LOG1PDIV) return log1pdiv<S>(x);
    S C, Sn, G, F;
    if (kind == CHX_SP_SICOS1MDIV) {      // (1 - S C)(u) / u = 4 F(4u): sin(2a) = 2 sin a cos a
        sinc_family<S>(4.0 * x, C, Sn, G, F);
        return 4.0 * F;
    }
    sinc_family<S>(x, C, Sn, G, F);
    if (kind == CHX_SP_SI1MDIV) return F;
    if (kind == CHX_SP_SIPSICOS3MDIV) return ttensor_E<S>(x, C, Sn);
    // CHX_SP_SICOSKUDDELMUDDEL15MDIV: the reference substitutes 1/56 at exactly 0 (utils/autograd.py:305-308); its
    // derivative there is not finite (:323-335); special_kernel writes NaN for it
    if (val(x) == 0.0) return cst<S>(1.0 / 56.0);
    return ttensor_H<S>(x, C, Sn);
}

// a == b limits of the symmetric divided differences (utils/autograd.py:392,470,588): -C'(a), -S'(a), -(S^2)'(a)
template <typename S>
__device__ S special_diag(int kind, S a) {
    S C, Sn, G, F;
    sinc_family<S>(a, C, Sn, G, F);
    if (kind == CHX_SP_COSSQRTMCOSDIVDIFF) return 0.5 * Sn;
    if (kind == CHX_SP_SIMSIDIVDIFF) return 0.5 * (G - F);   // (S - C) / (2a)
    return Sn * (G - F);                                      // (1 - C^2 - a S C) / a^2
}

template <typename S>
__device__ S special_two(int kind, S a, S b) {
    if (kind == CHX_SP_SQRTA2MINUSBDIVA) {
        if (val(b) != 0.0) return (m_sqrt(a * a + b) - a) / b;
        // b == 0: value 1/(2a), partials -1/(2a^2) and -1/(8a^3) (utils/autograd.py:672-700)
        const S ia = 1.0 / a;
        return 0.5 * ia - 0.125 * ia * ia * ia * b;
    }
    S Ca, Sa, Ga, Fa, Cb, Sb, Gb, Fb;
    sinc_family<S>(a, Ca, Sa, Ga, Fa);
    sinc_family<S>(b, Cb, Sb, Gb, Fb);
    if (kind == CHX_SP_COSSQRTMCOSDIVDIFF) return (Cb - Ca) / (a - b);
    if (kind == CHX_SP_SIMSIDIVDIFF) return (Sa - Sb) / (b - a);
    return (Sb * Sb - Sa * Sa) / (a - b);
}

template <typename T>
__global__ __launch_bounds__(CHX_BLOCK) void special_kernel(int kind, const T* __restrict__ a, const T* __restrict__ b, int64_t n,
                                                           T* __restrict__ out, T* __restrict__ da, T* __restrict__ db) {
    for (int64_t i = (int64_t)blockIdx.x * CHX_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * CHX_BLOCK) {
        const double av = (double)a[i];
        if (kind < CHX_SP_COSSQRTMCOSDIVDIFF) {
            const Dual r = special_one<Dual>(kind, mk(av, 1.0));
            out[i] = (T)r.v;
            if (da) da[i] = (kind == CHX_SP_SICOSKUDDELMUDDEL15MDIV && av == 0.0) ? (T)NAN : (T)r.d;
            continue;
        }
        const double bv = (double)b[i];
        if (kind != CHX_SP_SQRTA2MINUSBDIVA && av == bv) {
            const Dual r = special_diag<Dual>(kind, mk(av, 1.0));   // symmetric in (a, b): each partial is half of d/dt f(t, t)
            out[i] = (T)r.v;
            if (da) da[i] = (T)(0.5 * r.d);
            if (db) db[i] = (T)(0.5 * r.d);
            continue;
        }
        const Dual ra = special_two<Dual>(kind, mk(av, 1.0), mk(bv, 0.0));
        out[i] = (T)ra.v;
        if (da) da[i] = (T)ra.d;
        if (db) db[i] = (T)special_two<Dual>(kind, mk(av, 0.0), mk(bv, 1.0)).d;
    }
}

}  // namespace

extern "C" int chx_special(int kind, const void* a, const void* b, int64_t n, int dtype, void* out, void* da, void* db,
                           void* stream) {
    if (kind < 0 || kind > CHX_SP_SQRTA2MINUSBDIVA || !a || !out || n < 0) return CHX_ERR_INVALID_ARG;
    if (kind >= CHX_SP_COSSQRTMCOSDIVDIFF && !b) return CHX_ERR_INVALID_ARG;
    if (dtype != CHX_F32 && dtype != CHX_F64) return CHX_ERR_DTYPE;
    if (n == 0) return CHX_OK;
    const int grid = chx_grid_for(n, CHX_BLOCK, 4096);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == CHX_F32)
        hipLaunchKernelGGL(special_kernel<float>, dim3(grid), dim3(CHX_BLOCK), 0, s, kind, (const float*)a, (const float*)b, n,
                           (float*)out, (float*)da, (float*)db);
    else
        hipLaunchKernelGGL(special_kernel<double>, dim3(grid), dim3(CHX_BLOCK), 0, s, kind, (const double*)a, (const double*)b,
                           n, (double*)out, (double*)da, (double*)db);
    CHX_CHECK_LAUNCH();
    return CHX_OK;
}

extern "C" int chx_t_num_params(int kind) {
    switch (kind) {
        case CHX_T_DRIFT: return 1;
        case CHX_T_QUADRUPOLE: return 5;
        case CHX_T_DIPOLE: return 9;
        case CHX_T_SEXTUPOLE: return 5;
        case CHX_T_GENERAL: return 4;
    }
    return -1;
}

extern "C" int chx_build_ttensor(int kind, const void* params, const void* energy, double mass_eV, int64_t B,
                                 int64_t Bp, int64_t Be, int dtype, void* T_out, void* stream) {
    const int P = chx_t_num_params(kind);
    if (P < 0 || !params || !energy || !T_out || B < 1 || B > 0x7fffffffLL) return CHX_ERR_INVALID_ARG;
    if (!chx_bcast_ok(Bp, B) || !chx_bcast_ok(Be, B)) return CHX_ERR_INVALID_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == CHX_F32)
        hipLaunchKernelGGL(ttensor_kernel<float>, dim3((unsigned)B), dim3(kTBlock), 0, s, kind, (const float*)params,
                           (const float*)energy, mass_eV, Bp, Be, P, (float*)T_out);
    else if (dtype == CHX_F64)
        hipLaunchKernelGGL(ttensor_kernel<double>, dim3((unsigned)B), dim3(kTBlock), 0, s, kind, (const double*)params,
                           (const double*)energy, mass_eV, Bp, Be, P, (double*)T_out);
    else
        return CHX_ERR_DTYPE;
    CHX_CHECK_LAUNCH();
    return CHX_OK;
}

extern "C" int chx_build_ttensor_vjp(int kind, const void* params, const void* energy, double mass_eV,
                                     const void* dT, int64_t B, int64_t Bp, int64_t Be, int dtype, void* dparams,
                                     void* denergy, void* stream) {
    const int P = chx_t_num_params(kind);
    if (P < 0 || !params || !energy || !dT || !dparams || !denergy || B < 1 || B > 0x7fffffffLL) return CHX_ERR_INVALID_ARG;
    if (!chx_bcast_ok(Bp, B) || !chx_bcast_ok(Be, B)) return CHX_ERR_INVALID_ARG;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)B, (unsigned)(P + 1));
    if (dtype == CHX_F32)
        hipLaunchKernelGGL(ttensor_vjp_kernel<float>, grid, dim3(kTBlock), 0, s, kind, (const float*)params,
                           (const float*)energy, mass_eV, (const float*)dT, Bp, Be, P, (float*)dparams, (float*)denergy);
    else if (dtype == CHX_F64)
        hipLaunchKernelGGL(ttensor_vjp_kernel<double>, grid, dim3(kTBlock), 0, s, kind, (const double*)params,
                           (const double*)energy, mass_eV, (const double*)dT, Bp, Be, P, (double*)dparams,
                           (double*)denergy);
    else
        return CHX_ERR_DTYPE;
    CHX_CHECK_LAUNCH();
    return CHX_OK;
}

extern "C" int chx_kind_num_params(int kind) { return kind_num_params(kind); }
extern "C" int chx_abi_version(void) { return CHX_ABI_VERSION; }
extern "C" const char* chx_status_string(int status) {
    switch (status) {
        case CHX_OK: return "ok";
        case CHX_ERR_INVALID_ARG: return "invalid argument";
        case CHX_ERR_DTYPE: return "unsupported dtype";
        case CHX_ERR_MISALIGNED: return "misaligned buffer";
        case CHX_ERR_LAUNCH: return "kernel launch failed";
        case CHX_ERR_WORKSPACE: return "workspace missing or too small";
        case CHX_ERR_NO_DEVICE: return "no HIP device";
        default: return "unknown status";
    }
}

extern "C" int chx_build_rmatrix(int kind, const void* params, const void* energy, double mass_eV,
                                 double n_charges, int64_t B, int64_t Bp, int64_t Be, int dtype,
                                 void* R_out, void* stream) {
    const int P = kind_num_params(kind);
    if (P < 0 || B < 1 || !energy || !R_out || (P > 0 && !params)) return CHX_ERR_INVALID_ARG;
    if (!chx_bcast_ok(Bp, B) || !chx_bcast_ok(Be, B)) return CHX_ERR_INVALID_ARG;
    const int grid = (int)((B + 63) / 64);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == CHX_F32)
        hipLaunchKernelGGL(build_kernel<float>, dim3(grid), dim3(64), 0, s, kind, (const float*)params,
                           (const float*)energy, mass_eV, n_charges, B, Bp, Be, P, (float*)R_out);
    else if (dtype == CHX_F64)
        hipLaunchKernelGGL(build_kernel<double>, dim3(grid), dim3(64), 0, s, kind, (const double*)params,
                           (const double*)energy, mass_eV, n_charges, B, Bp, Be, P, (double*)R_out);
    else
        return CHX_ERR_DTYPE;
    CHX_CHECK_LAUNCH();
    return CHX_OK;
}

extern "C" int chx_build_rmatrix_vjp(int kind, const void* params, const void* energy, double mass_eV,
                                     double n_charges, const void* dR, int64_t B, int64_t Bp,
                                     int64_t Be, int dtype, void* dparams, void* denergy,
                                     void* stream) {
    const int P = kind_num_params(kind);
    if (P < 0 || B < 1 || !energy || !dR || !denergy || (P > 0 && (!params || !dparams)))
        return CHX_ERR_INVALID_ARG;
    if (!chx_bcast_ok(Bp, B) || !chx_bcast_ok(Be, B)) return CHX_ERR_INVALID_ARG;
    const int64_t work = B * (P + 1);
    if (work > 0x7fffffffLL) return CHX_ERR_INVALID_ARG;
    const int grid = (int)work;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == CHX_F32)
        hipLaunchKernelGGL(build_vjp_kernel<float>, dim3(grid), dim3(64), 0, s, kind,
                           (const float*)params, (const float*)energy, mass_eV, n_charges,
                           (const float*)dR, B, Bp, Be, P, (float*)dparams, (float*)denergy);
    else if (dtype == CHX_F64)
        hipLaunchKernelGGL(build_vjp_kernel<double>, dim3(grid), dim3(64), 0, s, kind,
                           (const double*)params, (const double*)energy, mass_eV, n_charges,
                           (const double*)dR, B, Bp, Be, P, (double*)dparams, (double*)denergy);
    else
        return CHX_ERR_DTYPE;
    CHX_CHECK_LAUNCH();
    return CHX_OK;
}

extern "C" int chx_cavity_coeffs(const void* params, const void* energy, double mass_eV,
                                 double n_charges, int64_t B, int64_t Bp, int64_t Be, int dtype,
                                 double* coeffs, void* energy_out, void* stream) {
    if (!params || !energy || !coeffs || !energy_out || B < 1) return CHX_ERR_INVALID_ARG;
    if (!chx_bcast_ok(Bp, B) || !chx_bcast_ok(Be, B)) return CHX_ERR_INVALID_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == CHX_F32)
        hipLaunchKernelGGL(cavity_coeff_kernel<float>, dim3(1), dim3(CHX_BLOCK), 0, s,
                           (const float*)params, (const float*)energy, mass_eV, n_charges, B, Bp, Be,
                           coeffs, (float*)energy_out);
    else if (dtype == CHX_F64)
        hipLaunchKernelGGL(cavity_coeff_kernel<double>, dim3(1), dim3(CHX_BLOCK), 0, s,
                           (const double*)params, (const double*)energy, mass_eV, n_charges, B, Bp,
                           Be, coeffs, (double*)energy_out);
    else
        return CHX_ERR_DTYPE;
    CHX_CHECK_LAUNCH();
    return CHX_OK;
}

extern "C" int chx_cavity_prepare_scalars(const void* const* param_ptrs, const void* energy, int kind, double mass_eV,
                                          double n_charges, int dtype, void* R_out, double* coeffs, void* energy_out,
                                          const void* s_in, void* s_out, void* stream) {
    if (!param_ptrs || !energy || !R_out || !coeffs || !energy_out || ((s_in == nullptr) != (s_out == nullptr)))
        return CHX_ERR_INVALID_ARG;
    if (kind != CHX_CAVITY_SW && kind != CHX_CAVITY_TW) return CHX_ERR_INVALID_ARG;
    for (int k = 0; k < 4; ++k)
        if (!param_ptrs[k]) return CHX_ERR_INVALID_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == CHX_F32)
        hipLaunchKernelGGL(cavity_prepare_scalars_kernel<float>, dim3(1), dim3(64), 0, s, (const float*)param_ptrs[0],
                           (const float*)param_ptrs[1], (const float*)param_ptrs[2], (const float*)param_ptrs[3],
                           (const float*)energy, kind, mass_eV, n_charges, (float*)R_out, coeffs, (float*)energy_out,
                           (const float*)s_in, (float*)s_out);
    else if (dtype == CHX_F64)
        hipLaunchKernelGGL(cavity_prepare_scalars_kernel<double>, dim3(1), dim3(64), 0, s, (const double*)param_ptrs[0],
                           (const double*)param_ptrs[1], (const double*)param_ptrs[2], (const double*)param_ptrs[3],
                           (const double*)energy, kind, mass_eV, n_charges, (double*)R_out, coeffs, (double*)energy_out,
                           (const double*)s_in, (double*)s_out);
    else
        return CHX_ERR_DTYPE;
    CHX_CHECK_LAUNCH();
    return CHX_OK;
}

// ---- maps of a whole run of elements with SCALAR parameters, one launch per kBuildChunk elements ----------------------
// The usual control loop (change a few magnet settings, track, read a screen) rebuilds the maps of every element whose
// setting changed. Through chx_build_rmatrix that is one packed-parameter tensor (a torch.stack), one output allocation
// and one launch PER ELEMENT; here the host only collects device pointers to the scalars where they already live, and one
// thread per element evaluates its builder. Kinds and pointers travel by value in the kernel arguments.
constexpr int kBuildChunk = 40;  // 40 x (9 pointers + 1 kind byte) = 2.9 KiB of kernel arguments
struct BuildScalarsArgs {
    const void* par[kBuildChunk][CHX_MAX_PARAMS];
    uint8_t kind[kBuildChunk];
    uint16_t need[kBuildChunk];   // backward only: bit k = slot k of element e is wanted (bit CHX_MAX_PARAMS: the energy)
};

template <typename T>
__global__ __launch_bounds__(64) void build_scalars_kernel(BuildScalarsArgs args, int n, const T* __restrict__ energy,
                                                           double mass, double nq, T* __restrict__ R_out) {
    const int e = threadIdx.x;
    if (e >= n) return;
    const int kind = args.kind[e];
    const int P = kind_num_params(kind);
    double p[CHX_MAX_PARAMS];
    for (int k = 0; k < P; ++k) p[k] = (double)*(const T*)args.par[e][k];
    Mat7<double> R;
    build_kind<double>(kind, p, (double)energy[0], mass, nq, R);
    for (int q = 0; q < 49; ++q) R_out[e * 49 + q] = (T)R.m[q];
}

extern "C" int chx_build_rmatrix_scalars(const int32_t* kinds, const void* const* param_ptrs, int64_t E, const void* energy,
                                         double mass_eV, double n_charges, int dtype, void* R_out, void* stream) {
    if (!kinds || !param_ptrs || !energy || !R_out || E < 1) return CHX_ERR_INVALID_ARG;
    if (dtype != CHX_F32 && dtype != CHX_F64) return CHX_ERR_DTYPE;
    hipStream_t s = (hipStream_t)stream;
    const size_t esz = dtype == CHX_F32 ? 4 : 8;
    for (int64_t done = 0; done < E; done += kBuildChunk) {
        BuildScalarsArgs a;
        const int n = (int)((E - done < kBuildChunk) ? (E - done) : kBuildChunk);
        for (int e = 0; e < n; ++e) {
            const int kind = kinds[done + e];
            const int P = kind_num_params(kind);
            if (P < 0) return CHX_ERR_INVALID_ARG;
            a.kind[e] = (uint8_t)kind;
            for (int k = 0; k < CHX_MAX_PARAMS; ++k) {
                a.par[e][k] = k < P ? param_ptrs[(done + e) * CHX_MAX_PARAMS + k] : nullptr;
                if (k < P && !a.par[e][k]) return CHX_ERR_INVALID_ARG;
            }
        }
        for (int e = n; e < kBuildChunk; ++e) {
            a.kind[e] = 0;
            for (int k = 0; k < CHX_MAX_PARAMS; ++k) a.par[e][k] = nullptr;
        }
        char* out = (char*)R_out + (size_t)done * 49 * esz;
        if (dtype == CHX_F32)
            hipLaunchKernelGGL(build_scalars_kernel<float>, dim3(1), dim3(64), 0, s, a, n, (const float*)energy, mass_eV,
                               n_charges, (float*)out);
        else
            hipLaunchKernelGGL(build_scalars_kernel<double>, dim3(1), dim3(64), 0, s, a, n, (const double*)energy, mass_eV,
                               n_charges, (double*)out);
        CHX_CHECK_LAUNCH();
    }
    return CHX_OK;
}

// ---- backward of chx_build_rmatrix_scalars + chx_compose_maps (a run of scalar-parameter elements, gradients wanted) -----
// T = R_{E-1} ... R_0. With P_e = R_{e-1} ... R_0 and G_e = (R_{E-1} ... R_{e+1})^T dT:  dL/dR_e = G_e P_e^T,
// G_{e-1} = R_e^T G_e. One wave: a forward sweep leaves the prefixes in `ws`, the backward sweep replaces them by dL/dR_e
// (fp64 throughout). A second launch contracts dL/dR_e with the builders' derivatives (dual numbers, one thread per
// (element, input) like build_vjp_kernel): autograd through a whole run costs two launches instead of ~3 nodes per element.
template <typename T>
__global__ __launch_bounds__(64) void compose_scalars_bwd_kernel(const T* __restrict__ maps, int E, const T* __restrict__ dT,
                                                                 double* __restrict__ ws) {
    __shared__ double P[49], R[49], G[49];
    const int lane = threadIdx.x, i = lane / 7, j = lane - 7 * i;
    const bool on = lane < 49;
    if (on) P[lane] = (i == j) ? 1.0 : 0.0;
    __syncthreads();
    for (int e = 0; e < E; ++e) {
        if (on) {
            ws[e * 49 + lane] = P[lane];
            R[lane] = (double)maps[e * 49 + lane];
        }
        __syncthreads();
        double acc = 0.0;
        if (on) {
            acc = R[i * 7] * P[j];
            for (int k = 1; k < 7; ++k) acc = fma(R[i * 7 + k], P[k * 7 + j], acc);
        }
        __syncthreads();
        if (on) P[lane] = acc;
        __syncthreads();
    }
    if (on) G[lane] = (double)dT[lane];
    __syncthreads();
    for (int e = E - 1; e >= 0; --e) {
        if (on) {
            P[lane] = ws[e * 49 + lane];
            R[lane] = (double)maps[e * 49 + lane];
        }
        __syncthreads();
        double dr = 0.0, gn = 0.0;
        if (on) {
            for (int k = 0; k < 7; ++k) {
                dr = fma(G[i * 7 + k], P[j * 7 + k], dr);
                gn = fma(R[k * 7 + i], G[k * 7 + j], gn);
            }
            ws[e * 49 + lane] = dr;
        }
        __syncthreads();
        if (on) G[lane] = gn;
        __syncthreads();
    }
}

// The same sweep for VECTORISED maps (chx_compose_maps with gradients: segment.py:534-543 under autograd): one wave per batch
// row, element maps through the by-value pointer table of compose_kernel, dM[e][b] = G_e P_e^T in T. ws: [B][E][49] doubles.
template <typename T>
__global__ __launch_bounds__(64) void compose_vjp_kernel(ComposeArgs args, int E, const T* __restrict__ dT, double* __restrict__ ws,
                                                         T* __restrict__ dM /*[E][B][49]*/) {
    __shared__ double P[49], R[49], G[49];
    const int64_t b = blockIdx.x, B = gridDim.x;
    const int lane = threadIdx.x, i = lane / 7, j = lane - 7 * i;
    const bool on = lane < 49;
    double* wb = ws + b * (int64_t)E * 49;
    auto map_of = [&](int e) { return (const T*)args.ptr[e] + (args.bcast[e] ? 0 : b * 49); };
    if (on) P[lane] = (i == j) ? 1.0 : 0.0;
    __syncthreads();
    for (int e = 0; e < E; ++e) {
        if (on) {
            wb[e * 49 + lane] = P[lane];
            R[lane] = (double)map_of(e)[lane];
        }
        __syncthreads();
        double acc = 0.0;
        if (on) {
            acc = R[i * 7] * P[j];
            for (int k = 1; k < 7; ++k) acc = fma(R[i * 7 + k], P[k * 7 + j], acc);
        }
        __syncthreads();
        if (on) P[lane] = acc;
        __syncthreads();
    }
    if (on) G[lane] = (double)dT[b * 49 + lane];
    __syncthreads();
    for (int e = E - 1; e >= 0; --e) {
        if (on) {
            P[lane] = wb[e * 49 + lane];
            R[lane] = (double)map_of(e)[lane];
        }
        __syncthreads();
        double dr = 0.0, gn = 0.0;
        if (on) {
            for (int k = 0; k < 7; ++k) {
                dr = fma(G[i * 7 + k], P[j * 7 + k], dr);
                gn = fma(R[k * 7 + i], G[k * 7 + j], gn);
            }
            dM[((int64_t)e * B + b) * 49 + lane] = (T)dr;
        }
        __syncthreads();
        if (on) G[lane] = gn;
        __syncthreads();
    }
}

// one WAVE per (element, slot k) (see Mat7<Dual>): slots 0 .. P-1 are the element's parameters, slot CHX_MAX_PARAMS the energy,
// the rest are written as zeros. out[e][CHX_MAX_PARAMS + 1]
// dL/dR_e of compose_scalars_bwd_kernel for ONE element, by the calling wave alone (the same products in the same order: prefix
// P_e = R_{e-1} ... R_0, G_e = (R_{E-1} ... R_{e+1})^T dT, dL/dR_e = G_e P_e^T): E - 1 cooperative 7x7 products instead of a
// launch of its own in front of the VJP kernel — for the short runs of an optimisation loop (E <= kRunVjpFuseE) the launch costs
// more than the sweep. dRl: 49 doubles of LDS; tmp: 3 * 49 doubles of LDS.
constexpr int kRunVjpFuseE = 16;

template <typename T>
__device__ __forceinline__ void wave_element_cotangent(const T* __restrict__ maps, int E, int e, const T* __restrict__ dT, double* tmp,
                                                       double* dRl) {
    double* P = tmp;
    double* R = tmp + 49;
    double* G = tmp + 98;
    const int lane = threadIdx.x & 63, i = lane / 7, j = lane - 7 * i;
    const bool on = lane < 49;
    if (on) P[lane] = (i == j) ? 1.0 : 0.0;
    chx_wave_sync();
    for (int q = 0; q < e; ++q) {
        if (on) R[lane] = (double)maps[q * 49 + lane];
        chx_wave_sync();
        double acc = 0.0;
        if (on) {
            acc = R[i * 7] * P[j];
            for (int k = 1; k < 7; ++k) acc = fma(R[i * 7 + k], P[k * 7 + j], acc);
        }
        chx_wave_sync();
        if (on) P[lane] = acc;
        chx_wave_sync();
    }
    if (on) G[lane] = (double)dT[lane];
    chx_wave_sync();
    for (int q = E - 1; q > e; --q) {
        if (on) R[lane] = (double)maps[q * 49 + lane];
        chx_wave_sync();
        double gn = 0.0;
        if (on)
            for (int k = 0; k < 7; ++k) gn = fma(R[k * 7 + i], G[k * 7 + j], gn);
        chx_wave_sync();
        if (on) G[lane] = gn;
        chx_wave_sync();
    }
    if (on) {
        double dr = 0.0;
        for (int k = 0; k < 7; ++k) dr = fma(G[i * 7 + k], P[j * 7 + k], dr);
        dRl[lane] = dr;
    }
    chx_wave_sync();
}

// dR: dL/dR_e of every element ([n][49], from compose_scalars_bwd_kernel), or NULL: formed here from maps / dT (fused form)
// entry.grad set (chx_run_vjp_entry): dT is not given but formed by every wave itself — the cotangent the composed map C receives
// from ONE property of the beam y = C x (its moments mom_y, the incoming beam's mom_x; chx_moment_entry_mapped_bwd's arithmetic,
// rounded to T like the tensor that launch would have written): the backward pass of d sigma_x(screen) / d k1 in one launch.
struct VjpEntry {
    const void* grad;        // one value of T, or NULL
    const double* mom_y;
    const double* mom_x;
    const void* C;           // [49] of T
    int index, take_sqrt;
};

template <typename T>
__global__ __launch_bounds__(64) void build_scalars_vjp_kernel(BuildScalarsArgs args, int n, const T* __restrict__ energy,
                                                               double mass, double nq, const double* __restrict__ dR,
                                                               const T* __restrict__ maps, const T* __restrict__ dT,
                                                               T* __restrict__ out, VjpEntry entry) {
    __shared__ double cot[4 * 49];
    __shared__ double entry_lds[3 * 36 + CHX_MOM_NOUT];
    __shared__ T dT_own[49];
    const int idx = blockIdx.x;
    const int e = idx / (CHX_MAX_PARAMS + 1), k = idx - e * (CHX_MAX_PARAMS + 1);
    if (e >= n) return;
    const int lane = threadIdx.x;
    const int kind = args.kind[e];
    const int P = kind_num_params(kind);
    const bool is_energy = k == CHX_MAX_PARAMS;
    T* o = out + e * (CHX_MAX_PARAMS + 1);
    if ((!is_energy && k >= P) || !((args.need[e] >> k) & 1)) {
        if (lane == 0) o[k] = (T)0;
        return;
    }
    dual_arena_reset();
    Dual p[CHX_MAX_PARAMS];
    for (int q = 0; q < P; ++q) p[q] = mk((double)*(const T*)args.par[e][q], (!is_energy && q == k) ? 1.0 : 0.0);
    const Dual en = mk((double)energy[0], is_energy ? 1.0 : 0.0);
    Mat7<Dual> R;
    build_kind<Dual>(kind, p, en, mass, nq, R);
    chx_wave_sync();
    const double* dRe = dR ? dR + e * 49 : cot + 147;
    if (!dR && entry.grad) {
        moment_entry_gradient((double)*(const T*)entry.grad, entry.mom_y, entry.index, entry.take_sqrt, entry_lds + 108);
        mapped_bwd_row_wave<T, T>(entry_lds + 108, (const T*)entry.C, entry.mom_x, entry_lds, dT_own);
        chx_wave_sync();
        dT = dT_own;
    }
    if (!dR) wave_element_cotangent<T>(maps, n, e, dT, cot, cot + 147);
    double acc = lane < 49 ? dRe[lane] * R.m[lane].d : 0.0;
    acc = chx_wave_sum(acc);
    if (lane == 0) o[k] = (T)acc;
}

extern "C" size_t chx_run_vjp_workspace_bytes(int64_t E) { return E < 1 ? 0 : (size_t)E * 49 * sizeof(double); }

extern "C" int chx_run_vjp(const int32_t* kinds, const void* const* param_ptrs, int64_t E, const void* energy, double mass_eV,
                           double n_charges, int dtype, const void* maps, const void* dT, void* dinputs, void* workspace,
                           size_t workspace_bytes, void* stream) {
    return chx_run_vjp_masked(kinds, param_ptrs, E, energy, mass_eV, n_charges, dtype, maps, dT, nullptr, dinputs, workspace,
                              workspace_bytes, stream);
}

static int run_vjp_launch(const int32_t* kinds, const void* const* param_ptrs, int64_t E, const void* energy, double mass_eV,
                          double n_charges, int dtype, const void* maps, const void* dT, const uint16_t* need, void* dinputs,
                          void* workspace, size_t workspace_bytes, void* stream, const VjpEntry& entry);

extern "C" int chx_run_vjp_masked(const int32_t* kinds, const void* const* param_ptrs, int64_t E, const void* energy,
                                  double mass_eV, double n_charges, int dtype, const void* maps, const void* dT,
                                  const uint16_t* need, void* dinputs, void* workspace, size_t workspace_bytes, void* stream) {
    if (!dT) return CHX_ERR_INVALID_ARG;
    return run_vjp_launch(kinds, param_ptrs, E, energy, mass_eV, n_charges, dtype, maps, dT, need, dinputs, workspace, workspace_bytes,
                          stream, VjpEntry{});
}

extern "C" int chx_moment_entry_mapped_bwd(const void* grad, const double* mom_y, int index, int take_sqrt, const void* R,
                                           const double* mom_x, int64_t B, int64_t BR, int64_t Bm, int dtype, void* dR,
                                           int dR_is_double, void* stream);

extern "C" int chx_run_vjp_entry(const int32_t* kinds, const void* const* param_ptrs, int64_t E, const void* energy, double mass_eV,
                                 double n_charges, int dtype, const void* maps, const uint16_t* need, const void* grad,
                                 const double* mom_y, int index, int take_sqrt, const void* C, const double* mom_x, void* dinputs,
                                 void* workspace, size_t workspace_bytes, void* stream) {
    if (!grad || !mom_y || !mom_x || !C || index < 2 || index >= CHX_MOM_NOUT) return CHX_ERR_INVALID_ARG;
    if (E >= 1 && E <= kRunVjpFuseE && E <= kBuildChunk) {
        VjpEntry entry;
        entry.grad = grad;
        entry.mom_y = mom_y;
        entry.mom_x = mom_x;
        entry.C = C;
        entry.index = index;
        entry.take_sqrt = take_sqrt;
        return run_vjp_launch(kinds, param_ptrs, E, energy, mass_eV, n_charges, dtype, maps, nullptr, need, dinputs, workspace,
                              workspace_bytes, stream, entry);
    }
    // a long run: the cotangent as a launch of its own, at the head of the workspace's spare 49 values (chx_run_vjp_entry_workspace_bytes)
    if (dtype != CHX_F32 && dtype != CHX_F64) return CHX_ERR_DTYPE;
    const size_t base = chx_run_vjp_workspace_bytes(E);
    if (!workspace || E < 1 || workspace_bytes < base + 49 * sizeof(double)) return CHX_ERR_WORKSPACE;
    void* dT = (char*)workspace + base;
    const int rc = chx_moment_entry_mapped_bwd(grad, mom_y, index, take_sqrt, C, mom_x, 1, 1, 1, dtype, dT, 0, stream);
    if (rc != CHX_OK) return rc;
    return run_vjp_launch(kinds, param_ptrs, E, energy, mass_eV, n_charges, dtype, maps, dT, need, dinputs, workspace, base, stream,
                          VjpEntry{});
}

extern "C" size_t chx_run_vjp_entry_workspace_bytes(int64_t E) { return E < 1 ? 0 : chx_run_vjp_workspace_bytes(E) + 49 * sizeof(double); }

static int run_vjp_launch(const int32_t* kinds, const void* const* param_ptrs, int64_t E, const void* energy, double mass_eV,
                          double n_charges, int dtype, const void* maps, const void* dT, const uint16_t* need, void* dinputs,
                          void* workspace, size_t workspace_bytes, void* stream, const VjpEntry& entry) {
    if (!kinds || !param_ptrs || !energy || !maps || (!dT && !entry.grad) || !dinputs || E < 1 || E > 65535) return CHX_ERR_INVALID_ARG;
    if (dtype != CHX_F32 && dtype != CHX_F64) return CHX_ERR_DTYPE;
    if (!workspace || workspace_bytes < chx_run_vjp_workspace_bytes(E)) return CHX_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    double* ws = (double*)workspace;
    const bool fused = E <= kRunVjpFuseE && E <= kBuildChunk;      // every wave forms its element's cotangent itself
    if (!fused) {
        if (dtype == CHX_F32)
            hipLaunchKernelGGL(compose_scalars_bwd_kernel<float>, dim3(1), dim3(64), 0, s, (const float*)maps, (int)E, (const float*)dT,
                               ws);
        else
            hipLaunchKernelGGL(compose_scalars_bwd_kernel<double>, dim3(1), dim3(64), 0, s, (const double*)maps, (int)E,
                               (const double*)dT, ws);
        CHX_CHECK_LAUNCH();
    }
    const size_t esz = dtype == CHX_F32 ? 4 : 8;
    for (int64_t done = 0; done < E; done += kBuildChunk) {
        BuildScalarsArgs a;
        const int n = (int)((E - done < kBuildChunk) ? (E - done) : kBuildChunk);
        for (int e = 0; e < kBuildChunk; ++e) {
            const int kind = e < n ? kinds[done + e] : 0;
            const int P = kind_num_params(kind);
            if (P < 0) return CHX_ERR_INVALID_ARG;
            a.kind[e] = (uint8_t)kind;
            a.need[e] = (e < n) ? (need ? need[done + e] : (uint16_t)0xffff) : (uint16_t)0;
            for (int k = 0; k < CHX_MAX_PARAMS; ++k) {
                a.par[e][k] = (e < n && k < P) ? param_ptrs[(done + e) * CHX_MAX_PARAMS + k] : nullptr;
                if (e < n && k < P && !a.par[e][k]) return CHX_ERR_INVALID_ARG;
            }
        }
        char* out = (char*)dinputs + (size_t)done * (CHX_MAX_PARAMS + 1) * esz;
        const unsigned blocks = (unsigned)(n * (CHX_MAX_PARAMS + 1));
        if (dtype == CHX_F32)
            hipLaunchKernelGGL(build_scalars_vjp_kernel<float>, dim3(blocks), dim3(64), 0, s, a, n, (const float*)energy, mass_eV,
                               n_charges, fused ? (const double*)nullptr : ws + done * 49, (const float*)maps, (const float*)dT,
                               (float*)out, entry);
        else
            hipLaunchKernelGGL(build_scalars_vjp_kernel<double>, dim3(blocks), dim3(64), 0, s, a, n, (const double*)energy, mass_eV,
                               n_charges, fused ? (const double*)nullptr : ws + done * 49, (const double*)maps, (const double*)dT,
                               (double*)out, entry);
        CHX_CHECK_LAUNCH();
    }
    return CHX_OK;
}

// ---- prefix products of a run (Segment.get_beam_attrs_along_segment, segment.py:658-700): out[e] = M_e ... M_1 M_0 for
// every e, fp64 accumulation carried from element to element, each prefix rounded to T once. One wave per batch row.
template <typename T>
__global__ __launch_bounds__(64) void compose_prefix_kernel(const T* __restrict__ maps, int E, int64_t B, int64_t Bm,
                                                            T* __restrict__ out) {
    __shared__ double cur[49];
    __shared__ double nxt[49];
    const int64_t b = blockIdx.x;
    const int lane = threadIdx.x;
    const int i = lane / 7, j = lane - 7 * i;
    double acc = (lane < 49) ? ((i == j) ? 1.0 : 0.0) : 0.0;
    for (int e = 0; e < E; ++e) {
        const T* Me = maps + ((int64_t)e * Bm + (Bm == 1 ? 0 : b)) * 49;
        if (lane < 49) { cur[lane] = acc; nxt[lane] = (double)Me[lane]; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (lane < 49) {
            double s = nxt[i * 7] * cur[j];
            for (int k = 1; k < 7; ++k) s = fma(nxt[i * 7 + k], cur[k * 7 + j], s);
            acc = s;
            out[((int64_t)e * B + b) * 49 + lane] = (T)acc;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

extern "C" int chx_compose_prefix(const void* maps, int64_t E, int64_t B, int64_t Bm, int dtype, void* out, void* stream) {
    if (!maps || !out || E < 1 || B < 1 || B > 0x7fffffffLL || !chx_bcast_ok(Bm, B)) return CHX_ERR_INVALID_ARG;
    if (dtype != CHX_F32 && dtype != CHX_F64) return CHX_ERR_DTYPE;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == CHX_F32)
        hipLaunchKernelGGL(compose_prefix_kernel<float>, dim3((unsigned)B), dim3(64), 0, s, (const float*)maps, (int)E, B, Bm,
                           (float*)out);
    else
        hipLaunchKernelGGL(compose_prefix_kernel<double>, dim3((unsigned)B), dim3(64), 0, s, (const double*)maps, (int)E, B, Bm,
                           (double*)out);
    CHX_CHECK_LAUNCH();
    return CHX_OK;
}

// ---- persistent map of a run of scalar-parameter elements (Segment.track's steady state) ---------------------------
// A control loop tracks the same lattice again and again, usually with unchanged settings. Instead of the host proving
// that nothing changed (one Python attribute read per tensor of the run) the device does: ONE workgroup reads every
// parameter where it lives, compares it with the value the stored map was built from and returns at once when all are
// equal; otherwise it rebuilds the element maps (rounded to T like chx_build_rmatrix_scalars) and composes them with the
// association of chx_compose_maps — bit-identical to build + compose. The pointers travel by value in the kernel
// arguments (packed: only the parameters each kind has), the state lives in a caller-owned device buffer.
constexpr int kRunMaxE = 192;      // one compose launch (kComposeChunk)
constexpr int kRunMaxPtr = 400;    // 3.2 KB of the 4 KB kernel-argument space
struct RunArgs {
    const void* ptr[kRunMaxPtr];
    uint16_t off[kRunMaxE];        // first pointer of element e
    uint8_t kind[kRunMaxE];
};

template <typename T>
__global__ __launch_bounds__(CHX_BLOCK) void run_map_kernel(RunArgs a, int E, int nptr, const T* __restrict__ energy,
                                                           double mass, double nq, double* __restrict__ seen /*[nptr + 3]*/,
                                                           T* __restrict__ maps /*[E][49]*/, T* __restrict__ R /*[49]*/,
                                                           const T* __restrict__ s_in, T* __restrict__ s_out) {
    // 1. anything different from what R was built from? (NaN-initialised state: the first call is always dirty)
    // (seen == NULL: no stored state, the maps are always built — chx_run_build_compose)
    int dirty = seen ? 0 : 1;
    for (int q = threadIdx.x; seen && q < nptr + 3; q += CHX_BLOCK) {
        const double v = q < nptr ? (double)*(const T*)a.ptr[q] : (q == nptr ? (double)energy[0] : (q == nptr + 1 ? mass : nq));
        if (!(v == seen[q])) dirty = 1;
    }
    // 1b. path length behind the run, s + (((L_0 + L_1) + L_2) + ...) in T like the reference's `incoming.s + segment.length`
    // (segment.py:54-58, element.py:189): the first parameter of every kind is its length. Read from the settings on every
    // call — the host keeps no copy that an in-place edit of a length could leave stale. All lengths are fetched in parallel,
    // lane 0 adds them in element order.
    __shared__ T lens[kRunMaxE];
    if (s_out) {
        for (int e = threadIdx.x; e < E; e += CHX_BLOCK) lens[e] = *(const T*)a.ptr[a.off[e]];
    }
    dirty = __syncthreads_or(dirty);
    if (threadIdx.x == 0 && s_out) {
        T total = lens[0];
        for (int e = 1; e < E; ++e) total = total + lens[e];
        *s_out = *s_in + total;
    }
    if (!dirty) return;
    // 2. element maps, fp64 inside, rounded to T (chx_build_rmatrix_scalars)
    for (int e = threadIdx.x; e < E; e += CHX_BLOCK) {
        const int kind = a.kind[e];
        const int P = kind_num_params(kind);
        double p[CHX_MAX_PARAMS];
        for (int k = 0; k < P; ++k) p[k] = (double)*(const T*)a.ptr[a.off[e] + k];
        Mat7<double> M;
        build_kind<double>(kind, p, (double)energy[0], mass, nq, M);
        for (int q = 0; q < 49; ++q) maps[e * 49 + q] = (T)M.m[q];
    }
    for (int q = threadIdx.x; seen && q < nptr + 3; q += CHX_BLOCK)
        seen[q] = q < nptr ? (double)*(const T*)a.ptr[q] : (q == nptr ? (double)energy[0] : (q == nptr + 1 ? mass : nq));
    __syncthreads();
    // 3. R = M_{E-1} ... M_0 (chx_compose_maps)
    if (E == 1) {
        if (threadIdx.x < 49) R[threadIdx.x] = maps[threadIdx.x];
        return;
    }
    compose_block<T>([&](int e) { return (const T*)maps + e * 49; }, E, 0, R);
}

extern "C" size_t chx_run_state_bytes(int64_t E) {
    if (E < 1 || E > kRunMaxE) return 0;
    // seen[kRunMaxPtr + 3] doubles, maps[E][49] and R[49] as doubles (room for either dtype)
    return (size_t)(kRunMaxPtr + 4 + (E + 1) * 49) * sizeof(double);
}

static int run_args(const int32_t* kinds, const void* const* param_ptrs, int64_t E, RunArgs& a, int& nptr) {
    if (!kinds || !param_ptrs || E < 1 || E > kRunMaxE) return CHX_ERR_INVALID_ARG;
    nptr = 0;
    for (int e = 0; e < (int)E; ++e) {
        const int P = kind_num_params(kinds[e]);
        if (P < 0 || nptr + P > kRunMaxPtr) return CHX_ERR_INVALID_ARG;
        a.kind[e] = (uint8_t)kinds[e];
        a.off[e] = (uint16_t)nptr;
        for (int k = 0; k < P; ++k) {
            const void* q = param_ptrs[(size_t)e * CHX_MAX_PARAMS + k];
            if (!q) return CHX_ERR_INVALID_ARG;
            a.ptr[nptr++] = q;
        }
    }
    for (int e = (int)E; e < kRunMaxE; ++e) { a.kind[e] = 0; a.off[e] = 0; }
    for (int q = nptr; q < kRunMaxPtr; ++q) a.ptr[q] = nullptr;
    return CHX_OK;
}

extern "C" int chx_run_map(const int32_t* kinds, const void* const* param_ptrs, int64_t E, const void* energy, double mass_eV,
                           double n_charges, int dtype, void* state, size_t state_bytes, void** R_out, const void* s_in,
                           void* s_out, void* stream) {
    if (!energy || !state || ((s_in == nullptr) != (s_out == nullptr))) return CHX_ERR_INVALID_ARG;
    if (dtype != CHX_F32 && dtype != CHX_F64) return CHX_ERR_DTYPE;
    RunArgs a;
    int nptr = 0;
    int st = run_args(kinds, param_ptrs, E, a, nptr);
    if (st != CHX_OK) return st;
    if (state_bytes < chx_run_state_bytes(E)) return CHX_ERR_WORKSPACE;
    double* seen = (double*)state;
    double* maps = seen + kRunMaxPtr + 4;
    double* R = maps + E * 49;
    if (R_out) *R_out = R;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == CHX_F32)
        hipLaunchKernelGGL(run_map_kernel<float>, dim3(1), dim3(CHX_BLOCK), 0, s, a, (int)E, nptr, (const float*)energy, mass_eV,
                           n_charges, seen, (float*)maps, (float*)R, (const float*)s_in, (float*)s_out);
    else
        hipLaunchKernelGGL(run_map_kernel<double>, dim3(1), dim3(CHX_BLOCK), 0, s, a, (int)E, nptr, (const double*)energy,
                           mass_eV, n_charges, seen, maps, R, (const double*)s_in, (double*)s_out);
    CHX_CHECK_LAUNCH();
    return CHX_OK;
}

// ---- a run whose settings are VECTORISED over B lattice settings (a k1 scan, a batched environment, an orbit-response measurement:
// some parameters are (B,) tensors, the others scalars; the energy and every length scalar): all B composed maps in ONE launch — a
// workgroup per batch row builds the row's element maps (fp64 inside, rounded to T like chx_build_rmatrix) and composes them with
// the association of chx_compose_maps: bit-identical to the per-element builds + chx_compose_maps of the general path, which cost
// ~100 us of host time per vectorised element and step. A pointer with its lowest bit set addresses a (B,) array (element b of
// it belongs to row b), any other a scalar. No stored state: such settings change with every step.
template <typename T>
__global__ __launch_bounds__(CHX_BLOCK) void run_map_batched_kernel(RunArgs a, int E, const T* __restrict__ energy, double mass,
                                                                   double nq, T* __restrict__ maps_ws, T* __restrict__ R,
                                                                   int energy_rows /*energy is a (B,) array: a scan of beam energies*/) {
    extern __shared__ __attribute__((aligned(16))) unsigned char run_lds[];
    const int64_t b = blockIdx.x;
    T* maps = maps_ws ? maps_ws + b * E * 49 : reinterpret_cast<T*>(run_lds);
    for (int e = threadIdx.x; e < E; e += CHX_BLOCK) {
        const int kind = a.kind[e];
        const int P = kind_num_params(kind);
        double p[CHX_MAX_PARAMS];
        for (int k = 0; k < P; ++k) {
            const uintptr_t q = reinterpret_cast<uintptr_t>(a.ptr[a.off[e] + k]);
            const T* base = reinterpret_cast<const T*>(q & ~(uintptr_t)1);
            p[k] = (double)((q & 1) ? base[b] : base[0]);
        }
        Mat7<double> M;
        build_kind<double>(kind, p, (double)energy[energy_rows ? b : 0], mass, nq, M);
        for (int q = 0; q < 49; ++q) maps[e * 49 + q] = (T)M.m[q];
    }
    __syncthreads();
    T* Rb = R + b * 49;
    if (E == 1) {
        if (threadIdx.x < 49) Rb[threadIdx.x] = maps[threadIdx.x];
        return;
    }
    compose_block<T>([&](int e) { return (const T*)maps + e * 49; }, E, 0, Rb);
}

extern "C" size_t chx_run_map_batched_workspace_bytes(int64_t E, int64_t B, int dtype) {
    if (E < 1 || E > kRunMaxE || B < 1) return 0;
    const size_t per_row = (size_t)E * 49 * (dtype == CHX_F32 ? 4 : 8);
    return per_row <= 48 * 1024 ? 0 : per_row * (size_t)B;     // rows whose element maps fit the LDS need no scratch
}

extern "C" int chx_run_map_batched(const int32_t* kinds, const void* const* param_ptrs, const uint8_t* batched, int64_t E, int64_t B,
                                   const void* energy, int energy_rows, double mass_eV, double n_charges, int dtype, void* workspace,
                                   size_t workspace_bytes, void* R_out, void* stream) {
    if (!energy || !R_out || !batched || B < 1 || B > 0x7fffffffLL) return CHX_ERR_INVALID_ARG;
    if (dtype != CHX_F32 && dtype != CHX_F64) return CHX_ERR_DTYPE;
    RunArgs a;
    int nptr = 0;
    int st = run_args(kinds, param_ptrs, E, a, nptr);
    if (st != CHX_OK) return st;
    for (int e = 0, q = 0; e < (int)E; ++e) {
        const int P = kind_num_params(kinds[e]);
        for (int k = 0; k < P; ++k, ++q) {
            if (reinterpret_cast<uintptr_t>(a.ptr[q]) & 1) return CHX_ERR_MISALIGNED;
            if (batched[(size_t)e * CHX_MAX_PARAMS + k]) {
                if (k == 0) return CHX_ERR_INVALID_ARG;             // (a vectorised length: the path length is per row — not here)
                a.ptr[q] = reinterpret_cast<const void*>(reinterpret_cast<uintptr_t>(a.ptr[q]) | 1);
            }
        }
    }
    const size_t need = chx_run_map_batched_workspace_bytes(E, B, dtype);
    if (need && (!workspace || workspace_bytes < need)) return CHX_ERR_WORKSPACE;
    const size_t lds = need ? 0 : (size_t)E * 49 * (dtype == CHX_F32 ? 4 : 8);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == CHX_F32)
        hipLaunchKernelGGL(run_map_batched_kernel<float>, dim3((unsigned)B), dim3(CHX_BLOCK), lds, s, a, (int)E, (const float*)energy, mass_eV,
                           n_charges, need ? (float*)workspace : (float*)nullptr, (float*)R_out, energy_rows ? 1 : 0);
    else
        hipLaunchKernelGGL(run_map_batched_kernel<double>, dim3((unsigned)B), dim3(CHX_BLOCK), lds, s, a, (int)E, (const double*)energy,
                           mass_eV, n_charges, need ? (double*)workspace : (double*)nullptr, (double*)R_out, energy_rows ? 1 : 0);
    CHX_CHECK_LAUNCH();
    return CHX_OK;
}

// forward of a run whose settings carry gradients, one call: element maps into maps[E][7][7] (kept for the backward pass) and
// their product into R_out[7][7] — chx_build_rmatrix_scalars + chx_compose_maps, bit-identical to the two calls
extern "C" int chx_run_build_compose(const int32_t* kinds, const void* const* param_ptrs, int64_t E, const void* energy,
                                     double mass_eV, double n_charges, int dtype, void* maps, void* R_out, void* stream) {
    if (!maps || !R_out || !energy || E < 1 || E > 4096) return CHX_ERR_INVALID_ARG;
    if (dtype != CHX_F32 && dtype != CHX_F64) return CHX_ERR_DTYPE;
    {   // a run that fits the persistent-plan kernel: element maps and their product in ONE launch (run_map_kernel without stored
        // state; the same arithmetic, bit for bit, as the two launches below)
        RunArgs a;
        int nptr = 0;
        if (E <= kRunMaxE && run_args(kinds, param_ptrs, E, a, nptr) == CHX_OK) {
            hipStream_t s = (hipStream_t)stream;
            if (dtype == CHX_F32)
                hipLaunchKernelGGL(run_map_kernel<float>, dim3(1), dim3(CHX_BLOCK), 0, s, a, (int)E, nptr, (const float*)energy, mass_eV,
                                   n_charges, (double*)nullptr, (float*)maps, (float*)R_out, (const float*)nullptr, (float*)nullptr);
            else
                hipLaunchKernelGGL(run_map_kernel<double>, dim3(1), dim3(CHX_BLOCK), 0, s, a, (int)E, nptr, (const double*)energy,
                                   mass_eV, n_charges, (double*)nullptr, (double*)maps, (double*)R_out, (const double*)nullptr,
                                   (double*)nullptr);
            CHX_CHECK_LAUNCH();
            return CHX_OK;
        }
    }
    int st = chx_build_rmatrix_scalars(kinds, param_ptrs, E, energy, mass_eV, n_charges, dtype, maps, stream);
    if (st != CHX_OK) return st;
    const size_t step = 49 * (dtype == CHX_F32 ? 4 : 8);
    if (E == 1) {
        if (hipMemcpyAsync(R_out, maps, step, hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess) return CHX_ERR_LAUNCH;
        return CHX_OK;
    }
    const void* ptrs[4096];
    uint8_t bc[4096];
    for (int64_t e = 0; e < E; ++e) {
        ptrs[e] = (const char*)maps + e * step;
        bc[e] = 1;
    }
    return chx_compose_maps(ptrs, bc, E, 1, dtype, R_out, stream);
}


extern "C" int chx_run_track(const int32_t* kinds, const void* const* param_ptrs, int64_t E, const void* energy, double mass_eV,
                             double n_charges, int dtype, void* state, size_t state_bytes, const void* x_in, void* x_out,
                             int64_t N, const void* s_in, void* s_out, void* stream) {
    void* R = nullptr;
    int st = chx_run_map(kinds, param_ptrs, E, energy, mass_eV, n_charges, dtype, state, state_bytes, &R, s_in, s_out, stream);
    if (st != CHX_OK) return st;
    return chx_apply_affine7(x_in, R, x_out, 1, 1, 1, N, dtype, stream);
}

// ---- a whole stretch of lattice in two launches (chx_lattice_track, chx_apply.hip) --------------------------------------
// Segment.track (segment.py:545-574) walks [run of skippable elements, active Cavity, run, Cavity, ...] element by element; each
// item costs two launches here (map / coefficients, then the particle pass) and ~4 us of host time per launch, so a 16-cell
// linac on a 1e4-particle beam is 64 launches = 0.6 ms of host for ~60 us of kernels. The items of such a stretch depend on one
// another only through the REFERENCE ENERGY (every cavity hands E + V cos(phi) q on, cavity.py:113-122) and the path length —
// both scalars. So: ONE launch with a workgroup per item prepares every map (each workgroup first walks the energy through the
// cavities in front of its item: a few cosines), and ONE launch carries every particle through all items in registers
// (lattice_apply_kernel). The arithmetic per item is that of run_map_kernel / cavity_prepare_scalars_kernel — bit-identical maps,
// coefficients, energies and path length.
//   table (int64 words, device): items[n_items][4] = {type 0 run / 1 cavity, E, first element, -} or {2 active BPM, 0, index of
//   its misalignment's address in ptrs, reading slot} or {3 active aperture, 0 rectangular / 1 elliptical, index of the addresses
//   of x_max and y_max in ptrs, -}, elem_kind[n_elems],
//   elem_poff[n_elems] (first pointer of the element), ptrs[n_ptrs] (device addresses of the scalar settings, kind order)
//   state: R[n_items][49] (T, in double-sized slots), coeffs[n_items][8] double, emaps[n_elems][49] (T, double-sized slots)
constexpr int kLatticeMaxItems = 1024;

// Active Screens of the stretch (items of type 4, chx_lattice_track_screens): this call's output buffers by value, and where each
// screen's image starts among the spare workgroups that zero the images (kScreenZeroBytes each) — the memset launch of an image
// rides in the preparation launch.
constexpr int64_t kScreenZeroBytes = 16384;
struct LatticeScreens {
    chx_lattice_screen s[CHX_LATTICE_MAX_SCREENS];
    int64_t zero_first[CHX_LATTICE_MAX_SCREENS + 1];
    int n;
};

__device__ __forceinline__ void lattice_zero_images(const LatticeScreens& scr, int64_t z) {
    int slot = 0;
    while (slot + 1 < scr.n && z >= scr.zero_first[slot + 1]) ++slot;
    char* base = (char*)scr.s[slot].image;
    const int64_t lo = (z - scr.zero_first[slot]) * kScreenZeroBytes;
    int64_t hi = lo + kScreenZeroBytes;
    if (hi > scr.s[slot].image_bytes) hi = scr.s[slot].image_bytes;
    if (!base || lo >= hi) return;
    if ((((uintptr_t)base) & 15) == 0) {
        const int64_t n16 = (hi - lo) >> 4;
        float4* d = (float4*)(base + lo);
        // (streaming stores: 40 MB of zeros for an ARES-sized float64 image should not pass through the L2 on their way out)
        for (int64_t i = threadIdx.x; i < n16; i += CHX_BLOCK) chx_nt_store(make_float4(0.f, 0.f, 0.f, 0.f), d + i);
        for (int64_t i = lo + (n16 << 4) + threadIdx.x * 4; i < hi; i += CHX_BLOCK * 4) *(uint32_t*)(base + i) = 0u;
    } else {
        for (int64_t i = lo + threadIdx.x * 4; i < hi; i += CHX_BLOCK * 4) *(uint32_t*)(base + i) = 0u;
    }
}

template <typename T>
__global__ __launch_bounds__(CHX_BLOCK) void lattice_prepare_kernel(const int64_t* __restrict__ items, const int64_t* __restrict__ elem_kind,
                                                                   const int64_t* __restrict__ elem_poff, const int64_t* __restrict__ ptrs,
                                                                   int n_items, int n_elems, const T* __restrict__ energy, double mass, double nq,
                                                                   double* __restrict__ Rs, double* __restrict__ coeffs,
                                                                   double* __restrict__ emaps, T* __restrict__ energy_out,
                                                                   const T* __restrict__ s_in, T* __restrict__ s_out,
                                                                   int energy_rows /*energy is a (rows,) array*/,
                                                                   int energy_out_rows /*energy_out is a (rows,) array*/,
                                                                   LatticeScreens scr) {
    __shared__ double e_in_sh;
    if ((int)blockIdx.x >= n_items) {                      // (spare workgroups: the screens' images start from zero)
        lattice_zero_images(scr, (int64_t)blockIdx.x - n_items);
        return;
    }
    const int b = blockIdx.x;
    // blockIdx.y = row of a batch of lattice settings (gridDim.y = 1: scalar settings): a pointer with its lowest bit set addresses
    // a (rows,) array whose element `row` belongs to this row (chx_run_map_batched's convention), any other a scalar. The maps and
    // coefficient rows of row r sit behind those of the rows before it: Rs[(item * rows + r)], emaps[r][element].
    const int64_t row = blockIdx.y, rows = gridDim.y;
    const int type = (int)items[b * 4], E = (int)items[b * 4 + 1], elem0 = (int)items[b * 4 + 2];
    auto setting_of = [&](int64_t q, int64_t r) {
        const uintptr_t a = (uintptr_t)ptrs[q];
        const T* base = reinterpret_cast<const T*>(a & ~(uintptr_t)1);
        return (a & 1) ? base[r] : base[0];
    };
    auto setting = [&](int64_t q) { return setting_of(q, row); };
    {
        // the reference energy this item sees: through the cavities in front of it, rounded to T after each (the energy is a
        // tensor of the beam's dtype between two elements). The energy gains of the cavities in front are LOADED side by side (one
        // thread per item: four dependent loads each — item, element, address, value — 60 in a row for the last item of a 16-cell
        // linac when one thread walks them, ~60 us of a 105 us launch at 64 rows) and then summed in order by one thread.
        __shared__ double gain_sh[CHX_BLOCK];
        __shared__ int is_cavity_sh[CHX_BLOCK];
        double e = (double)energy[energy_rows ? row : 0];
        for (int base = 0; base < b; base += CHX_BLOCK) {
            const int i = base + (int)threadIdx.x;
            int cav = 0;
            double dEn = 0.0;
            if (i < b && items[i * 4] == 1) {
                const int64_t po = elem_poff[items[i * 4 + 2]];
                dEn = (double)setting(po + 1) * cos((double)setting(po + 2) * (kPi / 180.0)) * nq * -1.0;
                cav = 1;
            }
            gain_sh[threadIdx.x] = dEn;
            is_cavity_sh[threadIdx.x] = cav;
            __syncthreads();
            if (threadIdx.x == 0) {
                const int n = (b - base < CHX_BLOCK) ? (b - base) : CHX_BLOCK;
                for (int j = 0; j < n; ++j)
                    if (is_cavity_sh[j]) e = (double)(T)(e + gain_sh[j]);
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) e_in_sh = e;
    }
    if (b == n_items - 1 && s_out && row == 0) {           // (uniform over the workgroup)
        // path length behind the stretch: s + run length (((L0 + L1) + L2) + ...) / + cavity length, item by item, in T. Every
        // item's own length is summed by a thread of its own (its loads wait side by side with the other items'), one thread then
        // adds the items in order.
        __shared__ T total_sh[CHX_BLOCK];
        __shared__ int has_length_sh[CHX_BLOCK];
        T sv = *s_in;
        for (int base = 0; base < n_items; base += CHX_BLOCK) {
            const int i = base + (int)threadIdx.x;
            int has = 0;
            T total = (T)0;
            if (i < n_items && items[i * 4] < 2) {         // (a beam position monitor / an aperture / a screen: no length)
                const int Ei = (int)items[i * 4 + 1], e0 = (int)items[i * 4 + 2];
                total = setting(elem_poff[e0]);
                for (int e = 1; e < Ei; ++e) total = total + setting(elem_poff[e0 + e]);
                has = 1;
            } else if (i < n_items && items[i * 4] == 4) {
                has = 2 + (int)items[i * 4 + 3];           // an active screen: its record takes the path length up to here
            }
            total_sh[threadIdx.x] = total;
            has_length_sh[threadIdx.x] = has;
            __syncthreads();
            if (threadIdx.x == 0) {
                const int n = (n_items - base < CHX_BLOCK) ? (n_items - base) : CHX_BLOCK;
                for (int j = 0; j < n; ++j) {
                    if (has_length_sh[j] == 1) sv = sv + total_sh[j];
                    else if (has_length_sh[j] >= 2 && has_length_sh[j] - 2 < scr.n && scr.s[has_length_sh[j] - 2].s)
                        *(T*)scr.s[has_length_sh[j] - 2].s = sv;
                }
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) *s_out = sv;
    }
    __syncthreads();
    const double E0 = e_in_sh;
    T* R = reinterpret_cast<T*>(Rs + ((int64_t)b * rows + row) * 49);
    if (type >= 2) {                                       // an active BPM / aperture / screen: nothing to build (lattice_apply_kernel acts there)
        if (b == n_items - 1 && threadIdx.x == 0 && (row == 0 || energy_out_rows)) energy_out[energy_out_rows ? row : 0] = (T)E0;
        if (type == 4 && threadIdx.x == 0 && row == 0) {
            const int slot = (int)items[b * 4 + 3];
            if (slot < scr.n && scr.s[slot].energy) *(T*)scr.s[slot].energy = (T)E0;
        }
        return;
    }
    if (type == 1) {
        // cavity.py:157 switches the second-order path-length terms on for the WHOLE batch when ANY row gains energy: with a
        // vectorised voltage or phase the rows of this cavity are looked at together (uniform branch: one item per workgroup)
        int gains = 0;
        {
            const int64_t po = elem_poff[elem0];
            const bool vec = (((uintptr_t)ptrs[po + 1]) | ((uintptr_t)ptrs[po + 2])) & 1;
            if (vec) {
                for (int64_t r = threadIdx.x; r < rows; r += CHX_BLOCK)
                    gains |= ((double)setting_of(po + 1, r) * cos((double)setting_of(po + 2, r) * (kPi / 180.0)) * nq * -1.0) > 0.0;
                gains = __syncthreads_or(gains);
            } else {
                gains = ((double)setting(po + 1) * cos((double)setting(po + 2) * (kPi / 180.0)) * nq * -1.0) > 0.0;
            }
        }
        // the cavity's map and its coefficient row are two long chains of float64 arithmetic on ONE lane each (a few thousand dependent
        // instructions): on lanes of two different waves they run side by side (a 16-cell linac: 14.5 -> 13.6 us per launch; at 64 rows
        // of settings the launch stays at 76 us — 2048 workgroups of four waves are several rounds of the chip, see below)
        if (threadIdx.x == 0 || threadIdx.x == 64) {
            const int64_t po = elem_poff[elem0];
            const double p[4] = {(double)setting(po), (double)setting(po + 1), (double)setting(po + 2), (double)setting(po + 3)};
            if (threadIdx.x == 0) {
                Mat7<double> M;
                build_kind<double>((int)elem_kind[elem0], p, E0, mass, nq, M);
                for (int q = 0; q < 49; ++q) R[q] = (T)M.m[q];
            } else {
                const double E1 = cavity_coeff_row(p[0], p[1], p[2], p[3], E0, mass, nq, gains != 0, coeffs + ((int64_t)b * rows + row) * CHX_CAV_NCOEF);
                if (b == n_items - 1 && (row == 0 || energy_out_rows)) energy_out[energy_out_rows ? row : 0] = (T)E1;
            }
        }
        return;
    }
    if (b == n_items - 1 && threadIdx.x == 0 && (row == 0 || energy_out_rows)) energy_out[energy_out_rows ? row : 0] = (T)E0;
    T* maps = reinterpret_cast<T*>(emaps + (row * (int64_t)n_elems + elem0) * 49);
    for (int e = threadIdx.x; e < E; e += CHX_BLOCK) {
        const int kind = (int)elem_kind[elem0 + e];
        const int P = kind_num_params(kind);
        const int64_t po = elem_poff[elem0 + e];
        double p[CHX_MAX_PARAMS];
        for (int k = 0; k < P; ++k) p[k] = (double)setting(po + k);
        Mat7<double> M;
        build_kind<double>(kind, p, E0, mass, nq, M);
        for (int q = 0; q < 49; ++q) maps[e * 49 + q] = (T)M.m[q];
    }
    __syncthreads();
    // a screen right behind this run may want the run's composed map (chx_lattice_screen.map)
    T* map_out = nullptr;
    if (b + 1 < n_items && items[(b + 1) * 4] == 4 && row == 0) {
        const int slot = (int)items[(b + 1) * 4 + 3];
        if (slot < scr.n) {
            map_out = (T*)scr.s[slot].map;
            T* emaps_out = (T*)scr.s[slot].element_maps;
            if (emaps_out)
                for (int q = threadIdx.x; q < E * 49; q += CHX_BLOCK) emaps_out[q] = maps[q];
        }
    }
    if (E == 1) {
        if (threadIdx.x < 49) {
            R[threadIdx.x] = maps[threadIdx.x];
            if (map_out) map_out[threadIdx.x] = maps[threadIdx.x];
        }
        return;
    }
    compose_block<T>([&](int e) { return (const T*)maps + e * 49; }, E, 0, R);
    if (map_out && threadIdx.x < 49) map_out[threadIdx.x] = R[threadIdx.x];       // (thread t of wave 0 wrote R[t] itself)
}

// The same preparation for MANY rows of vectorised settings (an orbit response over thousands of corrector settings): one WAVE per
// (item, row) instead of one workgroup — a run between two monitors holds two or three elements, and 75 items x 4096 rows are
// 300 000 workgroups of 256 mostly idle threads (6.4 ms) otherwise. For stretches without cavities (one reference energy) whose
// runs hold at most 64 elements: lane e builds element e, the wave composes the maps with the association of compose_block (four
// chunk products from the identity, then P3 P2 P1 P0) — the same bits.
template <typename T, typename MapOf>
__device__ __forceinline__ void compose_wave(MapOf map_of, int E, T* __restrict__ R_row, double* cur4 /*[4][49]*/, double* nxt /*[49]*/) {
    const int lane = threadIdx.x & 63;
    const int i = lane / 7, j = lane - 7 * i;
    const int per = (E + 3) / 4;
    double acc = 0.0;
    for (int w = 0; w < 4; ++w) {
        const int e0 = w * per, e1 = (e0 + per < E) ? e0 + per : E;
        double* cur = cur4 + w * 49;
        acc = (lane < 49) ? ((i == j) ? 1.0 : 0.0) : 0.0;
        if (lane < 49) cur[lane] = acc;
        chx_wave_sync();
        for (int e = e0; e < e1; ++e) {
            const T* Re = map_of(e);
            if (lane < 49) nxt[lane] = (double)Re[lane];
            chx_wave_sync();
            if (lane < 49) {
                double s = nxt[i * 7] * cur[j];
                for (int k = 1; k < 7; ++k) s = fma(nxt[i * 7 + k], cur[k * 7 + j], s);
                acc = s;
            }
            chx_wave_sync();
            if (lane < 49) cur[lane] = acc;
            chx_wave_sync();
        }
    }
    for (int w = 1; w < 4; ++w) {
        if (lane < 49) {
            double s = cur4[w * 49 + i * 7] * cur4[j];
            for (int k = 1; k < 7; ++k) s = fma(cur4[w * 49 + i * 7 + k], cur4[k * 7 + j], s);
            acc = s;
        }
        chx_wave_sync();
        if (lane < 49) cur4[lane] = acc;
        chx_wave_sync();
    }
    if (lane < 49) R_row[lane] = (T)acc;
}

// CAV: the stretch may hold active cavities (lattice_prepare_kernel's arithmetic for them: the reference energy walked through the
// cavities in front of the item and rounded to T behind each, the cavity's map on lane 0 and its coefficient row on lane 1, cavity.py:157's
// batch-wide switch looked up over all rows). With a workgroup per (item, row) a 16-cell linac at 64 energies is 2048 workgroups of four
// mostly idle waves — several rounds of the chip, 76 us; a wave per (item, row): one round.
template <typename T, bool CAV>
__global__ __launch_bounds__(CHX_BLOCK) void lattice_prepare_rows_kernel(const int64_t* __restrict__ items, const int64_t* __restrict__ elem_kind,
                                                                        const int64_t* __restrict__ elem_poff, const int64_t* __restrict__ ptrs,
                                                                        int n_items, int n_elems, int64_t rows, const T* __restrict__ energy,
                                                                        double mass, double nq, double* __restrict__ Rs,
                                                                        double* __restrict__ coeffs, double* __restrict__ emaps,
                                                                        T* __restrict__ energy_out, const T* __restrict__ s_in,
                                                                        T* __restrict__ s_out,
                                                                        int energy_rows /*energy is a (rows,) array*/,
                                                                        int energy_out_rows /*energy_out is a (rows,) array*/) {
    __shared__ double lds[CHX_BLOCK / 64][5 * 49];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t pid = (int64_t)blockIdx.x * (CHX_BLOCK / 64) + wave;
    if (pid >= (int64_t)n_items * rows) return;             // (whole waves; nothing below synchronises the workgroup)
    const int b = (int)(pid / rows);
    const int64_t row = pid - (int64_t)b * rows;
    auto setting_of = [&](int64_t q, int64_t r) {
        const uintptr_t a = (uintptr_t)ptrs[q];
        const T* base = reinterpret_cast<const T*>(a & ~(uintptr_t)1);
        return (a & 1) ? base[r] : base[0];
    };
    auto setting = [&](int64_t q) { return setting_of(q, row); };
    if (!CAV) {
        if (energy_rows && b == 0 && lane == 0) energy_out[row] = energy[row];       // (no cavity in the stretch)
        if (pid == 0 && lane == 0 && !energy_rows) *energy_out = energy[0];
    }
    if (pid == 0 && lane == 0 && s_out) {
        T sv = *s_in;
        for (int i = 0; i < n_items; ++i) {
            if (items[i * 4] >= 2) continue;                // (monitors and apertures have no length)
            const int Ei = (int)items[i * 4 + 1], e0 = (int)items[i * 4 + 2];
            T total = setting(elem_poff[e0]);
            for (int e = 1; e < Ei; ++e) total = total + setting(elem_poff[e0 + e]);
            sv = sv + total;
        }
        *s_out = sv;
    }
    const int type = (int)items[b * 4];
    double E0 = (double)energy[energy_rows ? row : 0];
    if (CAV) {
        // the reference energy this item sees: the gains of the cavities in front of it are evaluated side by side (lane i: item
        // base + i), lane 0 adds them in item order, rounding to T behind every cavity like the tensor between two elements
        double* gain = &lds[wave][0];
        double* flag = &lds[wave][64];
        for (int base = 0; base < b; base += 64) {
            const int i = base + lane;
            double dEn = 0.0, cav = 0.0;
            if (i < b && items[i * 4] == 1) {
                const int64_t po = elem_poff[items[i * 4 + 2]];
                dEn = (double)setting(po + 1) * cos((double)setting(po + 2) * (kPi / 180.0)) * nq * -1.0;
                cav = 1.0;
            }
            gain[lane] = dEn;
            flag[lane] = cav;
            chx_wave_sync();
            if (lane == 0) {
                const int n = (b - base < 64) ? (b - base) : 64;
                double e = E0;
                for (int j = 0; j < n; ++j)
                    if (flag[j] != 0.0) e = (double)(T)(e + gain[j]);
                gain[128] = e;
            }
            chx_wave_sync();
            E0 = gain[128];
            chx_wave_sync();
        }
    }
    T* R = reinterpret_cast<T*>(Rs + ((int64_t)b * rows + row) * 49);
    if (type >= 2) {
        if (CAV && b == n_items - 1 && lane == 0 && (row == 0 || energy_out_rows)) energy_out[energy_out_rows ? row : 0] = (T)E0;
        return;
    }
    if (type == 1) {
        if (!CAV) return;
        const int elem0c = (int)items[b * 4 + 2];
        const int64_t po = elem_poff[elem0c];
        // cavity.py:157: the second-order path-length terms are on for the WHOLE batch when ANY row gains energy
        bool gains;
        const bool vec = (((uintptr_t)ptrs[po + 1]) | ((uintptr_t)ptrs[po + 2])) & 1;
        if (vec) {
            bool g = false;
            for (int64_t r = lane; r < rows; r += 64)
                g = g || ((double)setting_of(po + 1, r) * cos((double)setting_of(po + 2, r) * (kPi / 180.0)) * nq * -1.0) > 0.0;
            gains = __ballot(g) != 0ull;
        } else {
            gains = ((double)setting(po + 1) * cos((double)setting(po + 2) * (kPi / 180.0)) * nq * -1.0) > 0.0;
        }
        if (lane < 2) {
            const double p[4] = {(double)setting(po), (double)setting(po + 1), (double)setting(po + 2), (double)setting(po + 3)};
            if (lane == 0) {
                Mat7<double> M;
                build_kind<double>((int)elem_kind[elem0c], p, E0, mass, nq, M);
                for (int q = 0; q < 49; ++q) R[q] = (T)M.m[q];
            } else {
                const double E1 = cavity_coeff_row(p[0], p[1], p[2], p[3], E0, mass, nq, gains, coeffs + ((int64_t)b * rows + row) * CHX_CAV_NCOEF);
                if (b == n_items - 1 && (row == 0 || energy_out_rows)) energy_out[energy_out_rows ? row : 0] = (T)E1;
            }
        }
        return;
    }
    if (CAV && b == n_items - 1 && lane == 0 && (row == 0 || energy_out_rows)) energy_out[energy_out_rows ? row : 0] = (T)E0;
    const int E = (int)items[b * 4 + 1], elem0 = (int)items[b * 4 + 2];
    T* maps = reinterpret_cast<T*>(emaps + (row * (int64_t)n_elems + elem0) * 49);
    if (lane < E) {
        const int kind = (int)elem_kind[elem0 + lane];
        const int P = kind_num_params(kind);
        const int64_t po = elem_poff[elem0 + lane];
        double p[CHX_MAX_PARAMS];
        for (int k = 0; k < P; ++k) p[k] = (double)setting(po + k);
        Mat7<double> M;
        build_kind<double>(kind, p, E0, mass, nq, M);
        for (int q = 0; q < 49; ++q) maps[lane * 49 + q] = (T)M.m[q];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (E == 1) {
        if (lane < 49) R[lane] = maps[lane];
        return;
    }
    compose_wave<T>([&](int e) { return (const T*)maps + e * 49; }, E, R, &lds[wave][0], &lds[wave][4 * 49]);
}

extern "C" size_t chx_lattice_state_bytes_batched(int64_t n_items, int64_t n_elems, int64_t rows) {
    if (n_items < 1 || n_items > kLatticeMaxItems || n_elems < 1 || rows < 1 || rows > 65535) return 0;      // (BPM items hold no element)
    return (size_t)(n_items * (49 + CHX_CAV_NCOEF) + n_elems * 49) * (size_t)rows * sizeof(double);
}

extern "C" size_t chx_lattice_state_bytes(int64_t n_items, int64_t n_elems) { return chx_lattice_state_bytes_batched(n_items, n_elems, 1); }

extern "C" int chx_lattice_prepare_batched(const int64_t* table, int64_t n_items, int64_t n_elems, int64_t n_ptrs, int64_t rows,
                                           const void* energy, double mass_eV, double n_charges, int dtype, void* state, size_t state_bytes,
                                           void* energy_out, const void* s_in, void* s_out, void* stream) {
    return chx_lattice_prepare_rows(table, n_items, n_elems, n_ptrs, rows, 0, energy, mass_eV, n_charges, dtype, state, state_bytes,
                                    energy_out, s_in, s_out, stream);
}

// small_runs bit 0: the caller vouches that the stretch holds NO cavity and that every run has at most 64 elements; bit 3
// (CHX_LATTICE_SHORT_RUNS): every run has at most 64 elements, cavities or not — either way the wave-per-(item, row) kernel prepares
// it (same results; with cavities every wave walks the energy through the cavities in front of its item), for any number of rows
extern "C" int chx_lattice_prepare_rows(const int64_t* table, int64_t n_items, int64_t n_elems, int64_t n_ptrs, int64_t rows, int small_runs,
                                        const void* energy, double mass_eV, double n_charges, int dtype, void* state, size_t state_bytes,
                                        void* energy_out, const void* s_in, void* s_out, void* stream) {
    return chx_lattice_prepare_screens(table, n_items, n_elems, n_ptrs, rows, small_runs, energy, mass_eV, n_charges, dtype, state,
                                       state_bytes, energy_out, s_in, s_out, nullptr, 0, stream);
}

// ---- a stretch's table from HOST memory to the device without a staging buffer: the words ride in the kernel arguments ------------
// A control loop that ASSIGNS new tensors to a few settings per step (`quad.k1 = action[0]`, README.md:73-77 of the reference) leaves
// the lattice's layout alone and changes a handful of addresses in the table; re-uploading it through a page-locked staging tensor and
// an asynchronous copy cost ~12 us of host time per step (allocation, two torch calls). Kernel arguments are copied at launch: no
// staging buffer to keep alive, no race with a launch still in flight, stream-ordered in front of the preparation launch.
namespace {
constexpr int kTableStoreWords = 448;            // 3.5 KB of the kernel-argument segment
struct TableWords { int64_t w[kTableStoreWords]; };
__global__ __launch_bounds__(CHX_BLOCK) void table_store_kernel(TableWords t, int n, int64_t* __restrict__ dst) {
    for (int i = threadIdx.x; i < n; i += CHX_BLOCK) dst[i] = t.w[i];
}
}  // namespace

extern "C" int64_t chx_table_store_max_words(void) { return kTableStoreWords; }

extern "C" int chx_table_store(const int64_t* host_words, int64_t n, void* table, void* stream) {
    if (!host_words || !table || n < 1 || n > kTableStoreWords) return CHX_ERR_INVALID_ARG;
    TableWords t;
    for (int64_t i = 0; i < n; ++i) t.w[i] = host_words[i];
    hipLaunchKernelGGL(table_store_kernel, dim3(1), dim3(CHX_BLOCK), 0, (hipStream_t)stream, t, (int)n, (int64_t*)table);
    CHX_CHECK_LAUNCH();
    return CHX_OK;
}

extern "C" int chx_lattice_prepare_screens(const int64_t* table, int64_t n_items, int64_t n_elems, int64_t n_ptrs, int64_t rows, int small_runs,
                                           const void* energy, double mass_eV, double n_charges, int dtype, void* state, size_t state_bytes,
                                           void* energy_out, const void* s_in, void* s_out, const chx_lattice_screen* screens,
                                           int64_t n_screens, void* stream) {
    if (!table || !energy || !state || !energy_out || ((s_in == nullptr) != (s_out == nullptr)) || n_ptrs < n_elems)
        return CHX_ERR_INVALID_ARG;
    if (n_screens < 0 || n_screens > CHX_LATTICE_MAX_SCREENS || (n_screens > 0 && (!screens || rows != 1))) return CHX_ERR_INVALID_ARG;
    LatticeScreens scr;
    scr.n = (int)n_screens;
    int64_t zero_blocks = 0;
    for (int k = 0; k < CHX_LATTICE_MAX_SCREENS; ++k) {
        scr.zero_first[k] = zero_blocks;
        if (k < n_screens) {
            scr.s[k] = screens[k];
            if (!screens[k].image && screens[k].mom_partials) {
                // the sets of moment sums the particle pass adds into start from zero like an image does (this kernel reads
                // `image` for nothing else)
                scr.s[k].image = screens[k].mom_partials;
                scr.s[k].image_bytes = CHX_LATTICE_MOMENT_DOUBLES * (int64_t)sizeof(double);
            }
            if (scr.s[k].image) {
                if (scr.s[k].image_bytes < 0 || (scr.s[k].image_bytes & 3)) return CHX_ERR_INVALID_ARG;
                zero_blocks += (scr.s[k].image_bytes + kScreenZeroBytes - 1) / kScreenZeroBytes;
            }
        } else {
            scr.s[k] = chx_lattice_screen{};
        }
    }
    scr.zero_first[CHX_LATTICE_MAX_SCREENS] = zero_blocks;
    if (zero_blocks > 1000000) return CHX_ERR_INVALID_ARG;
    const int energy_rows = (small_runs & CHX_LATTICE_ENERGY_ROWS) ? 1 : 0;   // bit 1: energy / energy_out are (rows,) arrays
    // bit 2: energy_out alone is a (rows,) array (a cavity with a vectorised voltage or phase behind a scalar incoming energy)
    const int energy_out_rows = (energy_rows || (small_runs & CHX_LATTICE_ENERGY_OUT_ROWS)) ? 1 : 0;
    if (energy_out_rows != energy_rows && (small_runs & 1)) return CHX_ERR_INVALID_ARG;       // (no cavity in a small-runs stretch)
    // bit 3: every run has at most 64 elements, cavities or not — the wave-per-(item, row) kernel with the cavity arithmetic
    const int short_runs = (small_runs & CHX_LATTICE_SHORT_RUNS) ? 1 : 0;
    small_runs &= 1;
    const size_t need = chx_lattice_state_bytes_batched(n_items, n_elems, rows);
    if (need == 0) return CHX_ERR_INVALID_ARG;
    if (state_bytes < need) return CHX_ERR_WORKSPACE;
    if (dtype != CHX_F32 && dtype != CHX_F64) return CHX_ERR_DTYPE;
    const int64_t* items = table;
    const int64_t* elem_kind = items + n_items * 4;
    const int64_t* elem_poff = elem_kind + n_elems;
    const int64_t* ptrs = elem_poff + n_elems;
    double* Rs = (double*)state;
    double* coeffs = Rs + n_items * rows * 49;
    double* emaps = coeffs + n_items * rows * CHX_CAV_NCOEF;
    hipStream_t s = (hipStream_t)stream;
    if ((small_runs || short_runs) && rows > 1) {
        const int64_t pairs = n_items * rows;
        const unsigned g = (unsigned)((pairs + CHX_BLOCK / 64 - 1) / (CHX_BLOCK / 64));
#define CHX_PREPARE_ROWS(T, CAV)                                                                                                          \
    hipLaunchKernelGGL((lattice_prepare_rows_kernel<T, CAV>), dim3(g), dim3(CHX_BLOCK), 0, s, items, elem_kind, elem_poff, ptrs, (int)n_items, \
                       (int)n_elems, rows, (const T*)energy, mass_eV, n_charges, Rs, coeffs, emaps, (T*)energy_out, (const T*)s_in,          \
                       (T*)s_out, energy_rows, energy_out_rows)
        if (dtype == CHX_F32) {
            if (small_runs) CHX_PREPARE_ROWS(float, false);
            else CHX_PREPARE_ROWS(float, true);
        } else {
            if (small_runs) CHX_PREPARE_ROWS(double, false);
            else CHX_PREPARE_ROWS(double, true);
        }
#undef CHX_PREPARE_ROWS
        CHX_CHECK_LAUNCH();
        return CHX_OK;
    }
    const dim3 grid((unsigned)(n_items + zero_blocks), (unsigned)rows);
    if (dtype == CHX_F32)
        hipLaunchKernelGGL(lattice_prepare_kernel<float>, grid, dim3(CHX_BLOCK), 0, s, items, elem_kind, elem_poff, ptrs,
                           (int)n_items, (int)n_elems, (const float*)energy, mass_eV, n_charges, Rs, coeffs, emaps, (float*)energy_out,
                           (const float*)s_in, (float*)s_out, energy_rows, energy_out_rows, scr);
    else
        hipLaunchKernelGGL(lattice_prepare_kernel<double>, grid, dim3(CHX_BLOCK), 0, s, items, elem_kind, elem_poff,
                           ptrs, (int)n_items, (int)n_elems, (const double*)energy, mass_eV, n_charges, Rs, coeffs, emaps, (double*)energy_out,
                           (const double*)s_in, (double*)s_out, energy_rows, energy_out_rows, scr);
    CHX_CHECK_LAUNCH();
    return CHX_OK;
}

extern "C" int chx_lattice_prepare(const int64_t* table, int64_t n_items, int64_t n_elems, int64_t n_ptrs, const void* energy,
                                   double mass_eV, double n_charges, int dtype, void* state, size_t state_bytes, void* energy_out,
                                   const void* s_in, void* s_out, void* stream) {
    return chx_lattice_prepare_batched(table, n_items, n_elems, n_ptrs, 1, energy, mass_eV, n_charges, dtype, state, state_bytes,
                                       energy_out, s_in, s_out, stream);
}

extern "C" size_t chx_compose_maps_vjp_workspace_bytes(int64_t E, int64_t B) {
    return (E < 1 || B < 1) ? 0 : (size_t)E * B * 49 * sizeof(double);
}

extern "C" int chx_compose_maps_vjp(const void* const* R_ptrs, const uint8_t* bcast, int64_t E, int64_t B, int dtype, const void* dT,
                                    void* dM, void* workspace, size_t workspace_bytes, void* stream) {
    if (!R_ptrs || !bcast || !dT || !dM || E < 1 || E > kComposeChunk || B < 1 || B > 0x7fffffffLL) return CHX_ERR_INVALID_ARG;
    if (dtype != CHX_F32 && dtype != CHX_F64) return CHX_ERR_DTYPE;
    if (!workspace || workspace_bytes < chx_compose_maps_vjp_workspace_bytes(E, B)) return CHX_ERR_WORKSPACE;
    ComposeArgs a;
    for (int e = 0; e < kComposeChunk; ++e) {
        a.ptr[e] = e < E ? R_ptrs[e] : nullptr;
        a.bcast[e] = e < E ? bcast[e] : 1;
        if (e < E && !a.ptr[e]) return CHX_ERR_INVALID_ARG;
    }
    hipStream_t s = (hipStream_t)stream;
    if (dtype == CHX_F32)
        hipLaunchKernelGGL(compose_vjp_kernel<float>, dim3((unsigned)B), dim3(64), 0, s, a, (int)E, (const float*)dT, (double*)workspace,
                           (float*)dM);
    else
        hipLaunchKernelGGL(compose_vjp_kernel<double>, dim3((unsigned)B), dim3(64), 0, s, a, (int)E, (const double*)dT,
                           (double*)workspace, (double*)dM);
    CHX_CHECK_LAUNCH();
    return CHX_OK;
}

extern "C" int chx_compose_maps(const void* const* R_ptrs, const uint8_t* bcast, int64_t E, int64_t B,
                                int dtype, void* R_out, void* stream) {
    if (!R_ptrs || !bcast || !R_out || E < 1 || B < 1 || B > 0x7fffffffLL) return CHX_ERR_INVALID_ARG;
    if (dtype != CHX_F32 && dtype != CHX_F64) return CHX_ERR_DTYPE;
    hipStream_t s = (hipStream_t)stream;
    int64_t done = 0;
    while (done < E) {
        ComposeArgs a;
        const int n = (int)((E - done < kComposeChunk) ? (E - done) : kComposeChunk);
        for (int e = 0; e < n; ++e) {
            if (!R_ptrs[done + e]) return CHX_ERR_INVALID_ARG;
            a.ptr[e] = R_ptrs[done + e];
            a.bcast[e] = bcast[done + e];
        }
        for (int e = n; e < kComposeChunk; ++e) { a.ptr[e] = nullptr; a.bcast[e] = 1; }
        if (dtype == CHX_F32)
            hipLaunchKernelGGL(compose_kernel<float>, dim3((unsigned)B), dim3(CHX_BLOCK), 0, s, a, n,
                               (int)(done > 0), (float*)R_out);
        else
            hipLaunchKernelGGL(compose_kernel<double>, dim3((unsigned)B), dim3(CHX_BLOCK), 0, s, a, n,
                               (int)(done > 0), (double*)R_out);
        CHX_CHECK_LAUNCH();
        done += n;
    }
    return CHX_OK;
}

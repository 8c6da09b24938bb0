// chx_sc_math.h — closed forms shared by the space-charge kernels and their backward passes; templates over
// S = double (tracking) or Dual (derivatives), see chx_dual.h.
#pragma once
#include "chx_dual.h"

namespace {

// scipy.constants (CODATA 2022, scipy 1.15.3) as used by the reference
constexpr double kC = 299792458.0;
constexpr double kElementaryCharge = 1.602176634e-19;
constexpr double kEvToKg = 1.7826619216278975e-36;  // physical_constants["electron volt-kilogram relationship"]

// space_charge_kick.py:103-123
template <typename S>
__device__ __forceinline__ S igf_primitive(S x, S y, S t) {
    const S r = m_sqrt(x * x + y * y + t * t);
    return -0.5 * t * t * m_atan(x * y / (t * r)) - 0.5 * y * y * m_atan(x * t / (y * r)) -
           0.5 * x * x * m_atan(y * t / (x * r)) + y * t * m_asinh(x / m_sqrt(y * y + t * t)) +
           x * t * m_asinh(y / m_sqrt(x * x + t * t)) + x * y * m_asinh(t / m_sqrt(x * x + y * y));
}

// Grid geometry of one kick from the three beam variances (space_charge_kick.py:531-550). Every step is rounded in T in
// the order the reference's tensor expressions round (sigma = sqrt(cov) cast to T, half = extent * sigma,
// cell = 2 half / g, gamma = E / m, beta, dt = L / (c beta)).
template <typename T>
__device__ __forceinline__ void sc_geometry_row(const double (&var)[3], const T* __restrict__ ext3, T energy, T length,
                                                double mass, double pot_factor, int gx, int gy, int gz,
                                                T* __restrict__ half3, T* __restrict__ cell3, T* __restrict__ gamma_out,
                                                T* __restrict__ dt, T* __restrict__ scale3, T* __restrict__ extent6,
                                                double* __restrict__ pot_scale) {
    const T g[3] = {(T)gx, (T)gy, (T)gz};
    double vol = 1.0;
    for (int d = 0; d < 3; ++d) {
        const T sig = (T)sqrt(var[d]);
        const T h = ext3[d] * sig;
        const T c = ((T)2 * h) / g[d];
        half3[d] = h;
        cell3[d] = c;
        extent6[d * 2 + 0] = -h;
        extent6[d * 2 + 1] = h;
        vol *= (double)c;
    }
    const T gam = energy / (T)mass;
    const T ig2 = (T)1 / (gam * gam);
    T one_minus = (T)1 - ig2;
    if (one_minus < (T)0) one_minus = (T)0;
    const T beta = (fabs((double)gam) > 0.0) ? (T)sqrt(one_minus) : (T)1;
    *gamma_out = gam;
    *dt = length / ((T)299792458.0 * beta);
    scale3[0] = (T)1;
    scale3[1] = (T)1;
    scale3[2] = -beta;
    *pot_scale = (1.0 / vol) * pot_factor;
}

template <typename S>
struct RefFrame {
    S gamma, beta, p0;  // reference gamma / beta, p0 = gamma beta m c
    double mc;          // m c
};

template <typename S>
__device__ __forceinline__ RefFrame<S> ref_frame(S energy, double mass_eV) {
    RefFrame<S> r;
    r.gamma = energy / mass_eV;                                                             // beam.py:323-326
    r.beta = (fabs(val(r.gamma)) > 0.0) ? m_sqrt(1.0 - 1.0 / (r.gamma * r.gamma)) : cst<S>(1.0);  // beam.py:328-336
    r.mc = mass_eV * kEvToKg * kC;
    r.p0 = r.gamma * r.beta * r.mc;
    return r;
}

// particle_beam.py:1316-1346
template <typename S>
__device__ __forceinline__ void to_si(const RefFrame<S>& r, const S (&v)[7], S (&s)[7]) {
    const S gi = r.gamma * (1.0 + v[5] * r.beta);
    const S bi = m_sqrt(1.0 - 1.0 / (gi * gi));
    const S P = gi * bi * r.mc;
    const S px = v[1] * r.p0, py = v[3] * r.p0;
    s[0] = v[0];
    s[1] = px;
    s[2] = v[2];
    s[3] = py;
    s[4] = v[4] * -r.beta;
    s[5] = m_sqrt(P * P - px * px - py * py);
    s[6] = v[6];
}

// particle_beam.py:1262-1314
template <typename S>
__device__ __forceinline__ void from_si(const RefFrame<S>& r, const S (&s)[7], S (&v)[7]) {
    const S p = m_sqrt(s[1] * s[1] + s[3] * s[3] + s[5] * s[5]);
    const S q = p / r.mc;
    const S g = m_sqrt(1.0 + q * q);
    v[0] = s[0];
    v[1] = s[1] / r.p0;
    v[2] = s[2];
    v[3] = s[3] / r.p0;
    v[4] = -s[4] / r.beta;
    v[5] = (g - r.gamma) / (r.beta * r.gamma);
    v[6] = s[6];
}

}  // namespace

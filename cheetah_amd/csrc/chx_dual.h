// chx_dual.h — forward-mode dual numbers for the device-side derivative kernels (chx_build_rmatrix_vjp,
// chx_build_ttensor_vjp, chx_dkd_track_bwd). Every closed form in libchx that has a derivative is a template over a
// scalar S that is either `double` (the forward kernels: identical arithmetic to the plain code) or `Dual`
// (value + one tangent); a backward kernel seeds one input at a time and contracts the tangent of the outputs with
// the incoming cotangent. This is what replaces torch autograd through the reference's tensor expressions.
#pragma once
#include <hip/hip_runtime.h>

struct Dual {
    double v, d;
};
__device__ __forceinline__ Dual mk(double v, double d) { Dual r; r.v = v; r.d = d; return r; }
__device__ __forceinline__ Dual operator+(Dual a, Dual b) { return mk(a.v + b.v, a.d + b.d); }
__device__ __forceinline__ Dual operator-(Dual a, Dual b) { return mk(a.v - b.v, a.d - b.d); }
__device__ __forceinline__ Dual operator*(Dual a, Dual b) { return mk(a.v * b.v, a.d * b.v + a.v * b.d); }
__device__ __forceinline__ Dual operator/(Dual a, Dual b) {
    const double q = a.v / b.v;
    return mk(q, (a.d - q * b.d) / b.v);
}
__device__ __forceinline__ Dual operator-(Dual a) { return mk(-a.v, -a.d); }
__device__ __forceinline__ Dual operator+(Dual a, double b) { return mk(a.v + b, a.d); }
__device__ __forceinline__ Dual operator+(double a, Dual b) { return mk(a + b.v, b.d); }
__device__ __forceinline__ Dual operator-(Dual a, double b) { return mk(a.v - b, a.d); }
__device__ __forceinline__ Dual operator-(double a, Dual b) { return mk(a - b.v, -b.d); }
__device__ __forceinline__ Dual operator*(Dual a, double b) { return mk(a.v * b, a.d * b); }
__device__ __forceinline__ Dual operator*(double a, Dual b) { return mk(a * b.v, a * b.d); }
__device__ __forceinline__ Dual operator/(Dual a, double b) { return mk(a.v / b, a.d / b); }
__device__ __forceinline__ Dual operator/(double a, Dual b) {
    const double q = a / b.v;
    return mk(q, -q * b.d / b.v);
}

__device__ __forceinline__ double val(double x) { return x; }
__device__ __forceinline__ double val(Dual x) { return x.v; }
__device__ __forceinline__ double tan_of(double x) { return 0.0 * x; }
__device__ __forceinline__ double tan_of(Dual x) { return x.d; }

__device__ __forceinline__ double m_sqrt(double x) { return sqrt(x); }
__device__ __forceinline__ double m_sin(double x) { return sin(x); }
__device__ __forceinline__ double m_cos(double x) { return cos(x); }
__device__ __forceinline__ double m_tan(double x) { return tan(x); }
__device__ __forceinline__ double m_sinh(double x) { return sinh(x); }
__device__ __forceinline__ double m_cosh(double x) { return cosh(x); }
__device__ __forceinline__ double m_log1p(double x) { return log1p(x); }
__device__ __forceinline__ Dual m_sqrt(Dual x) { const double s = sqrt(x.v); return mk(s, 0.5 * x.d / s); }
__device__ __forceinline__ Dual m_sin(Dual x) { return mk(sin(x.v), cos(x.v) * x.d); }
__device__ __forceinline__ Dual m_cos(Dual x) { return mk(cos(x.v), -sin(x.v) * x.d); }
__device__ __forceinline__ Dual m_tan(Dual x) { const double t = tan(x.v); return mk(t, (1.0 + t * t) * x.d); }
__device__ __forceinline__ Dual m_sinh(Dual x) { return mk(sinh(x.v), cosh(x.v) * x.d); }
__device__ __forceinline__ Dual m_cosh(Dual x) { return mk(cosh(x.v), sinh(x.v) * x.d); }
__device__ __forceinline__ Dual m_log1p(Dual x) { return mk(log1p(x.v), x.d / (1.0 + x.v)); }

template <typename S> __device__ __forceinline__ S cst(double c);
template <> __device__ __forceinline__ double cst<double>(double c) { return c; }
template <> __device__ __forceinline__ Dual cst<Dual>(double c) { return mk(c, 0.0); }

__device__ __forceinline__ double m_asin(double x) { return asin(x); }
__device__ __forceinline__ double m_atan2(double y, double x) { return atan2(y, x); }
__device__ __forceinline__ Dual m_asin(Dual x) { return mk(asin(x.v), x.d / sqrt(1.0 - x.v * x.v)); }
__device__ __forceinline__ Dual m_atan2(Dual y, Dual x) {
    return mk(atan2(y.v, x.v), (x.v * y.d - y.v * x.d) / (x.v * x.v + y.v * y.v));
}

__device__ __forceinline__ double m_atan(double x) { return atan(x); }
__device__ __forceinline__ double m_asinh(double x) { return asinh(x); }
__device__ __forceinline__ double m_abs(double x) { return fabs(x); }
__device__ __forceinline__ Dual m_atan(Dual x) { return mk(atan(x.v), x.d / (1.0 + x.v * x.v)); }
__device__ __forceinline__ Dual m_asinh(Dual x) { return mk(asinh(x.v), x.d / sqrt(1.0 + x.v * x.v)); }
// d|x| = sign(x) dx with sign(0) = 0: torch's convention for `abs` (the reference's trilinear weights are 1 - |u - i|,
// space_charge_kick.py:413-415, so a particle sitting exactly on a grid node gets this sub-gradient there)
__device__ __forceinline__ Dual m_abs(Dual x) { return x.v < 0.0 ? mk(-x.v, -x.d) : (x.v > 0.0 ? x : mk(0.0, 0.0)); }

// ---- F32: the same closed forms evaluated in float32 (opt-in `precision="storage"` of the drift-kick-drift kernels: the
// reference's Bmad-X arithmetic runs in the storage dtype, cheetah/utils/bmadx.py). A wrapper instead of plain `float` so that
// the double literals of the templates (1.0 + x, 0.5 * x, kPi / 2.0 ...) are rounded to float and the operation itself is a
// float operation — `1.0 + x` with a plain float would be promoted to a v_add_f64.
struct F32 {
    float v;
};
__device__ __forceinline__ F32 mkf(float v) { F32 r; r.v = v; return r; }
__device__ __forceinline__ F32 operator+(F32 a, F32 b) { return mkf(a.v + b.v); }
__device__ __forceinline__ F32 operator-(F32 a, F32 b) { return mkf(a.v - b.v); }
__device__ __forceinline__ F32 operator*(F32 a, F32 b) { return mkf(a.v * b.v); }
// divisions and square roots through the hardware approximations (v_rcp_f32, v_sqrt_f32: 1 ulp) instead of the IEEE-exact
// expansions (~10 instructions each): this type exists to be HBM-bound, and its results carry the rounding of a float32
// evaluation of the whole map anyway
__device__ __forceinline__ F32 operator/(F32 a, F32 b) { return mkf(a.v * __builtin_amdgcn_rcpf(b.v)); }
__device__ __forceinline__ F32 operator-(F32 a) { return mkf(-a.v); }
__device__ __forceinline__ F32 operator+(F32 a, double b) { return mkf(a.v + (float)b); }
__device__ __forceinline__ F32 operator+(double a, F32 b) { return mkf((float)a + b.v); }
__device__ __forceinline__ F32 operator-(F32 a, double b) { return mkf(a.v - (float)b); }
__device__ __forceinline__ F32 operator-(double a, F32 b) { return mkf((float)a - b.v); }
__device__ __forceinline__ F32 operator*(F32 a, double b) { return mkf(a.v * (float)b); }
__device__ __forceinline__ F32 operator*(double a, F32 b) { return mkf((float)a * b.v); }
__device__ __forceinline__ F32 operator/(F32 a, double b) { return mkf(a.v * __builtin_amdgcn_rcpf((float)b)); }
__device__ __forceinline__ F32 operator/(double a, F32 b) { return mkf((float)a * __builtin_amdgcn_rcpf(b.v)); }
__device__ __forceinline__ double val(F32 x) { return (double)x.v; }
__device__ __forceinline__ double tan_of(F32 x) { return 0.0 * x.v; }
template <> __device__ __forceinline__ F32 cst<F32>(double c) { return mkf((float)c); }
__device__ __forceinline__ F32 m_sqrt(F32 x) { return mkf(__builtin_amdgcn_sqrtf(x.v)); }
__device__ __forceinline__ F32 m_sin(F32 x) { return mkf(sinf(x.v)); }
__device__ __forceinline__ F32 m_cos(F32 x) { return mkf(cosf(x.v)); }
__device__ __forceinline__ F32 m_tan(F32 x) { return mkf(tanf(x.v)); }
__device__ __forceinline__ F32 m_sinh(F32 x) { return mkf(sinhf(x.v)); }
__device__ __forceinline__ F32 m_cosh(F32 x) { return mkf(coshf(x.v)); }
__device__ __forceinline__ F32 m_log1p(F32 x) { return mkf(log1pf(x.v)); }
__device__ __forceinline__ F32 m_asin(F32 x) { return mkf(asinf(x.v)); }
__device__ __forceinline__ F32 m_atan2(F32 y, F32 x) { return mkf(atan2f(y.v, x.v)); }
__device__ __forceinline__ F32 m_atan(F32 x) { return mkf(atanf(x.v)); }
__device__ __forceinline__ F32 m_asinh(F32 x) { return mkf(asinhf(x.v)); }
__device__ __forceinline__ F32 m_abs(F32 x) { return mkf(fabsf(x.v)); }

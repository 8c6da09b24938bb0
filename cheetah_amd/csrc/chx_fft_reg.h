// chx_fft_reg.h — register-resident complex FFTs of 2, 4, 8 and 16 points for gfx950.
// A complex number is a 2-vector (re, im): on CDNA4 every complex add is ONE v_pk_add_f32, every multiplication by a constant
// twiddle one v_pk_mul_f32 + one v_pk_fma_f32, and a multiplication by -+i an operand swizzle. The 16-point transform is two
// layers of radix-4 butterflies (4 x 4 Cooley-Tukey): 64 packed adds + 8 non-trivial twiddles ≈ 100 VALU instructions per
// lane, a third of the radix-2 formulation it replaces. fp64 takes the same code path with scalar v_add_f64 / v_fma_f64.
// Host-callable as well (tests/host/fft_reg_check.hip pins the butterflies against a direct DFT without a GPU).
#pragma once
#include <hip/hip_runtime.h>

template <typename T> using vec2 = T __attribute__((ext_vector_type(2)));

namespace chx_fft {

template <typename T> __host__ __device__ __forceinline__ vec2<T> mul_mi(vec2<T> a) { return vec2<T>{a.y, -a.x}; }   // a * (-i)
template <typename T> __host__ __device__ __forceinline__ vec2<T> mul_pi(vec2<T> a) { return vec2<T>{-a.y, a.x}; }   // a * (+i)
// a * W_4 of the transform direction: -i forward, +i inverse
template <bool INV, typename T> __host__ __device__ __forceinline__ vec2<T> rot(vec2<T> a) { return INV ? mul_pi(a) : mul_mi(a); }
// a * (wr + i wi)
template <typename T> __host__ __device__ __forceinline__ vec2<T> cmul(vec2<T> a, T wr, T wi) {
    const vec2<T> r = vec2<T>{a.x, a.x} * vec2<T>{wr, wi};
    return __builtin_elementwise_fma(vec2<T>{a.y, a.y}, vec2<T>{-wi, wr}, r);
}
template <typename T> __host__ __device__ __forceinline__ vec2<T> cmul(vec2<T> a, vec2<T> w) { return cmul(a, w.x, w.y); }
// a * conj(w)
template <typename T> __host__ __device__ __forceinline__ vec2<T> cmul_conj(vec2<T> a, vec2<T> w) { return cmul(a, w.x, -w.y); }

template <bool INV, typename T>
__host__ __device__ __forceinline__ void dft4(vec2<T>& a0, vec2<T>& a1, vec2<T>& a2, vec2<T>& a3) {
    const vec2<T> t0 = a0 + a2, t1 = a0 - a2, t2 = a1 + a3, t3 = rot<INV>(a1 - a3);
    a0 = t0 + t2; a2 = t0 - t2; a1 = t1 + t3; a3 = t1 - t3;
}
// the same with a2 = a3 = 0 (zero-padded upper half of a line)
template <bool INV, typename T>
__host__ __device__ __forceinline__ void dft4_upper_zero(vec2<T>& a0, vec2<T>& a1, vec2<T>& a2, vec2<T>& a3) {
    const vec2<T> r = rot<INV>(a1), s = a0;
    a0 = s + a1; a2 = s - a1; a1 = s + r; a3 = s - r;
}

// cos / sin of 2 pi k / 16
#define CHX_C16_1 0.92387953251128673848
#define CHX_C16_2 0.70710678118654752440
#define CHX_C16_3 0.38268343236508977173

// x[k] <- sum_j x[j] exp(-+2 pi i j k / 16). UPPER_ZERO: x[8..15] are zero on entry (not read).
template <typename T, bool INV, bool UPPER_ZERO = false>
__host__ __device__ __forceinline__ void fft16(vec2<T> (&x)[16]) {
    constexpr double CS[10] = {1.0, CHX_C16_1, CHX_C16_2, CHX_C16_3, 0.0, -CHX_C16_3, -CHX_C16_2, -CHX_C16_1, -1.0, -CHX_C16_1};
    constexpr double SN[10] = {0.0, CHX_C16_3, CHX_C16_2, CHX_C16_1, 1.0, CHX_C16_1, CHX_C16_2, CHX_C16_3, 0.0, -CHX_C16_3};
    // layer 1: 4-point transforms over n1 of x[4 n1 + n2]; result y[n2][k1] left at x[4 k1 + n2]
#pragma unroll
    for (int n2 = 0; n2 < 4; ++n2) {
        if (UPPER_ZERO) dft4_upper_zero<INV>(x[n2], x[4 + n2], x[8 + n2], x[12 + n2]);
        else dft4<INV>(x[n2], x[4 + n2], x[8 + n2], x[12 + n2]);
    }
    // twiddles W_16^(n2 k1)
#pragma unroll
    for (int k1 = 1; k1 < 4; ++k1) {
#pragma unroll
        for (int n2 = 1; n2 < 4; ++n2) {
            const int e = n2 * k1;
            if (e == 4) x[4 * k1 + n2] = rot<INV>(x[4 * k1 + n2]);
            else x[4 * k1 + n2] = cmul(x[4 * k1 + n2], (T)CS[e], (T)(INV ? SN[e] : -SN[e]));
        }
    }
    // layer 2: 4-point transforms over n2; X[k1 + 4 k2] left at x[4 k1 + k2]
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) dft4<INV>(x[4 * k1], x[4 * k1 + 1], x[4 * k1 + 2], x[4 * k1 + 3]);
    // natural order (register renaming: no instructions)
#pragma unroll
    for (int a = 0; a < 4; ++a) {
#pragma unroll
        for (int b = a + 1; b < 4; ++b) {
            const vec2<T> t = x[4 * a + b];
            x[4 * a + b] = x[4 * b + a];
            x[4 * b + a] = t;
        }
    }
}

template <typename T, bool INV>
__host__ __device__ __forceinline__ void fft8(vec2<T> (&x)[8]) {
    vec2<T> e0 = x[0], e1 = x[2], e2 = x[4], e3 = x[6], o0 = x[1], o1 = x[3], o2 = x[5], o3 = x[7];
    dft4<INV>(e0, e1, e2, e3);
    dft4<INV>(o0, o1, o2, o3);
    constexpr T h = (T)CHX_C16_2;
    o1 = cmul(o1, h, INV ? h : -h);       // W_8
    o2 = rot<INV>(o2);                    // W_8^2
    o3 = cmul(o3, -h, INV ? h : -h);      // W_8^3
    x[0] = e0 + o0; x[4] = e0 - o0;
    x[1] = e1 + o1; x[5] = e1 - o1;
    x[2] = e2 + o2; x[6] = e2 - o2;
    x[3] = e3 + o3; x[7] = e3 - o3;
}

// R-point transform, R in {2, 4, 8, 16}
template <typename T, int R, bool INV>
__host__ __device__ __forceinline__ void fft_small(vec2<T> (&x)[R]) {
    if constexpr (R == 16) {
        fft16<T, INV>(x);
    } else if constexpr (R == 8) {
        fft8<T, INV>(x);
    } else if constexpr (R == 4) {
        dft4<INV>(x[0], x[1], x[2], x[3]);
    } else {
        static_assert(R == 2, "fft_small: R must be 2, 4, 8 or 16");
        const vec2<T> a = x[0];
        x[0] = a + x[1];
        x[1] = a - x[1];
    }
}

}  // namespace chx_fft

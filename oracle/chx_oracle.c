/* chx_oracle.c — CPU restatement of the reference hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file restates, in plain C, the arithmetic of desy-ml/cheetah's
 * `Segment.track(ParticleBeam)` path so that the HIP kernels of libchx can be checked
 * against it on machines where the (Python) reference is not present.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the product
 * (cheetah_amd) never does.  Parity of this oracle against the real reference is pinned
 * by tests/golden/ (fixtures generated in the build container by importing
 * /root/reference; generator: tests/golden/generate_golden.py) — see
 * tests/test_oracle_golden.py.
 *
 * Every function names the reference code it follows (paths under /root/reference/cheetah).
 * Unlike the device kernels (which use real-branch series near singular points) the
 * formulas here follow the reference expression by expression: closed forms with the
 * reference's explicit `where(x != 0, limit)` special cases.
 *
 * dtype: 0 = float32, 1 = float64 (element type of the `void*` buffers).  Unless noted the
 * arithmetic is carried out in double and rounded once on store; the cloud-in-cell /
 * histogram INDEX arithmetic is carried out in the working dtype, operation by operation,
 * exactly as the reference's tensor ops do (bit-exact indices).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifndef CHXO_API
#define CHXO_API __attribute__((visibility("default")))
#endif

static const double kC = 299792458.0;             /* scipy.constants.speed_of_light */
static const double kE = 1.602176634e-19;         /* scipy.constants.elementary_charge */
static const double kEvToKg = 1.7826619216278975e-36; /* "electron volt-kilogram relationship" */
static const double kPi = 3.14159265358979323846;

static inline double ld(const void* p, int dtype, int64_t i) {
    return dtype == 0 ? (double)((const float*)p)[i] : ((const double*)p)[i];
}
static inline void st(void* p, int dtype, int64_t i, double v) {
    if (dtype == 0) ((float*)p)[i] = (float)v;
    else ((double*)p)[i] = v;
}

/* ---- special functions --------------------------------------------------------------- */
/* cos(sqrt(k2) L) and sinc(sqrt(k2) L / pi) * L for real k2 of either sign: the real part of
 * the reference's complex evaluation (track_methods.py:44-49). */
static void cs_pair(double k2, double L, double* c, double* s) {
    if (k2 > 0) {
        const double k = sqrt(k2);
        *c = cos(k * L);
        *s = (k * L != 0.0) ? sin(k * L) / (k * L) * L : L;
    } else if (k2 < 0) {
        const double k = sqrt(-k2);
        *c = cosh(k * L);
        *s = (k * L != 0.0) ? sinh(k * L) / (k * L) * L : L;
    } else {
        *c = 1.0;
        *s = L;
    }
}
/* sinc(z/pi) for z = sqrt(u) (complex if u<0), real part */
static double sinc_sqrt(double u) {
    if (u > 0) { const double a = sqrt(u); return sin(a) / a; }
    if (u < 0) { const double a = sqrt(-u); return sinh(a) / a; }
    return 1.0;
}
/* utils/autograd.py:125-128  si1mdiv(x) = (1 - si(sqrt x)) / x, 1/6 at 0 */
static double si1mdiv(double x) { return x != 0.0 ? (1.0 - sinc_sqrt(x)) / x : 1.0 / 6.0; }
/* utils/autograd.py:94-95  log1pdiv(x) = log1p(x)/x, 1 at 0 */
static double log1pdiv(double x) { return x != 0.0 ? log1p(x) / x : 1.0; }

/* ---- 7x7 helpers ------------------------------------------------------------------------ */
static void eye7(double* R) { memset(R, 0, 49 * sizeof(double)); for (int i = 0; i < 7; ++i) R[i * 8] = 1.0; }
static void mm7(const double* A, const double* B, double* C) {
    double T[49];
    for (int i = 0; i < 7; ++i)
        for (int j = 0; j < 7; ++j) {
            double s = 0.0;
            for (int k = 0; k < 7; ++k) s += A[i * 7 + k] * B[k * 7 + j];
            T[i * 7 + j] = s;
        }
    memcpy(C, T, sizeof(T));
}
static void rel_factors(double energy, double mass, double* gamma, double* ig2, double* beta) {
    *gamma = energy / mass;              /* utils/physics.py:15-17 */
    *ig2 = 1.0 / ((*gamma) * (*gamma));
    *beta = sqrt(1.0 - *ig2);
}

/* track_methods.py:284-299 */
static void drift_matrix(double L, double energy, double mass, double* R) {
    double g, ig2, b;
    rel_factors(energy, mass, &g, &ig2, &b);
    eye7(R);
    R[0 * 7 + 1] = L;
    R[2 * 7 + 3] = L;
    R[4 * 7 + 5] = -L / (b * b) * ig2;
}

/* track_methods.py:17-77 */
static void base_rmatrix(double L, double k1, double hx, double energy, double mass, double* R) {
    double g, ig2, beta;
    rel_factors(energy, mass, &g, &ig2, &beta);
    const double kx2 = k1 + hx * hx, ky2 = -k1;
    double cx, sx, cy, sy;
    cs_pair(kx2, L, &cx, &sx);
    cs_pair(ky2, L, &cy, &sy);
    /* r = sinc(0.5 kx L / pi); dx = hx * 0.5 * L^2 * r^2  (track_methods.py:51-52) */
    const double r = sinc_sqrt(0.25 * kx2 * L * L);
    const double dx = hx * 0.5 * L * L * r * r;
    const double r56 = hx * hx * L * L * L * si1mdiv(kx2 * L * L) / (beta * beta) - L / (beta * beta) * ig2;
    eye7(R);
    R[0] = cx; R[1] = sx; R[5] = dx / beta;
    R[7] = -kx2 * sx; R[8] = cx; R[12] = sx * hx / beta;
    R[2 * 7 + 2] = cy; R[2 * 7 + 3] = sy;
    R[3 * 7 + 2] = -ky2 * sy; R[3 * 7 + 3] = cy;
    R[4 * 7 + 0] = sx * hx / beta; R[4 * 7 + 1] = dx / beta; R[4 * 7 + 5] = r56;
}

/* track_methods.py:302-323 */
static void rotation_matrix(double angle, double* R) {
    const double cs = cos(angle), sn = sin(angle);
    eye7(R);
    R[0 * 7 + 0] = cs; R[0 * 7 + 2] = sn; R[1 * 7 + 1] = cs; R[1 * 7 + 3] = sn;
    R[2 * 7 + 0] = -sn; R[2 * 7 + 2] = cs; R[3 * 7 + 1] = -sn; R[3 * 7 + 3] = cs;
}
static void transpose7(const double* A, double* At) {
    double T[49];
    for (int i = 0; i < 7; ++i) for (int j = 0; j < 7; ++j) T[i * 7 + j] = A[j * 7 + i];
    memcpy(At, T, sizeof(T));
}

/* quadrupole.py:93-110 + track_methods.py:345-382 */
static void quadrupole_map(const double* p, double energy, double mass, double* R) {
    const double L = p[0], k1 = p[1], tilt = p[2], mx = p[3], my = p[4];
    double base[49], entry[49], exitm[49], tmp[49];
    base_rmatrix(L, k1, 0.0, energy, mass, base);
    rotation_matrix(tilt, entry);
    transpose7(entry, exitm);
    const double cs = cos(tilt), sn = sin(tilt);
    entry[0 * 7 + 6] = -mx * cs - my * sn;
    entry[2 * 7 + 6] = mx * sn - my * cs;
    exitm[0 * 7 + 6] = mx;
    exitm[2 * 7 + 6] = my;
    mm7(base, entry, tmp);
    mm7(exitm, tmp, R);
}

/* dipole.py:372-394, 430-466 ; hx = angle/length (dipole.py:133-135) */
static void dipole_map(const double* p, double energy, double mass, double* R) {
    const double L = p[0], angle = p[1], k1 = p[2], e1 = p[3], e2 = p[4], tilt = p[5], fint = p[6],
                 fint_exit = p[7], gap = p[8];
    const double hx = angle / L;
    double base[49], enter[49], exitm[49], rot[49], rotT[49], t[49];
    base_rmatrix(L, k1, hx, energy, mass, base);
    {
        const double sec = 1.0 / cos(e1);
        const double phi = fint * hx * gap * sec * (1.0 + sin(e1) * sin(e1));
        eye7(enter);
        enter[1 * 7 + 0] = hx * tan(e1);
        enter[3 * 7 + 2] = -hx * tan(e1 - phi);
    }
    {
        const double sec = 1.0 / cos(e2);
        const double phi = fint_exit * hx * gap * sec * (1.0 + sin(e2) * sin(e2)); /* dipole.py:453-459 uses self.gap */
        eye7(exitm);
        exitm[1 * 7 + 0] = hx * tan(e2);
        exitm[3 * 7 + 2] = -hx * tan(e2 - phi);
    }
    mm7(base, enter, t);       /* R_exit @ R @ R_enter */
    mm7(exitm, t, t);
    rotation_matrix(tilt, rot);
    transpose7(rot, rotT);
    mm7(t, rot, t);            /* rotation.mT @ R @ rotation */
    mm7(rotT, t, R);
}

/* cavity.py:253-358 */
static void cavity_map(const double* p, double energy, double mass, double nq, int standing, double* R) {
    const double L = p[0], V = p[1], phi = p[2] * (kPi / 180.0), freq = p[3];
    const double veff = -V * nq;
    const double dEn = veff * cos(phi);
    const double Ei = energy / mass, dE = dEn / mass, Ef = Ei + dE, Ep = dE / L;
    const double k = 2.0 * kPi * freq / kC;
    double r11, r12, r21, r22, r55, r56, r65, r66;
    if (standing) {
        const double alpha = sqrt(0.125) * veff / energy * log1pdiv(dEn / energy);
        const double beta0 = sqrt(1.0 - 1.0 / (Ei * Ei)), beta1 = sqrt(1.0 - 1.0 / (Ef * Ef));
        r11 = cos(alpha) - sqrt(2.0) * cos(phi) * sin(alpha);
        r12 = (alpha != 0.0 ? sin(alpha) / alpha : 1.0) * log1pdiv(dEn / energy) * L;
        r21 = -(veff / ((energy + dEn) * sqrt(2.0) * L) * (0.5 + cos(phi) * cos(phi)) * sin(alpha));
        r22 = Ei / Ef * (cos(alpha) + sqrt(2.0) * cos(phi) * sin(alpha));
        r55 = 1.0 + ((dE != 0.0) ? k * L * beta0 * tan(phi) * (Ei * Ef * (beta0 * beta1 - 1.0) + 1.0) / (beta1 * Ef * dE) : 0.0);
        r56 = -L / (Ef * Ef * Ei * beta1) * (Ef + Ei) / (beta1 + beta0);
        r65 = k * sin(phi) * veff / (beta1 * (energy + dEn));
        r66 = Ei / Ef * beta0 / beta1;
    } else {
        /* M = M_f_exit @ M_body @ M_f_entry  (cavity.py:316-326) */
        const double body01 = L * log1pdiv(dE / Ei), body11 = Ei / Ef;
        const double fe = -Ep / (2.0 * Ei), fx = Ep / (2.0 * Ef);
        const double m00 = 1.0 + body01 * fe, m01 = body01, m10 = body11 * fe, m11 = body11;
        r11 = m00; r12 = m01; r21 = fx * m00 + m10; r22 = fx * m01 + m11;
        r55 = 1.0; r56 = 0.0;
        r65 = k * sin(phi) * veff / (energy + dEn);
        r66 = r22;
    }
    eye7(R);
    R[0] = r11; R[1] = r12; R[7] = r21; R[8] = r22;
    R[2 * 7 + 2] = r11; R[2 * 7 + 3] = r12; R[3 * 7 + 2] = r21; R[3 * 7 + 3] = r22;
    R[4 * 7 + 4] = r55; R[4 * 7 + 5] = r56; R[5 * 7 + 4] = r65; R[5 * 7 + 5] = r66;
}

/* solenoid.py:75-116 */
static void solenoid_map(const double* p, double energy, double mass, double* R) {
    const double L = p[0], k = p[1], mx = p[2], my = p[3];
    const double gamma = energy / mass;
    const double c = cos(L * k), s = sin(L * k);
    const double s_k = (L * k != 0.0 ? sin(L * k) / (L * k) : 1.0) * L; /* sinc(L k / pi) * L */
    double body[49], entry[49], exitm[49], t[49];
    eye7(body);
    body[0] = c * c;       body[1] = c * s_k;   body[2] = s * c;        body[3] = s * s_k;
    body[7] = -k * s * c;  body[8] = c * c;     body[9] = -k * s * s;   body[10] = s * c;
    body[14] = -s * c;     body[15] = -s * s_k; body[16] = c * c;       body[17] = c * s_k;
    body[21] = k * s * s;  body[22] = -s * c;   body[23] = -k * s * c;  body[24] = c * c;
    body[4 * 7 + 5] = L / (1.0 - gamma * gamma);
    eye7(entry); eye7(exitm);               /* track_methods.py:326-342 */
    entry[0 * 7 + 6] = -mx; entry[2 * 7 + 6] = -my;
    exitm[0 * 7 + 6] = mx;  exitm[2 * 7 + 6] = my;
    mm7(body, entry, t);
    mm7(exitm, t, R);
}

/* undulator.py:79-125 */
static void undulator_map(const double* p, double energy, double mass, double* R) {
    const double L = p[0], kx = p[1], ky = p[2], period = p[3];
    double g, ig2, beta;
    rel_factors(energy, mass, &g, &ig2, &beta);
    eye7(R);
    R[4 * 7 + 5] = -L * ig2 * (1.0 / (beta * beta) + 0.5 * (kx * kx + ky * ky));
    const double sf = period > 0.0 ? sqrt(2.0) * kPi / (period * g * beta) : 0.0;
    const double wx = sf * kx, wy = sf * ky;
    R[2 * 7 + 2] = cos(wx * L); R[2 * 7 + 3] = (wx * L != 0.0 ? sin(wx * L) / (wx * L) : 1.0) * L;
    R[3 * 7 + 2] = -sin(wx * L) * wx; R[3 * 7 + 3] = cos(wx * L);
    R[0] = cos(wy * L); R[1] = (wy * L != 0.0 ? sin(wy * L) / (wy * L) : 1.0) * L;
    R[7] = -sin(wy * L) * wy; R[8] = cos(wy * L);
}

static int kind_np(int kind) {
    static const int np[11] = {0, 1, 5, 9, 2, 2, 3, 4, 4, 4, 4};
    return (kind >= 0 && kind < 11) ? np[kind] : -1;
}

/* params/energy/R_out are double here (the oracle is dtype-agnostic for the tiny maps) */
CHXO_API int chxo_build_rmatrix(int kind, const double* params, const double* energy, double mass,
                                double nq, int64_t B, int64_t Bp, int64_t Be, double* R_out) {
    const int P = kind_np(kind);
    if (P < 0) return -1;
    for (int64_t b = 0; b < B; ++b) {
        const double* p = params + (Bp == 1 ? 0 : b) * P;
        const double en = energy[Be == 1 ? 0 : b];
        double* R = R_out + b * 49;
        switch (kind) {
            case 0: eye7(R); break;
            case 1: drift_matrix(p[0], en, mass, R); break;
            case 2: quadrupole_map(p, en, mass, R); break;
            case 3: dipole_map(p, en, mass, R); break;
            case 4: drift_matrix(p[0], en, mass, R); R[1 * 7 + 6] = p[1]; break;              /* horizontal_corrector.py:72-76 */
            case 5: drift_matrix(p[0], en, mass, R); R[3 * 7 + 6] = p[1]; break;              /* vertical_corrector.py:73-76 */
            case 6: drift_matrix(p[0], en, mass, R); R[1 * 7 + 6] = p[1]; R[3 * 7 + 6] = p[2]; break; /* combined_corrector.py:92-96 */
            case 7: cavity_map(p, en, mass, nq, 1, R); break;
            case 8: cavity_map(p, en, mass, nq, 0, R); break;
            case 9: solenoid_map(p, en, mass, R); break;
            case 10: undulator_map(p, en, mass, R); break;
        }
    }
    return 0;
}

/* segment.py:534-543: tm = R_e @ tm, starting from eye(7). R is [E][BR][49], BR in {1,B}. */
CHXO_API int chxo_compose(const double* R, int64_t E, int64_t B, int64_t BR, double* R_out) {
    for (int64_t b = 0; b < B; ++b) {
        double tm[49];
        eye7(tm);
        for (int64_t e = 0; e < E; ++e) mm7(R + (e * BR + (BR == 1 ? 0 : b)) * 49, tm, tm);
        memcpy(R_out + b * 49, tm, sizeof(tm));
    }
    return 0;
}

/* element.py:182: new_particles = particles @ tm.mT.
 * mode 0: accumulate in double, round once (the "truth" the tolerances are stated against);
 * mode 1: accumulate in the working dtype as the fused-multiply-add chain j = 0..6 that the
 *         device kernels use (bit-exact comparison of the fp32/fp64 kernels). */
CHXO_API int chxo_apply(const void* x_in, const void* R, void* x_out, int64_t B, int64_t Bx, int64_t BR,
                        int64_t N, int dtype, int mode) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int64_t b = 0; b < B; ++b)
        for (int64_t n = 0; n < N; ++n) {
            const int64_t xi = ((Bx == 1 ? 0 : b) * N + n) * 7, xo = (b * N + n) * 7;
            const int64_t rb = (BR == 1 ? 0 : b) * 49;
            if (mode == 1 && dtype == 0) {
                const float* x = (const float*)x_in + xi;
                const float* r = (const float*)R + rb;
                float y[7];
                for (int i = 0; i < 7; ++i) {
                    float acc = r[i * 7] * x[0];
                    for (int j = 1; j < 7; ++j) acc = fmaf(r[i * 7 + j], x[j], acc);
                    y[i] = acc;
                }
                memcpy((float*)x_out + xo, y, sizeof(y));
            } else if (mode == 1) {
                const double* x = (const double*)x_in + xi;
                const double* r = (const double*)R + rb;
                double y[7];
                for (int i = 0; i < 7; ++i) {
                    double acc = r[i * 7] * x[0];
                    for (int j = 1; j < 7; ++j) acc = fma(r[i * 7 + j], x[j], acc);
                    y[i] = acc;
                }
                memcpy((double*)x_out + xo, y, sizeof(y));
            } else {
                double x[7], y[7];
                for (int j = 0; j < 7; ++j) x[j] = ld(x_in, dtype, xi + j);
                for (int i = 0; i < 7; ++i) {
                    double acc = 0.0;
                    for (int j = 0; j < 7; ++j) acc += ld(R, dtype, rb + i * 7 + j) * x[j];
                    y[i] = acc;
                }
                for (int i = 0; i < 7; ++i) st(x_out, dtype, xo + i, y[i]);
            }
        }
    return 0;
}

/* segment.py:571-572 `for e in elements: beam = e.track(beam)` for E linear elements, fp32 or fp64
 * fma-chain arithmetic (mode 1 of chxo_apply), ping-ponging between two caller buffers so that no
 * allocation or first-touch page fault falls into the timed region of the CPU baseline.
 * R is [E][49] of the working dtype; the result ends in `buf_a` if E is even, else in `buf_b`...
 * to keep it simple the function returns 0 and ALWAYS leaves the result in buf_out. */
CHXO_API int chxo_track_elementwise(const void* x_in, const void* R, void* buf_out, void* buf_tmp, int64_t E,
                                    int64_t N, int dtype) {
    const void* src = x_in;
    for (int64_t e = 0; e < E; ++e) {
        void* dst = (((E - 1 - e) & 1) == 0) ? buf_out : buf_tmp;
        const size_t esz = dtype == 0 ? 4 : 8;
        chxo_apply(src, (const char*)R + (size_t)e * 49 * esz, dst, 1, 1, 1, N, dtype, 1);
        src = dst;
    }
    return 0;
}

/* cavity.py:100-226: per-batch coefficients [a, b, k*beta0, phi, cos phi, T566, T556, T555] + E' */
CHXO_API int chxo_cavity_coeffs(const double* params, const double* energy, double mass, double nq,
                                int64_t B, int64_t Bp, int64_t Be, double* coeffs, double* energy_out) {
    int gain = 0;
    for (int64_t b = 0; b < B; ++b) {
        const double* p = params + (Bp == 1 ? 0 : b) * 4;
        if (p[1] * cos(p[2] * (kPi / 180.0)) * nq * -1.0 > 0.0) gain = 1; /* cavity.py:113-115,157 */
    }
    for (int64_t b = 0; b < B; ++b) {
        const double* p = params + (Bp == 1 ? 0 : b) * 4;
        const double L = p[0], V = p[1], phi = p[2] * (kPi / 180.0), freq = p[3];
        const double E0 = energy[Be == 1 ? 0 : b];
        double g0, ig2, b0;
        rel_factors(E0, mass, &g0, &ig2, &b0);
        const double dEn = V * cos(phi) * nq * -1.0;
        const double E1 = E0 + dEn;
        double g1, ig21, b1;
        rel_factors(E1, mass, &g1, &ig21, &b1);
        const double k = 2.0 * kPi * freq / kC;
        double T566 = 1.5 * L * ig2 / pow(b0, 3), T556 = 0.0, T555 = 0.0;
        if (gain) {
            const double dg = V / mass;
            T566 = L * (pow(b0, 3) * pow(g0, 3) - pow(b1, 3) * pow(g1, 3)) /
                   (2.0 * b0 * pow(b1, 3) * g0 * (g0 - g1) * pow(g1, 3));
            T556 = b0 * k * L * dg * g0 * (pow(b1, 3) * pow(g1, 3) + b0 * (g0 - pow(g1, 3))) * sin(phi) /
                   (pow(b1, 3) * pow(g1, 3) * (g0 - g1) * (g0 - g1));
            T555 = b0 * b0 * k * k * L * dg / 2.0 *
                   (dg * (2.0 * g0 * pow(g1, 3) * (b0 * pow(b1, 3) - 1.0) + g0 * g0 + 3.0 * g1 * g1 - 2.0) /
                        (pow(b1, 3) * pow(g1, 3) * pow(g0 - g1, 3)) * sin(phi) * sin(phi) -
                    (g1 * g0 * (b1 * b0 - 1.0) + 1.0) / (b1 * g1 * (g0 - g1) * (g0 - g1)) * cos(phi));
        }
        double* c = coeffs + b * 8;
        c[0] = E0 * b0 / (E1 * b1);
        c[1] = V * b0 / (E1 * b1);
        c[2] = b0 * k;
        c[3] = phi;
        c[4] = cos(phi);
        c[5] = T566; c[6] = T556; c[7] = T555;
        energy_out[b] = E1;
    }
    return 0;
}

/* cavity.py:112,135-151,220-226 */
CHXO_API int chxo_cavity_track(const void* x_in, const void* R, const double* coeffs, void* x_out,
                               int64_t B, int64_t Bx, int64_t N, int dtype) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int64_t b = 0; b < B; ++b)
        for (int64_t n = 0; n < N; ++n) {
            const int64_t xi = ((Bx == 1 ? 0 : b) * N + n) * 7, xo = (b * N + n) * 7;
            const double* c = coeffs + b * 8;
            double x[7], y[7];
            for (int j = 0; j < 7; ++j) x[j] = ld(x_in, dtype, xi + j);
            for (int i = 0; i < 7; ++i) {
                double acc = 0.0;
                for (int j = 0; j < 7; ++j) acc += ld(R, dtype, b * 49 + i * 7 + j) * x[j];
                y[i] = acc;
            }
            y[5] = x[5] * c[0] + c[1] * (cos(-x[4] * c[2] + c[3]) - c[4]);
            y[4] = y[4] + (c[5] * x[5] * x[5] + c[6] * x[4] * x[5] + c[7] * x[4] * x[4]);
            for (int i = 0; i < 7; ++i) st(x_out, dtype, xo + i, y[i]);
        }
    return 0;
}

/* particle_beam.py:1699-1717 + statistics.py:4-48: out[b] = {W, W2, mu[6], cov upper-tri[21]} */
CHXO_API int chxo_moments(const void* x, const void* w, int64_t B, int64_t Bx, int64_t Bw, int64_t N,
                          int dtype, double* out) {
    for (int64_t b = 0; b < B; ++b) {
        const int64_t xr = (Bx == 1 ? 0 : b) * N, wr = (Bw == 1 ? 0 : b) * N;
        double W = 0.0, W2 = 0.0, s[6] = {0};
        for (int64_t n = 0; n < N; ++n) {
            const double wv = w ? ld(w, dtype, wr + n) : 1.0;
            W += wv; W2 += wv * wv;
            for (int j = 0; j < 6; ++j) s[j] += wv * ld(x, dtype, (xr + n) * 7 + j);
        }
        double mu[6], m2[21] = {0};
        for (int j = 0; j < 6; ++j) mu[j] = s[j] / W;
        for (int64_t n = 0; n < N; ++n) {
            const double wv = w ? ld(w, dtype, wr + n) : 1.0;
            double d[6];
            for (int j = 0; j < 6; ++j) d[j] = ld(x, dtype, (xr + n) * 7 + j) - mu[j];
            int k = 0;
            for (int i = 0; i < 6; ++i) for (int j = i; j < 6; ++j) m2[k++] += wv * d[i] * d[j];
        }
        double* o = out + b * 29;
        o[0] = W; o[1] = W2;
        for (int j = 0; j < 6; ++j) o[2 + j] = mu[j];
        const double cf = W - W2 / W;
        for (int k = 0; k < 21; ++k) o[8 + k] = m2[k] / cf;
    }
    return 0;
}

/* ---- cloud in cell (utils/cloud_in_cell.py) and histogram, dtype-faithful ------------------ */
typedef struct chxo_cic_args {
    int32_t ndim; int32_t cols[3]; int32_t bins[3];
    int64_t B, Bx, Bq, Bs, Be, Bsc, Bsh, N;
    int32_t dtype; int32_t abs_charge;
    const void* x; const void* charge; const void* survival; const void* extent;
    const void* scale; const void* shift;
} chxo_cic_args;

#define DEFINE_CIC(T, SUF, FLOOR, FABS)                                                               \
    static int cic_locate_##SUF(const chxo_cic_args* a, int64_t b, int64_t n, int64_t* idx, T* frac) {  \
        int inside = 1;                                                                                \
        const T* x = (const T*)a->x;                                                                  \
        const T* ext = (const T*)a->extent + (a->Be == 1 ? 0 : b) * a->ndim * 2;                       \
        const int64_t row = (a->Bx == 1 ? 0 : b) * a->N + n;                                           \
        for (int d = 0; d < a->ndim; ++d) {                                                            \
            volatile T v = x[row * 7 + a->cols[d]];                                                    \
            if (a->scale) v = v * ((const T*)a->scale)[(a->Bsc == 1 ? 0 : b) * a->ndim + d];           \
            if (a->shift) v = v - ((const T*)a->shift)[(a->Bsh == 1 ? 0 : b) * a->ndim + d];           \
            const T l = ext[d * 2], r = ext[d * 2 + 1];                                                \
            inside = inside && (v >= l) && (v <= r);           /* cloud_in_cell.py:150-156 */          \
            volatile T bw = (r - l) / (T)a->bins[d];           /* :158 */                              \
            volatile T q = (v - l) / bw;                                                               \
            volatile T pb = q - (T)0.5;                        /* :159-165 */                          \
            T fl = FLOOR(pb);                                                                          \
            if (fl > (T)4.0e18) fl = (T)4.0e18;                                                        \
            if (fl < (T)-4.0e18) fl = (T)-4.0e18;                                                      \
            const int64_t i = (int64_t)fl;                     /* floor().long() :167-169 */           \
            idx[d] = i;                                                                                \
            frac[d] = pb - (T)i;                               /* :170-172 */                          \
        }                                                                                              \
        return inside;                                                                                 \
    }                                                                                                  \
    static T cic_charge_##SUF(const chxo_cic_args* a, int64_t b, int64_t n) {                          \
        volatile T c = a->charge ? ((const T*)a->charge)[(a->Bq == 1 ? 0 : b) * a->N + n] : (T)1;       \
        if (a->abs_charge) c = FABS(c);                                                                \
        if (a->survival) c = c * ((const T*)a->survival)[(a->Bs == 1 ? 0 : b) * a->N + n];            \
        return c;                                                                                      \
    }                                                                                                  \
    /* accumulates in T, corner-major / particle-minor like the 2^d scatter_add_ passes */             \
    static void cic_deposit_##SUF(const chxo_cic_args* a, T* grid) {                                   \
        int64_t total = 1, stride[3] = {0, 0, 0};                                                      \
        for (int d = a->ndim - 1; d >= 0; --d) { stride[d] = total; total *= a->bins[d]; }             \
        const int nc = 1 << a->ndim;                                                                   \
        for (int64_t b = 0; b < a->B; ++b) {                                                           \
            T* g = grid + b * total;                                                                   \
            for (int corner = 0; corner < nc; ++corner) {                                              \
                int o[3];                                                                              \
                if (a->ndim == 2) { o[0] = corner & 1; o[1] = (corner >> 1) & 1; o[2] = 0; }            \
                else if (a->ndim == 3) { o[0] = (corner >> 2) & 1; o[1] = (corner >> 1) & 1; o[2] = corner & 1; } \
                else { o[0] = corner; o[1] = 0; o[2] = 0; }                                            \
                for (int64_t n = 0; n < a->N; ++n) {                                                   \
                    int64_t idx[3]; T f[3];                                                            \
                    const int inside = cic_locate_##SUF(a, b, n, idx, f);                              \
                    volatile T c = cic_charge_##SUF(a, b, n) * (T)inside;                              \
                    int64_t off = 0; T wf[3] = {1, 1, 1};                                              \
                    for (int d = 0; d < a->ndim; ++d) {                                                \
                        const int64_t id = idx[d] + o[d];                                              \
                        const int ok = (id >= 0) && (id < a->bins[d]);                                 \
                        const int64_t ic = id < 0 ? 0 : (id > a->bins[d] - 1 ? a->bins[d] - 1 : id);   \
                        off += ic * stride[d];                                                         \
                        volatile T w0 = o[d] ? f[d] : ((T)1.0 - f[d]);                                 \
                        wf[d] = w0 * (T)ok;                                                            \
                    }                                                                                  \
                    volatile T src;                                                                    \
                    if (a->ndim == 1) src = c * wf[0];                                                 \
                    else if (a->ndim == 2) { volatile T t = c * wf[0]; src = t * wf[1]; }              \
                    else { volatile T t = wf[0] * wf[1]; volatile T u = t * wf[2]; src = c * u; }      \
                    volatile T acc = g[off] + src;                                                     \
                    g[off] = acc;                                                                      \
                }                                                                                      \
            }                                                                                          \
        }                                                                                              \
    }                                                                                                  \
    static int hist_bin_##SUF(const T* edges, int nbins, T v) {                                        \
        if (!(v >= edges[0]) || !(v <= edges[nbins])) return -1;                                       \
        int lo = 0, hi = nbins + 1;                                                                    \
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (edges[mid] > v) hi = mid; else lo = mid + 1; } \
        int pos = lo - 1;                                                                              \
        if (pos == nbins) pos -= 1;                                                                    \
        return pos;                                                                                    \
    }

DEFINE_CIC(float, f32, floorf, fabsf)
DEFINE_CIC(double, f64, floor, fabs)

CHXO_API int chxo_cic_deposit(const chxo_cic_args* a, void* grid) {
    if (a->dtype == 0) cic_deposit_f32(a, (float*)grid);
    else cic_deposit_f64(a, (double*)grid);
    return 0;
}

CHXO_API int chxo_cic_indices(const chxo_cic_args* a, int32_t* idx_out, void* frac_out) {
    for (int64_t b = 0; b < a->B; ++b)
        for (int64_t n = 0; n < a->N; ++n) {
            int64_t idx[3];
            if (a->dtype == 0) {
                float f[3];
                cic_locate_f32(a, b, n, idx, f);
                for (int d = 0; d < a->ndim; ++d) ((float*)frac_out)[(b * a->N + n) * a->ndim + d] = f[d];
            } else {
                double f[3];
                cic_locate_f64(a, b, n, idx, f);
                for (int d = 0; d < a->ndim; ++d) ((double*)frac_out)[(b * a->N + n) * a->ndim + d] = f[d];
            }
            for (int d = 0; d < a->ndim; ++d) {
                int64_t i = idx[d];
                if (i > 2147483647LL) i = 2147483647LL;
                if (i < -2147483647LL) i = -2147483647LL;
                idx_out[(b * a->N + n) * a->ndim + d] = (int32_t)i;
            }
        }
    return 0;
}

/* screen.py:292-311 -> torch.histogramdd with explicit edges (ATen HistogramKernel.cpp:
 * skip if elt < leftmost || rightmost < elt; pos = upper_bound(edges, elt) - 1; last edge inclusive).
 * x = col 0, y = col 2 of the 7-vector minus the screen misalignment (screen.py:200-212);
 * weight = |q| * survival; image[b][jy][jx]. ij_out (optional) gets (jx, jy) or (-1,-1). */
CHXO_API int chxo_hist2d(const void* x, const void* charge, const void* survival, const void* shift,
                         const void* edges_x, const void* edges_y, int64_t B, int64_t Bx, int64_t Bq,
                         int64_t Bs, int64_t Bsh, int64_t N, int nx, int ny, int dtype, void* image,
                         int32_t* ij_out) {
    for (int64_t b = 0; b < B; ++b)
        for (int64_t n = 0; n < N; ++n) {
            const int64_t row = (Bx == 1 ? 0 : b) * N + n;
            int jx, jy;
            double wgt;
            if (dtype == 0) {
                volatile float vx = ((const float*)x)[row * 7], vy = ((const float*)x)[row * 7 + 2];
                if (shift) { vx = vx - ((const float*)shift)[(Bsh == 1 ? 0 : b) * 2]; vy = vy - ((const float*)shift)[(Bsh == 1 ? 0 : b) * 2 + 1]; }
                jx = hist_bin_f32((const float*)edges_x, nx, vx);
                jy = hist_bin_f32((const float*)edges_y, ny, vy);
                volatile float c = charge ? fabsf(((const float*)charge)[(Bq == 1 ? 0 : b) * N + n]) : 1.0f;
                if (survival) c = c * ((const float*)survival)[(Bs == 1 ? 0 : b) * N + n];
                wgt = c;
            } else {
                volatile double vx = ((const double*)x)[row * 7], vy = ((const double*)x)[row * 7 + 2];
                if (shift) { vx = vx - ((const double*)shift)[(Bsh == 1 ? 0 : b) * 2]; vy = vy - ((const double*)shift)[(Bsh == 1 ? 0 : b) * 2 + 1]; }
                jx = hist_bin_f64((const double*)edges_x, nx, vx);
                jy = hist_bin_f64((const double*)edges_y, ny, vy);
                volatile double c = charge ? fabs(((const double*)charge)[(Bq == 1 ? 0 : b) * N + n]) : 1.0;
                if (survival) c = c * ((const double*)survival)[(Bs == 1 ? 0 : b) * N + n];
                wgt = c;
            }
            if (jx < 0 || jy < 0) { jx = -1; jy = -1; }
            if (ij_out) { ij_out[(b * N + n) * 2] = jx; ij_out[(b * N + n) * 2 + 1] = jy; }
            if (image && jx >= 0) {
                const int64_t o = (b * ny + jy) * (int64_t)nx + jx;
                if (dtype == 0) { volatile float s = ((float*)image)[o] + (float)wgt; ((float*)image)[o] = s; }
                else ((double*)image)[o] += wgt;
            }
        }
    return 0;
}

/* ---- space charge (space_charge_kick.py) ------------------------------------------------ */
/* :103-123 */
static double igf_primitive(double x, double y, double t) {
    const double r = sqrt(x * x + y * y + t * t);
    return -0.5 * t * t * atan(x * y / (t * r)) - 0.5 * y * y * atan(x * t / (y * r)) -
           0.5 * x * x * atan(y * t / (x * r)) + y * t * asinh(x / sqrt(y * y + t * t)) +
           x * t * asinh(y / sqrt(x * x + t * t)) + x * y * asinh(t / sqrt(x * x + y * y));
}

/* :163-291, fp64. cell[B][3] already holds (hx, hy, htau * gamma). G is [B][2gx][2gy][2gz]. */
CHXO_API int chxo_sc_igf(const double* cell, int64_t B, const int32_t* bins, double* G) {
    const int gx = bins[0], gy = bins[1], gz = bins[2];
    const int64_t GY = 2 * gy, GZ = 2 * gz, GN = (int64_t)8 * gx * gy * gz;
    memset(G, 0, (size_t)(B * GN) * sizeof(double));
    for (int64_t b = 0; b < B; ++b) {
        const double dx = cell[b * 3], dy = cell[b * 3 + 1], dt = cell[b * 3 + 2];
        double* Gb = G + b * GN;
#pragma omp parallel for collapse(2) schedule(static)
        for (int i = 0; i < gx; ++i)
            for (int j = 0; j < gy; ++j)
                for (int k = 0; k < gz; ++k) {
                    const double x = i * dx, y = j * dy, t = k * dt;
                    const double xp = x + 0.5 * dx, xm = x - 0.5 * dx, yp = y + 0.5 * dy, ym = y - 0.5 * dy,
                                 tp = t + 0.5 * dt, tm = t - 0.5 * dt;
                    const double g = igf_primitive(xp, yp, tp) - igf_primitive(xm, yp, tp) -
                                     igf_primitive(xp, ym, tp) - igf_primitive(xp, yp, tm) +
                                     igf_primitive(xp, ym, tm) + igf_primitive(xm, yp, tm) +
                                     igf_primitive(xm, ym, tp) - igf_primitive(xm, ym, tm);
                    const int64_t i2 = 2 * gx - i, j2 = 2 * gy - j, k2 = 2 * gz - k;
                    Gb[((int64_t)i * GY + j) * GZ + k] = g;
                    if (i > 0) Gb[(i2 * GY + j) * GZ + k] = g;
                    if (j > 0) Gb[((int64_t)i * GY + j2) * GZ + k] = g;
                    if (k > 0) Gb[((int64_t)i * GY + j) * GZ + k2] = g;
                    if (i > 0 && j > 0) Gb[(i2 * GY + j2) * GZ + k] = g;
                    if (j > 0 && k > 0) Gb[((int64_t)i * GY + j2) * GZ + k2] = g;
                    if (i > 0 && k > 0) Gb[(i2 * GY + j) * GZ + k2] = g;
                    if (i > 0 && j > 0 && k > 0) Gb[(i2 * GY + j2) * GZ + k2] = g;
                }
    }
    return 0;
}

/* :324-365 on a compact potential phi[B][gx][gy][gz] (double); F[B][gx][gy][gz][3] */
CHXO_API int chxo_sc_gradient(const double* phi, const double* cell, const double* gamma, int64_t B,
                              const int32_t* bins, double* F) {
    const int gx = bins[0], gy = bins[1], gz = bins[2];
    const int64_t n = (int64_t)gx * gy * gz;
    for (int64_t b = 0; b < B; ++b) {
        const double ig2 = gamma[b] != 0.0 ? 1.0 / (gamma[b] * gamma[b]) : 0.0;
        const double* p = phi + b * n;
        for (int i = 0; i < gx; ++i)
            for (int j = 0; j < gy; ++j)
                for (int k = 0; k < gz; ++k) {
                    const int64_t c = ((int64_t)i * gy + j) * gz + k;
                    double fx = 0, fy = 0, fz = 0;
                    if (i > 0 && i < gx - 1) fx = (p[c + (int64_t)gy * gz] - p[c - (int64_t)gy * gz]) * (0.5 * (1.0 / cell[b * 3]));
                    if (j > 0 && j < gy - 1) fy = (p[c + gz] - p[c - gz]) * (0.5 * (1.0 / cell[b * 3 + 1]));
                    if (k > 0 && k < gz - 1) fz = (p[c + 1] - p[c - 1]) * (0.5 * (1.0 / cell[b * 3 + 2]));
                    double* o = F + (b * n + c) * 3;
                    o[0] = -ig2 * fx; o[1] = -ig2 * fy; o[2] = -ig2 * fz;
                }
    }
    return 0;
}

typedef struct { double gamma, beta, p0, mc; } ref_frame_t;
static ref_frame_t ref_frame(double energy, double mass_eV) {
    ref_frame_t r;
    r.gamma = energy / mass_eV;                                               /* beam.py:323-326 */
    r.beta = fabs(r.gamma) > 0 ? sqrt(1.0 - 1.0 / (r.gamma * r.gamma)) : 1.0;   /* beam.py:328-336 */
    r.mc = mass_eV * kEvToKg * kC;
    r.p0 = r.gamma * r.beta * r.mc;
    return r;
}
/* particle_beam.py:1316-1346 */
static void to_si(const ref_frame_t* r, const double* v, double* s) {
    const double gi = r->gamma * (1.0 + v[5] * r->beta);
    const double bi = sqrt(1.0 - 1.0 / (gi * gi));
    const double P = gi * bi * r->mc; /* gamma * mass_kg * beta * c */
    const double px = v[1] * r->p0, py = v[3] * r->p0;
    s[0] = v[0]; s[1] = px; s[2] = v[2]; s[3] = py; s[4] = v[4] * -r->beta;
    s[5] = sqrt(P * P - px * px - py * py);
    s[6] = v[6];
}
/* particle_beam.py:1262-1314 */
static void from_si(const ref_frame_t* r, const double* s, double* v) {
    const double p = sqrt(s[1] * s[1] + s[3] * s[3] + s[5] * s[5]);
    const double q = p / r->mc;
    const double g = sqrt(1.0 + q * q);
    v[0] = s[0]; v[1] = s[1] / r->p0; v[2] = s[2]; v[3] = s[3] / r->p0;
    v[4] = -s[4] / r->beta;
    v[5] = (g - r->gamma) / (r->beta * r->gamma);
    v[6] = s[6];
}

CHXO_API int chxo_to_xyz_pxpypz(const void* x_in, const double* energy, double mass_eV, int64_t B,
                                int64_t Bx, int64_t Be, int64_t N, int dtype, void* out) {
    for (int64_t b = 0; b < B; ++b) {
        const ref_frame_t r = ref_frame(energy[Be == 1 ? 0 : b], mass_eV);
        for (int64_t n = 0; n < N; ++n) {
            double v[7], s[7];
            for (int j = 0; j < 7; ++j) v[j] = ld(x_in, dtype, (((Bx == 1 ? 0 : b) * N) + n) * 7 + j);
            to_si(&r, v, s);
            for (int j = 0; j < 7; ++j) st(out, dtype, (b * N + n) * 7 + j, s[j]);
        }
    }
    return 0;
}
CHXO_API int chxo_from_xyz_pxpypz(const void* x_in, const double* energy, double mass_eV, int64_t B,
                                  int64_t Bx, int64_t Be, int64_t N, int dtype, void* out) {
    for (int64_t b = 0; b < B; ++b) {
        const ref_frame_t r = ref_frame(energy[Be == 1 ? 0 : b], mass_eV);
        for (int64_t n = 0; n < N; ++n) {
            double v[7], s[7];
            for (int j = 0; j < 7; ++j) s[j] = ld(x_in, dtype, (((Bx == 1 ? 0 : b) * N) + n) * 7 + j);
            from_si(&r, s, v);
            for (int j = 0; j < 7; ++j) st(out, dtype, (b * N + n) * 7 + j, v[j]);
        }
    }
    return 0;
}

/* space_charge_kick.py:387-475 (node-based trilinear gather), :548-565 (kick), then back to
 * accelerator coordinates (:575-584). F is [B][gx][gy][gz][3] double; half/cell/dt double [B][3]/[B]. */
CHXO_API int chxo_sc_gather_kick(const void* x_in, const double* F, const double* half, const double* cell,
                                 const double* energy, const double* dt, double mass_eV, int64_t B,
                                 int64_t Bx, int64_t Be, int64_t N, const int32_t* bins, int dtype,
                                 void* x_out, double* forces_out /* optional [B][N][3] */) {
    const int g[3] = {bins[0], bins[1], bins[2]};
    const int64_t ncell = (int64_t)g[0] * g[1] * g[2];
    for (int64_t b = 0; b < B; ++b) {
        const ref_frame_t r = ref_frame(energy[Be == 1 ? 0 : b], mass_eV);
#pragma omp parallel for schedule(static)
        for (int64_t n = 0; n < N; ++n) {
            double v[7], s[7];
            for (int j = 0; j < 7; ++j) v[j] = ld(x_in, dtype, (((Bx == 1 ? 0 : b) * N) + n) * 7 + j);
            to_si(&r, v, s);
            const double pos[3] = {s[0], s[2], s[4]};
            double u[3]; int i0[3];
            for (int d = 0; d < 3; ++d) {
                u[d] = (pos[d] + half[b * 3 + d]) / cell[b * 3 + d];
                double fl = floor(u[d]);
                if (fl > 2.0e9) fl = 2.0e9;
                if (fl < -2.0e9) fl = -2.0e9;
                i0[d] = (int)fl;
            }
            double f[3] = {0, 0, 0};
            for (int ox = 0; ox < 2; ++ox) for (int oy = 0; oy < 2; ++oy) for (int oz = 0; oz < 2; ++oz) {
                const int ix = i0[0] + ox, iy = i0[1] + oy, iz = i0[2] + oz;
                if (ix < 0 || ix >= g[0] || iy < 0 || iy >= g[1] || iz < 0 || iz >= g[2]) continue;
                const double w = (1.0 - fabs(u[0] - ix)) * (1.0 - fabs(u[1] - iy)) * (1.0 - fabs(u[2] - iz)) * kE;
                const double* fc = F + (b * ncell + ((int64_t)ix * g[1] + iy) * g[2] + iz) * 3;
                f[0] += w * fc[0]; f[1] += w * fc[1]; f[2] += w * fc[2];
            }
            if (forces_out) for (int d = 0; d < 3; ++d) forces_out[(b * N + n) * 3 + d] = f[d];
            s[1] += f[0] * dt[b]; s[3] += f[1] * dt[b]; s[5] += f[2] * dt[b];
            from_si(&r, s, v);
            for (int j = 0; j < 7; ++j) st(x_out, dtype, (b * N + n) * 7 + j, v[j]);
        }
    }
    return 0;
}

/* non-linear tracking (drift_kick_drift, second_order) */
#include "chx_oracle_nonlinear.inc"

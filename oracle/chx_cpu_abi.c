/* chx_cpu_abi.c — include/chx_cpu.h: the core entry points of the C-ABI on host pointers, implemented by the CPU oracle.
 * TEST / CI INFRASTRUCTURE ONLY (see the header); the product library is cheetah_amd/libchx.so and has no CPU path. */
#define CHXO_API   /* the oracle's own entry points stay internal to this library (-fvisibility=hidden) */
#include "chx_oracle.c"

#include "../include/chx_cpu.h"

#define CPU_API __attribute__((visibility("default")))

static int bcast_ok(int64_t b, int64_t B) { return b == 1 || b == B; }

CPU_API int chx_abi_version_cpu(void) { return CHX_ABI_VERSION; }

/* chx_build_rmatrix: params[Bp][P] / energy[Be] / R_out[B][7][7] of `dtype`; evaluated in double, rounded once */
CPU_API int chx_build_rmatrix_cpu(int kind, const void* params, const void* energy, double mass_eV, double n_charges, int64_t B,
                                  int64_t Bp, int64_t Be, int dtype, void* R_out, void* stream) {
    (void)stream;
    const int P = kind_np(kind);
    if (P < 0 || !energy || !R_out || B < 1 || !bcast_ok(Bp, B) || !bcast_ok(Be, B) || (P > 0 && !params)) return CHX_ERR_INVALID_ARG;
    if (dtype != CHX_F32 && dtype != CHX_F64) return CHX_ERR_DTYPE;
    double* p = (double*)malloc(sizeof(double) * (size_t)(Bp * (P > 0 ? P : 1)));
    double* e = (double*)malloc(sizeof(double) * (size_t)Be);
    double* R = (double*)malloc(sizeof(double) * (size_t)B * 49);
    if (!p || !e || !R) { free(p); free(e); free(R); return CHX_ERR_WORKSPACE; }
    for (int64_t i = 0; i < Bp * P; ++i) p[i] = ld(params, dtype, i);
    for (int64_t i = 0; i < Be; ++i) e[i] = ld(energy, dtype, i);
    const int rc = chxo_build_rmatrix(kind, p, e, mass_eV, n_charges, B, Bp, Be, R);
    if (rc == 0)
        for (int64_t i = 0; i < B * 49; ++i) st(R_out, dtype, i, R[i]);
    free(p); free(e); free(R);
    return rc == 0 ? CHX_OK : CHX_ERR_INVALID_ARG;
}

/* chx_compose_maps: R_out[b] = M_{E-1}[b] ... M_0[b], accumulated in double, rounded once (segment.py:534-543) */
CPU_API int chx_compose_maps_cpu(const void* const* R_ptrs, const uint8_t* bcast, int64_t E, int64_t B, int dtype, void* R_out,
                                 void* stream) {
    (void)stream;
    if (!R_ptrs || !bcast || !R_out || E < 1 || B < 1) return CHX_ERR_INVALID_ARG;
    if (dtype != CHX_F32 && dtype != CHX_F64) return CHX_ERR_DTYPE;
    for (int64_t b = 0; b < B; ++b) {
        double tm[49], m[49];
        eye7(tm);
        for (int64_t e = 0; e < E; ++e) {
            if (!R_ptrs[e]) return CHX_ERR_INVALID_ARG;
            for (int k = 0; k < 49; ++k) m[k] = ld(R_ptrs[e], dtype, (bcast[e] ? 0 : b) * 49 + k);
            mm7(m, tm, tm);
        }
        for (int k = 0; k < 49; ++k) st(R_out, dtype, b * 49 + k, tm[k]);
    }
    return CHX_OK;
}

/* chx_apply_affine7: the fma chain of the device kernels in the working dtype (mode 1 of chxo_apply): bit-identical results */
CPU_API int chx_apply_affine7_cpu(const void* x_in, const void* R, void* x_out, int64_t B, int64_t Bx, int64_t BR, int64_t N,
                                  int dtype, void* stream) {
    (void)stream;
    if (!x_in || !R || !x_out || B < 1 || N < 1 || !bcast_ok(Bx, B) || !bcast_ok(BR, B)) return CHX_ERR_INVALID_ARG;
    if (dtype != CHX_F32 && dtype != CHX_F64) return CHX_ERR_DTYPE;
    return chxo_apply(x_in, R, x_out, B, Bx, BR, N, dtype, 1) == 0 ? CHX_OK : CHX_ERR_INVALID_ARG;
}

CPU_API int chx_moments_cpu(const void* x, const void* w, int64_t B, int64_t Bx, int64_t Bw, int64_t N, int dtype, double* out,
                            void* workspace, size_t workspace_bytes, void* stream) {
    (void)workspace; (void)workspace_bytes; (void)stream;
    if (!x || !out || B < 1 || N < 1 || !bcast_ok(Bx, B) || (w && !bcast_ok(Bw, B))) return CHX_ERR_INVALID_ARG;
    if (dtype != CHX_F32 && dtype != CHX_F64) return CHX_ERR_DTYPE;
    return chxo_moments(x, w, B, Bx, w ? Bw : 1, N, dtype, out) == 0 ? CHX_OK : CHX_ERR_INVALID_ARG;
}

/* chx_cic_deposit: grid += deposit; row-major grids (grid_strides all 0, or the row-major strides spelled out) */
CPU_API int chx_cic_deposit_cpu(const chx_cic_args* p, void* stream) {
    (void)stream;
    if (!p || !p->x || !p->extent || !p->grid || p->ndim < 1 || p->ndim > 3 || p->B < 1 || p->N < 1) return CHX_ERR_INVALID_ARG;
    if (p->dtype != CHX_F32 && p->dtype != CHX_F64) return CHX_ERR_DTYPE;
    int64_t total = 1, stride[3] = {0, 0, 0};
    for (int d = p->ndim - 1; d >= 0; --d) { stride[d] = total; total *= p->bins[d]; }
    int custom = 0;
    for (int d = 0; d < p->ndim; ++d) custom = custom || p->grid_strides[d] != 0;
    if (custom) {
        for (int d = 0; d < p->ndim; ++d)
            if (p->grid_strides[d] != stride[d]) return CHX_ERR_INVALID_ARG;
        if (p->grid_batch_stride != 0 && p->grid_batch_stride != total) return CHX_ERR_INVALID_ARG;
    }
    chxo_cic_args a;
    memset(&a, 0, sizeof(a));
    a.ndim = p->ndim;
    for (int d = 0; d < 3; ++d) { a.cols[d] = p->cols[d]; a.bins[d] = p->bins[d]; }
    a.B = p->B; a.Bx = p->Bx; a.Bq = p->Bq; a.Bs = p->Bs; a.Be = p->Be; a.Bsc = p->Bsc; a.Bsh = p->Bsh; a.N = p->N;
    a.dtype = p->dtype; a.abs_charge = p->abs_charge;
    a.x = p->x; a.charge = p->charge; a.survival = p->survival; a.extent = p->extent; a.scale = p->scale; a.shift = p->shift;
    return chxo_cic_deposit(&a, p->grid) == 0 ? CHX_OK : CHX_ERR_INVALID_ARG;
}

/* chx_track_elementwise: E passes of the fma chain, pass 0 from x_in, the others on x_out (a row is loaded whole before it is
 * stored, like the device's tile-local read-then-write) */
CPU_API int chx_track_elementwise_cpu(const void* x_in, const void* R, void* x_out, void* scratch, int64_t E, int64_t B, int64_t Bx,
                                      int64_t BR, int64_t N, int dtype, void* stream) {
    (void)scratch; (void)stream;
    if (!x_in || !R || !x_out || E < 1 || B < 1 || N < 1 || !bcast_ok(Bx, B) || !bcast_ok(BR, B) || x_in == x_out) return CHX_ERR_INVALID_ARG;
    if (dtype != CHX_F32 && dtype != CHX_F64) return CHX_ERR_DTYPE;
    const size_t esz = dtype == CHX_F32 ? 4 : 8;
    void* tmp = malloc((size_t)B * (size_t)N * 7 * esz);
    if (!tmp) return CHX_ERR_WORKSPACE;
    const void* src = x_in;
    int64_t src_B = Bx;
    for (int64_t e = 0; e < E; ++e) {
        void* dst = (((E - 1 - e) & 1) == 0) ? x_out : tmp;          /* ping-pong so that the last pass lands in x_out */
        if (chxo_apply(src, (const char*)R + (size_t)e * (size_t)BR * 49 * esz, dst, B, src_B, BR, N, dtype, 1) != 0) { free(tmp); return CHX_ERR_INVALID_ARG; }
        src = dst;
        src_B = B;
    }
    free(tmp);
    return CHX_OK;
}

CPU_API int chx_cavity_coeffs_cpu(const void* params, const void* energy, double mass_eV, double n_charges, int64_t B, int64_t Bp,
                                  int64_t Be, int dtype, double* coeffs, void* energy_out, void* stream) {
    (void)stream;
    if (!params || !energy || !coeffs || !energy_out || B < 1 || !bcast_ok(Bp, B) || !bcast_ok(Be, B)) return CHX_ERR_INVALID_ARG;
    if (dtype != CHX_F32 && dtype != CHX_F64) return CHX_ERR_DTYPE;
    double* p = (double*)malloc(sizeof(double) * (size_t)Bp * 4);
    double* e = (double*)malloc(sizeof(double) * (size_t)Be);
    double* eo = (double*)malloc(sizeof(double) * (size_t)B);
    if (!p || !e || !eo) { free(p); free(e); free(eo); return CHX_ERR_WORKSPACE; }
    for (int64_t i = 0; i < Bp * 4; ++i) p[i] = ld(params, dtype, i);
    for (int64_t i = 0; i < Be; ++i) e[i] = ld(energy, dtype, i);
    const int rc = chxo_cavity_coeffs(p, e, mass_eV, n_charges, B, Bp, Be, coeffs, eo);
    if (rc == 0)
        for (int64_t i = 0; i < B; ++i) st(energy_out, dtype, i, eo[i]);
    free(p); free(e); free(eo);
    return rc == 0 ? CHX_OK : CHX_ERR_INVALID_ARG;
}

CPU_API int chx_cavity_track_cpu(const void* x_in, const void* R, const double* coeffs, void* x_out, int64_t B, int64_t Bx, int64_t N,
                                 int dtype, void* stream) {
    (void)stream;
    if (!x_in || !R || !coeffs || !x_out || B < 1 || N < 1 || !bcast_ok(Bx, B)) return CHX_ERR_INVALID_ARG;
    if (dtype != CHX_F32 && dtype != CHX_F64) return CHX_ERR_DTYPE;
    return chxo_cavity_track(x_in, R, coeffs, x_out, B, Bx, N, dtype) == 0 ? CHX_OK : CHX_ERR_INVALID_ARG;
}

/* chx_hist2d: image[b][jy][jx] += |q| * survival of the particles inside the edges (the caller zeroes the image) */
CPU_API int chx_hist2d_cpu(const chx_hist2d_args* p, void* stream) {
    (void)stream;
    if (!p || !p->x || !p->edges_x || !p->edges_y || !p->image || p->B < 1 || p->N < 1 || p->nx < 1 || p->ny < 1) return CHX_ERR_INVALID_ARG;
    if (p->dtype != CHX_F32 && p->dtype != CHX_F64) return CHX_ERR_DTYPE;
    if (!bcast_ok(p->Bx, p->B) || (p->charge && !bcast_ok(p->Bq, p->B)) || (p->survival && !bcast_ok(p->Bs, p->B)) ||
        (p->shift && !bcast_ok(p->Bsh, p->B)))
        return CHX_ERR_INVALID_ARG;
    return chxo_hist2d(p->x, p->charge, p->survival, p->shift, p->edges_x, p->edges_y, p->B, p->Bx, p->Bq, p->Bs, p->Bsh, p->N, p->nx,
                       p->ny, p->dtype, p->image, NULL) == 0 ? CHX_OK : CHX_ERR_INVALID_ARG;
}

/* ---- chx_sc_kick: SpaceChargeKick.track (space_charge_kick.py:477-586) on host pointers ---------------------------------------------
 * beam sizes -> grid (:531-550) -> cloud-in-cell charge (:556-563) -> integrated Green function on the doubled grid (:163-291) ->
 * cyclic convolution (:293-322; radix-2 transforms below, the grids are powers of two like the device path's) -> field (:324-385) ->
 * trilinear gather + kick (:387-475, :548-584) -> optional linear map behind the kick (the device entry point's post_map). The steps
 * and their working precisions are those of oracle/chx_oracle.py `space_charge_kick` (index / weight arithmetic of the deposit in the
 * beam dtype, the Poisson solve in double). */
static void fft1d(double* re, double* im, int n, int64_t stride, int inverse) {
    for (int i = 1, j = 0; i < n; ++i) {                       /* bit reversal */
        int bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) {
            double t = re[i * stride]; re[i * stride] = re[j * stride]; re[j * stride] = t;
            t = im[i * stride]; im[i * stride] = im[j * stride]; im[j * stride] = t;
        }
    }
    for (int len = 2; len <= n; len <<= 1) {
        const double ang = (inverse ? 2.0 : -2.0) * kPi / len;
        for (int i = 0; i < n; i += len)
            for (int k = 0; k < len / 2; ++k) {
                const double wr = cos(ang * k), wi = sin(ang * k);
                const int64_t a = (int64_t)(i + k) * stride, b = (int64_t)(i + k + len / 2) * stride;
                const double xr = re[b] * wr - im[b] * wi, xi = re[b] * wi + im[b] * wr;
                re[b] = re[a] - xr; im[b] = im[a] - xi;
                re[a] += xr; im[a] += xi;
            }
    }
}

static void fft3d(double* re, double* im, int nx, int ny, int nz, int inverse) {
    const int64_t sy = nz, sx = (int64_t)ny * nz;
#pragma omp parallel for collapse(2) schedule(static)
    for (int i = 0; i < nx; ++i)
        for (int j = 0; j < ny; ++j) fft1d(re + i * sx + j * sy, im + i * sx + j * sy, nz, 1, inverse);
#pragma omp parallel for collapse(2) schedule(static)
    for (int i = 0; i < nx; ++i)
        for (int k = 0; k < nz; ++k) fft1d(re + i * sx + k, im + i * sx + k, ny, sy, inverse);
#pragma omp parallel for collapse(2) schedule(static)
    for (int j = 0; j < ny; ++j)
        for (int k = 0; k < nz; ++k) fft1d(re + j * sy + k, im + j * sy + k, nx, sx, inverse);
}

static int pow2_grid(int g) { return g >= 16 && g <= 512 && (g & (g - 1)) == 0; }

CPU_API size_t chx_sc_kick_workspace_bytes_cpu(int64_t B, int64_t N, const int32_t* bins, int dtype) {
    (void)N;
    return (B >= 1 && bins && (dtype == CHX_F32 || dtype == CHX_F64) && pow2_grid(bins[0]) && pow2_grid(bins[1]) && pow2_grid(bins[2])) ? 256 : 0;
}

CPU_API int chx_sc_kick_cpu(const void* x_in, const void* charge, const void* survival, const void* energy, const void* length,
                            const void* grid_extent, double mass_eV, int64_t B, int64_t Bx, int64_t Bq, int64_t Bs, int64_t Bext, int64_t N,
                            const int32_t* bins, int dtype, void* x_out, void* workspace, size_t workspace_bytes, void* stream,
                            void* side_stream, const void* post_map, int64_t BR) {
    (void)workspace; (void)workspace_bytes; (void)stream; (void)side_stream;
    if (!x_in || !charge || !survival || !energy || !length || !grid_extent || !x_out || !bins) return CHX_ERR_INVALID_ARG;
    if (B < 1 || N < 1 || !bcast_ok(Bx, B) || !bcast_ok(Bq, B) || !bcast_ok(Bs, B) || !bcast_ok(Bext, B)) return CHX_ERR_INVALID_ARG;
    if (dtype != CHX_F32 && dtype != CHX_F64) return CHX_ERR_DTYPE;
    if (!pow2_grid(bins[0]) || !pow2_grid(bins[1]) || !pow2_grid(bins[2])) return CHX_ERR_INVALID_ARG;
    if (post_map && !bcast_ok(BR, B)) return CHX_ERR_INVALID_ARG;
    const int g[3] = {bins[0], bins[1], bins[2]};
    const size_t esz = dtype == CHX_F32 ? 4 : 8;
    const int64_t ncell = (int64_t)g[0] * g[1] * g[2], npad = 8 * ncell;
    const int PX = 2 * g[0], PY = 2 * g[1], PZ = 2 * g[2];
    int rc = CHX_ERR_WORKSPACE;
    double* mom = (double*)malloc(sizeof(double) * (size_t)B * 29);
    void* geo_T = malloc(esz * (size_t)B * 9);                 /* extent[B][3][2] | scale[B][3] in the beam dtype */
    double* half = (double*)malloc(sizeof(double) * (size_t)B * 3);
    double* cell = (double*)malloc(sizeof(double) * (size_t)B * 3);
    double* cell_scaled = (double*)malloc(sizeof(double) * (size_t)B * 3);
    double* gamma = (double*)malloc(sizeof(double) * (size_t)B);
    double* en = (double*)malloc(sizeof(double) * (size_t)B);
    double* dtk = (double*)malloc(sizeof(double) * (size_t)B);
    void* rho = calloc((size_t)B * (size_t)ncell, esz);
    double* G = (double*)malloc(sizeof(double) * (size_t)B * (size_t)npad);
    double* gi = (double*)calloc((size_t)npad, sizeof(double));
    double* pr = (double*)malloc(sizeof(double) * (size_t)npad);
    double* pi = (double*)malloc(sizeof(double) * (size_t)npad);
    double* phi = (double*)malloc(sizeof(double) * (size_t)B * (size_t)ncell);
    double* F = (double*)malloc(sizeof(double) * (size_t)B * (size_t)ncell * 3);
    void* kicked = post_map ? malloc(esz * (size_t)B * (size_t)N * 7) : NULL;
    if (!mom || !geo_T || !half || !cell || !cell_scaled || !gamma || !en || !dtk || !rho || !G || !gi || !pr || !pi || !phi || !F ||
        (post_map && !kicked))
        goto done;
    rc = CHX_ERR_INVALID_ARG;
    if (chxo_moments(x_in, survival, B, Bx, Bs, N, dtype, mom) != 0) goto done;
    for (int64_t b = 0; b < B; ++b) {
        static const int diag[3] = {8, 19, 26};                /* cov_xx, cov_yy, cov_tautau in the [W, W2, mu(6), cov(21)] row */
        en[b] = ld(energy, dtype, b);                            /* energy / length: one value per batch row */
        gamma[b] = en[b] / mass_eV;
        const double beta = sqrt(1.0 - 1.0 / (gamma[b] * gamma[b]));
        dtk[b] = ld(length, dtype, b) / (kC * beta);           /* :548-550 */
        for (int d = 0; d < 3; ++d) {
            const double ext = ld(grid_extent, dtype, (Bext == 1 ? 0 : b) * 3 + d);
            double h, c;
            if (dtype == CHX_F32) {                               /* sigma, half, cell formed in the beam dtype (:531-547) */
                const float sig = (float)sqrt(mom[b * 29 + diag[d]]);
                const float hf = (float)ext * sig;
                const float cf = (2.0f * hf) / (float)g[d];
                h = hf; c = cf;
            } else {
                const double sig = sqrt(mom[b * 29 + diag[d]]);
                h = ext * sig;
                c = (2.0 * h) / (double)g[d];
            }
            half[b * 3 + d] = h;
            cell[b * 3 + d] = c;
            st(geo_T, dtype, b * 6 + d * 2, -h);
            st(geo_T, dtype, b * 6 + d * 2 + 1, h);
        }
        for (int d = 0; d < 3; ++d) st((char*)geo_T + esz * (size_t)B * 6, dtype, b * 3 + d, d == 2 ? -beta : 1.0);   /* z = tau * -beta */
        cell_scaled[b * 3] = cell[b * 3];
        cell_scaled[b * 3 + 1] = cell[b * 3 + 1];
        cell_scaled[b * 3 + 2] = dtype == CHX_F32 ? (double)((float)cell[b * 3 + 2] * (float)gamma[b]) : cell[b * 3 + 2] * gamma[b];   /* :170-176 */
    }
    {
        chxo_cic_args a;
        memset(&a, 0, sizeof(a));
        a.ndim = 3;
        a.cols[0] = 0; a.cols[1] = 2; a.cols[2] = 4;
        for (int d = 0; d < 3; ++d) a.bins[d] = g[d];
        a.B = B; a.Bx = Bx; a.Bq = Bq; a.Bs = Bs; a.Be = B; a.Bsc = B; a.Bsh = 1; a.N = N;
        a.dtype = dtype; a.abs_charge = 0;
        a.x = x_in; a.charge = charge; a.survival = survival; a.extent = geo_T; a.scale = (char*)geo_T + esz * (size_t)B * 6; a.shift = NULL;
        if (chxo_cic_deposit(&a, rho) != 0) goto done;
    }
    if (chxo_sc_igf(cell_scaled, B, bins, G) != 0) goto done;
    for (int64_t b = 0; b < B; ++b) {
        const double inv_vol = 1.0 / (cell[b * 3] * cell[b * 3 + 1] * cell[b * 3 + 2]);      /* :144-146 */
        memset(pr, 0, sizeof(double) * (size_t)npad);
        memset(pi, 0, sizeof(double) * (size_t)npad);
        for (int i = 0; i < g[0]; ++i)
            for (int j = 0; j < g[1]; ++j)
                for (int k = 0; k < g[2]; ++k)
                    pr[((int64_t)i * PY + j) * PZ + k] = ld(rho, dtype, b * ncell + ((int64_t)i * g[1] + j) * g[2] + k) * inv_vol;
        double* gr = G + b * npad;
        memset(gi, 0, sizeof(double) * (size_t)npad);
        fft3d(pr, pi, PX, PY, PZ, 0);
        fft3d(gr, gi, PX, PY, PZ, 0);
        for (int64_t q = 0; q < npad; ++q) {
            const double ar = pr[q], ai = pi[q];
            pr[q] = ar * gr[q] - ai * gi[q];
            pi[q] = ar * gi[q] + ai * gr[q];
        }
        fft3d(pr, pi, PX, PY, PZ, 1);
        const double norm = 1.0 / (double)npad / (4.0 * kPi * 8.8541878188e-12);             /* :306-316 */
        for (int i = 0; i < g[0]; ++i)
            for (int j = 0; j < g[1]; ++j)
                for (int k = 0; k < g[2]; ++k)
                    phi[b * ncell + ((int64_t)i * g[1] + j) * g[2] + k] = pr[((int64_t)i * PY + j) * PZ + k] * norm;
    }
    if (chxo_sc_gradient(phi, cell, gamma, B, bins, F) != 0) goto done;
    if (chxo_sc_gather_kick(x_in, F, half, cell, en, dtk, mass_eV, B, Bx, B, N, bins, dtype, post_map ? kicked : x_out, NULL) != 0) goto done;
    if (post_map && chxo_apply(kicked, post_map, x_out, B, B, BR, N, dtype, 1) != 0) goto done;
    rc = CHX_OK;
done:
    free(mom); free(geo_T); free(half); free(cell); free(cell_scaled); free(gamma); free(en); free(dtk); free(rho); free(G); free(gi);
    free(pr); free(pi); free(phi); free(F); free(kicked);
    return rc;
}

/* ---- round 6: the rest of SURVEY 8(b)'s list — the fused track, the backward entry points, the gather ------------------------------ */

/* chx_track_fused: the same rounding per element as chx_track_elementwise (the device keeps the row in registers; the numbers are
 * those of E passes) */
CPU_API int chx_track_fused_cpu(const void* x_in, const void* R, void* x_out, int64_t E, int64_t B, int64_t Bx, int64_t BR, int64_t N,
                                int dtype, void* stream) {
    return chx_track_elementwise_cpu(x_in, R, x_out, NULL, E, B, Bx, BR, N, dtype, stream);
}

/* chx_apply_affine7_bwd (element.py:180-191 under autograd): dX[b][n][j] = sum_i dY[b][n][i] R[b][i][j] (double accumulation,
 * rounded once to dtype; a shared x_in, Bx == 1, still gets B rows of dX like the device entry point — the caller sums them),
 * dR[b][i][j] = sum_n dY[b][n][i] X[b][n][j] in double */
CPU_API size_t chx_apply_bwd_workspace_bytes_cpu(int64_t B, int64_t N) { (void)B; (void)N; return 0; }
CPU_API int chx_apply_affine7_bwd_cpu(const void* dY, const void* R, const void* X, void* dX, double* dR, int64_t B, int64_t Bx,
                                      int64_t BR, int64_t N, int dtype, void* workspace, size_t workspace_bytes, void* stream) {
    (void)workspace; (void)workspace_bytes; (void)stream;
    if (!dY || B < 1 || N < 1 || !bcast_ok(Bx, B) || !bcast_ok(BR, B) || (dX && !R) || (dR && !X)) return CHX_ERR_INVALID_ARG;
    if (dtype != CHX_F32 && dtype != CHX_F64) return CHX_ERR_DTYPE;
    for (int64_t b = 0; b < B; ++b) {
        double acc[49];
        for (int k = 0; k < 49; ++k) acc[k] = 0.0;
        for (int64_t n = 0; n < N; ++n) {
            double g[7];
            for (int i = 0; i < 7; ++i) g[i] = ld(dY, dtype, (b * N + n) * 7 + i);
            if (dX)
                for (int j = 0; j < 7; ++j) {
                    double s = 0.0;
                    for (int i = 0; i < 7; ++i) s += g[i] * ld(R, dtype, (BR == 1 ? 0 : b) * 49 + i * 7 + j);
                    st(dX, dtype, (b * N + n) * 7 + j, s);
                }
            if (dR)
                for (int i = 0; i < 7; ++i)
                    for (int j = 0; j < 7; ++j) acc[i * 7 + j] += g[i] * ld(X, dtype, ((Bx == 1 ? 0 : b) * N + n) * 7 + j);
        }
        if (dR)
            for (int k = 0; k < 49; ++k) dR[b * 49 + k] = acc[k];
    }
    return CHX_OK;
}

/* chx_moments_bwd(_w): the cotangent of utils/statistics.py:4-62 (csrc/chx_moments.hip moments_bwd_kernel restated): with
 * cf = W - W2 / W, Gsym = g + g^T on the upper-triangular cotangent g of the covariances, d_n = x_n - mu,
 *   dX[n][a] = w_n (g_mu[a] / W + (Gsym d_n)[a] / cf)
 *   dW[n]    = g_W + 2 w_n g_W2 + (g_mu . d_n) / W + (d_n^T Gsym d_n / 2 - S (1 + W2 / W^2 - 2 w_n / W)) / cf,  S = sum g_ab cov_ab */
CPU_API int chx_moments_bwd_w_cpu(const void* x, const void* w, const double* out, const double* d_out, int64_t B, int64_t Bx, int64_t Bw,
                                  int64_t N, int dtype, void* dX, void* dW, void* stream) {
    (void)stream;
    if (!x || !out || !d_out || B < 1 || N < 1 || !bcast_ok(Bx, B) || (w && !bcast_ok(Bw, B))) return CHX_ERR_INVALID_ARG;
    if (dtype != CHX_F32 && dtype != CHX_F64) return CHX_ERR_DTYPE;
    for (int64_t b = 0; b < B; ++b) {
        const double* o = out + b * 29;
        const double* g = d_out + b * 29;
        const double W = o[0], W2 = o[1], icf = 1.0 / (W - W2 / W), kcf = 1.0 + W2 / (W * W);
        double G[6][6], S = 0.0;
        int k = 0;
        for (int i = 0; i < 6; ++i)
            for (int j = i; j < 6; ++j, ++k) {
                if (i == j) G[i][i] = 2.0 * g[8 + k];
                else G[i][j] = G[j][i] = g[8 + k];
                S += g[8 + k] * o[8 + k];
            }
        for (int64_t n = 0; n < N; ++n) {
            double d[6], quad = 0.0, lin = 0.0;
            for (int j = 0; j < 6; ++j) d[j] = ld(x, dtype, ((Bx == 1 ? 0 : b) * N + n) * 7 + j) - o[2 + j];
            const double wv = w ? ld(w, dtype, (Bw == 1 ? 0 : b) * N + n) : 1.0;
            for (int a = 0; a < 6; ++a) {
                double s = 0.0;
                for (int c = 0; c < 6; ++c) s += G[a][c] * d[c];
                quad += d[a] * s;
                lin += g[2 + a] * d[a];
                if (dX) st(dX, dtype, (b * N + n) * 7 + a, wv * (g[2 + a] / W + icf * s));
            }
            if (dX) st(dX, dtype, (b * N + n) * 7 + 6, 0.0);
            if (dW) st(dW, dtype, b * N + n, g[0] + 2.0 * wv * g[1] + lin / W + icf * (0.5 * quad - S * (kcf - 2.0 * wv / W)));
        }
    }
    return CHX_OK;
}

CPU_API int chx_moments_bwd_cpu(const void* x, const void* w, const double* out, const double* d_out, int64_t B, int64_t Bx, int64_t Bw,
                                int64_t N, int dtype, void* dX, void* stream) {
    return chx_moments_bwd_w_cpu(x, w, out, d_out, B, Bx, Bw, N, dtype, dX, NULL, stream);
}

/* chx_cic_deposit_bwd (utils/cloud_in_cell.py under autograd; csrc/chx_cic.hip cic_bwd_kernel restated): a particle inside the
 * extent receives dweight = sum over its in-grid corners of dgrid x corner weight, dpos[d] = charge x sum of dgrid x
 * (+-1 along d) x the other axes' weights / bin width; a particle outside gets zeros. Cell index, fraction and charge are the
 * oracle's dtype-faithful ones. Row-major grids. */
CPU_API int chx_cic_deposit_bwd_cpu(const chx_cic_args* p, const void* dgrid, void* dweight, void* dpos, void* stream) {
    (void)stream;
    if (!p || !p->x || !p->extent || !dgrid || p->ndim < 1 || p->ndim > 3 || p->B < 1 || p->N < 1) return CHX_ERR_INVALID_ARG;
    if (p->dtype != CHX_F32 && p->dtype != CHX_F64) return CHX_ERR_DTYPE;
    int64_t total = 1, stride[3] = {0, 0, 0};
    for (int d = p->ndim - 1; d >= 0; --d) { stride[d] = total; total *= p->bins[d]; }
    for (int d = 0; d < p->ndim; ++d)
        if (p->grid_strides[d] != 0 && p->grid_strides[d] != stride[d]) return CHX_ERR_INVALID_ARG;
    chxo_cic_args a;
    memset(&a, 0, sizeof(a));
    a.ndim = p->ndim;
    for (int d = 0; d < 3; ++d) { a.cols[d] = p->cols[d]; a.bins[d] = p->bins[d]; }
    a.B = p->B; a.Bx = p->Bx; a.Bq = p->Bq; a.Bs = p->Bs; a.Be = p->Be; a.Bsc = p->Bsc; a.Bsh = p->Bsh; a.N = p->N;
    a.dtype = p->dtype; a.abs_charge = p->abs_charge;
    a.x = p->x; a.charge = p->charge; a.survival = p->survival; a.extent = p->extent; a.scale = p->scale; a.shift = p->shift;
    const int nc = 1 << a.ndim;
    for (int64_t b = 0; b < a.B; ++b)
        for (int64_t n = 0; n < a.N; ++n) {
            int64_t idx[3] = {0, 0, 0};
            double f[3] = {0, 0, 0}, bw[3] = {1, 1, 1}, c;
            int inside;
            if (a.dtype == CHX_F32) {
                float ff[3];
                inside = cic_locate_f32(&a, b, n, idx, ff);
                c = (double)cic_charge_f32(&a, b, n);
                for (int d = 0; d < a.ndim; ++d) {
                    f[d] = ff[d];
                    const float* e = (const float*)a.extent + (a.Be == 1 ? 0 : b) * a.ndim * 2 + d * 2;
                    bw[d] = (double)((e[1] - e[0]) / (float)a.bins[d]);
                }
            } else {
                inside = cic_locate_f64(&a, b, n, idx, f);
                c = cic_charge_f64(&a, b, n);
                for (int d = 0; d < a.ndim; ++d) {
                    const double* e = (const double*)a.extent + (a.Be == 1 ? 0 : b) * a.ndim * 2 + d * 2;
                    bw[d] = (e[1] - e[0]) / (double)a.bins[d];
                }
            }
            double dw = 0.0, dp[3] = {0.0, 0.0, 0.0};
            for (int corner = 0; inside && corner < nc; ++corner) {
                const int o[3] = {corner & 1, (corner >> 1) & 1, (corner >> 2) & 1};
                int valid = 1;
                int64_t off = 0;
                double wf[3] = {1.0, 1.0, 1.0}, sg[3] = {0.0, 0.0, 0.0};
                for (int d = 0; d < a.ndim; ++d) {
                    const int64_t id = idx[d] + o[d];
                    valid = valid && id >= 0 && id < a.bins[d];
                    off += (id < 0 ? 0 : (id > a.bins[d] - 1 ? a.bins[d] - 1 : id)) * stride[d];
                    wf[d] = o[d] ? f[d] : 1.0 - f[d];
                    sg[d] = o[d] ? 1.0 : -1.0;
                }
                if (!valid) continue;
                const double gv = ld(dgrid, a.dtype, b * total + off);
                dw += gv * wf[0] * wf[1] * wf[2];
                for (int d = 0; d < a.ndim; ++d) {
                    double prod = sg[d];
                    for (int e = 0; e < a.ndim; ++e)
                        if (e != d) prod *= wf[e];
                    dp[d] += c * gv * prod / bw[d];
                }
            }
            if (dweight) st(dweight, a.dtype, b * a.N + n, dw);
            if (dpos)
                for (int d = 0; d < a.ndim; ++d) st(dpos, a.dtype, (b * a.N + n) * a.ndim + d, dp[d]);
        }
    return CHX_OK;
}

/* chx_sc_gather_kick (space_charge_kick.py:387-475, 548-584; particle_beam.py:1262-1346): F[B][gx][gy][gz][4] (x, y, z, pad) and
 * half / cell / energy / dt of `dtype`, the particle step in double like the oracle's, rounded once */
CPU_API int chx_sc_gather_kick_cpu(const void* x_in, const void* F, const void* half, const void* cell, const void* energy, const void* dt,
                                   double mass_eV, int64_t B, int64_t Bx, int64_t Be, int64_t N, const int32_t* bins, int dtype,
                                   void* x_out, void* stream) {
    (void)stream;
    if (!x_in || !F || !half || !cell || !energy || !dt || !x_out || !bins || B < 1 || N < 1 || !bcast_ok(Bx, B) || !bcast_ok(Be, B))
        return CHX_ERR_INVALID_ARG;
    if (dtype != CHX_F32 && dtype != CHX_F64) return CHX_ERR_DTYPE;
    if (bins[0] < 1 || bins[1] < 1 || bins[2] < 1) return CHX_ERR_INVALID_ARG;
    const int64_t ncell = (int64_t)bins[0] * bins[1] * bins[2];
    double* F3 = (double*)malloc(sizeof(double) * (size_t)B * (size_t)ncell * 3);
    double* hd = (double*)malloc(sizeof(double) * (size_t)B * 3);
    double* cd = (double*)malloc(sizeof(double) * (size_t)B * 3);
    double* ed = (double*)malloc(sizeof(double) * (size_t)Be);
    double* td = (double*)malloc(sizeof(double) * (size_t)B);
    int rc = CHX_ERR_WORKSPACE;
    if (F3 && hd && cd && ed && td) {
        for (int64_t i = 0; i < B * ncell; ++i)
            for (int d = 0; d < 3; ++d) F3[i * 3 + d] = ld(F, dtype, i * 4 + d);
        for (int64_t i = 0; i < B * 3; ++i) { hd[i] = ld(half, dtype, i); cd[i] = ld(cell, dtype, i); }
        for (int64_t i = 0; i < Be; ++i) ed[i] = ld(energy, dtype, i);
        for (int64_t i = 0; i < B; ++i) td[i] = ld(dt, dtype, i);
        rc = chxo_sc_gather_kick(x_in, F3, hd, cd, ed, td, mass_eV, B, Bx, Be, N, bins, dtype, x_out, NULL) == 0 ? CHX_OK : CHX_ERR_INVALID_ARG;
    }
    free(F3); free(hd); free(cd); free(ed); free(td);
    return rc;
}


/* ---- round 6, second batch: the algebra around a beam property (particle_beam.py:1672-1943) and the scalar-run builders ---- */

/* chx_merge_moments (exact pooled statistics of R shards, Chan et al.): per_rank[R][B][29] -> out[B][29] */
CPU_API int chx_merge_moments_cpu(const double* per_rank, int32_t R, int64_t B, double* out, void* stream) {
    (void)stream;
    if (!per_rank || !out || R < 1 || B < 1) return CHX_ERR_INVALID_ARG;
    for (int64_t b = 0; b < B; ++b) {
        double W = 0.0, W2 = 0.0, mu[6] = {0, 0, 0, 0, 0, 0}, M[21];
        for (int r = 0; r < R; ++r) {
            const double* p = per_rank + ((int64_t)r * B + b) * 29;
            if (!(p[0] > 0.0)) continue;
            W += p[0];
            W2 += p[1];
            for (int j = 0; j < 6; ++j) mu[j] += p[0] * p[2 + j];
        }
        for (int j = 0; j < 6; ++j) mu[j] /= W;
        for (int k = 0; k < 21; ++k) M[k] = 0.0;
        for (int r = 0; r < R; ++r) {
            const double* p = per_rank + ((int64_t)r * B + b) * 29;
            if (!(p[0] > 0.0)) continue;
            const double cf = p[0] - p[1] / p[0];
            int k = 0;
            for (int i = 0; i < 6; ++i)
                for (int j = i; j < 6; ++j, ++k) M[k] += p[8 + k] * cf + p[0] * (p[2 + i] - mu[i]) * (p[2 + j] - mu[j]);
        }
        double* o = out + b * 29;
        o[0] = W;
        o[1] = W2;
        for (int j = 0; j < 6; ++j) o[2 + j] = mu[j];
        for (int k = 0; k < 21; ++k) o[8 + k] = M[k] / (W - W2 / W);
    }
    return CHX_OK;
}

/* dR[49] of one batch row from the gradient g[29] of the moments of y = R x: mu' = A mu + b, cov' = A C A^T (element.py:180-191
 * followed by statistics.py:4-62): dA = 2 G A C + g_mu mu^T, db = g_mu, G symmetric with the off-diagonal gradients halved */
static void mapped_bwd_row(const double* g, const double* Rb, const double* m, double* o) {
    double G[36], C[36], AC[36];
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) {
            const int lo = i < j ? i : j, hi = i < j ? j : i;
            const int k = 8 + lo * 6 - (lo * (lo - 1)) / 2 + (hi - lo);
            G[i * 6 + j] = (i == j) ? g[k] : 0.5 * g[k];
            C[i * 6 + j] = m[k];
        }
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) {
            double acc = 0.0;
            for (int l = 0; l < 6; ++l) acc += Rb[i * 7 + l] * C[l * 6 + j];
            AC[i * 6 + j] = acc;
        }
    for (int k = 0; k < 49; ++k) o[k] = 0.0;
    for (int i = 0; i < 6; ++i) {
        for (int j = 0; j < 6; ++j) {
            double acc = 0.0;
            for (int l = 0; l < 6; ++l) acc += G[i * 6 + l] * AC[l * 6 + j];
            o[i * 7 + j] = 2.0 * acc + g[2 + i] * m[2 + j];
        }
        o[i * 7 + 6] = g[2 + i];
    }
}

CPU_API int chx_moments_mapped_bwd_cpu(const double* d_out, const void* R, const double* mom_x, int64_t B, int64_t BR, int64_t Bm,
                                       int dtype, double* dR, void* stream) {
    (void)stream;
    if (!d_out || !R || !mom_x || !dR || B < 1 || (BR != 1 && BR != B) || (Bm != 1 && Bm != B)) return CHX_ERR_INVALID_ARG;
    if (dtype != CHX_F32 && dtype != CHX_F64) return CHX_ERR_DTYPE;
    for (int64_t b = 0; b < B; ++b) {
        double Rb[49];
        for (int k = 0; k < 49; ++k) Rb[k] = ld(R, dtype, (BR == 1 ? 0 : b) * 49 + k);
        mapped_bwd_row(d_out + b * 29, Rb, mom_x + (Bm == 1 ? 0 : b) * 29, dR + b * 49);
    }
    return CHX_OK;
}

CPU_API int chx_moment_entry_cpu(const double* mom, int64_t B, int index, int take_sqrt, int dtype, void* out, void* stream) {
    (void)stream;
    if (!mom || !out || B < 1 || index < 0 || index >= 29) return CHX_ERR_INVALID_ARG;
    if (dtype != CHX_F32 && dtype != CHX_F64) return CHX_ERR_DTYPE;
    for (int64_t b = 0; b < B; ++b) st(out, dtype, b, take_sqrt ? sqrt(mom[b * 29 + index]) : mom[b * 29 + index]);
    return CHX_OK;
}

CPU_API int chx_moment_entry_mapped_bwd_cpu(const void* grad, const double* mom_y, int index, int take_sqrt, const void* R,
                                            const double* mom_x, int64_t B, int64_t BR, int64_t Bm, int dtype, void* dR, int dR_is_double,
                                            void* stream) {
    (void)stream;
    if (!grad || !mom_y || !R || !mom_x || !dR || B < 1 || (BR != 1 && BR != B) || (Bm != 1 && Bm != B) || index < 2 || index >= 29)
        return CHX_ERR_INVALID_ARG;
    if (dtype != CHX_F32 && dtype != CHX_F64) return CHX_ERR_DTYPE;
    for (int64_t b = 0; b < B; ++b) {
        double g[29], Rb[49], o[49];
        for (int k = 0; k < 29; ++k) g[k] = 0.0;
        g[index] = ld(grad, dtype, b);
        if (take_sqrt) g[index] = g[index] * 0.5 / sqrt(mom_y[b * 29 + index]);
        for (int k = 0; k < 49; ++k) Rb[k] = ld(R, dtype, (BR == 1 ? 0 : b) * 49 + k);
        mapped_bwd_row(g, Rb, mom_x + (Bm == 1 ? 0 : b) * 29, o);
        for (int k = 0; k < 49; ++k) {
            if (dR_is_double) ((double*)dR)[b * 49 + k] = o[k];
            else st(dR, dtype, b * 49 + k, o[k]);
        }
    }
    return CHX_OK;
}

/* chx_build_rmatrix_scalars / chx_run_build_compose: E elements whose parameters are scalars read where they live (host pointers
 * here), each map rounded to `dtype`; the composed map accumulated in double from the ROUNDED element maps, rounded once */
static int build_scalars(const int32_t* kinds, const void* const* param_ptrs, int64_t E, const void* energy, double mass_eV,
                         double n_charges, int dtype, void* maps) {
    if (!kinds || !param_ptrs || !energy || !maps || E < 1) return CHX_ERR_INVALID_ARG;
    if (dtype != CHX_F32 && dtype != CHX_F64) return CHX_ERR_DTYPE;
    const double e = ld(energy, dtype, 0);
    for (int64_t i = 0; i < E; ++i) {
        const int P = kind_np(kinds[i]);
        if (P < 0) return CHX_ERR_INVALID_ARG;
        double p[CHX_MAX_PARAMS], R[49];
        for (int k = 0; k < P; ++k) {
            if (!param_ptrs[i * CHX_MAX_PARAMS + k]) return CHX_ERR_INVALID_ARG;
            p[k] = ld(param_ptrs[i * CHX_MAX_PARAMS + k], dtype, 0);
        }
        if (chxo_build_rmatrix(kinds[i], p, &e, mass_eV, n_charges, 1, 1, 1, R) != 0) return CHX_ERR_INVALID_ARG;
        for (int k = 0; k < 49; ++k) st(maps, dtype, i * 49 + k, R[k]);
    }
    return CHX_OK;
}

CPU_API int chx_build_rmatrix_scalars_cpu(const int32_t* kinds, const void* const* param_ptrs, int64_t E, const void* energy,
                                          double mass_eV, double n_charges, int dtype, void* R_out, void* stream) {
    (void)stream;
    return build_scalars(kinds, param_ptrs, E, energy, mass_eV, n_charges, dtype, R_out);
}

CPU_API int chx_run_build_compose_cpu(const int32_t* kinds, const void* const* param_ptrs, int64_t E, const void* energy, double mass_eV,
                                      double n_charges, int dtype, void* maps, void* R_out, void* stream) {
    (void)stream;
    if (!R_out) return CHX_ERR_INVALID_ARG;
    const int rc = build_scalars(kinds, param_ptrs, E, energy, mass_eV, n_charges, dtype, maps);
    if (rc != CHX_OK) return rc;
    double tm[49], m[49];
    eye7(tm);
    for (int64_t i = 0; i < E; ++i) {
        for (int k = 0; k < 49; ++k) m[k] = ld(maps, dtype, i * 49 + k);
        mm7(m, tm, tm);
    }
    for (int k = 0; k < 49; ++k) st(R_out, dtype, k, tm[k]);
    return CHX_OK;
}

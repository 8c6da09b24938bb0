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

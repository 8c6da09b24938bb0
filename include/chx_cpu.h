/* chx_cpu.h — host twins of the core libchx entry points ("each entry point has a *_cpu twin with
 * identical signature (host pointers)").
 *
 * For downstream projects that want to link and test their binding of the C-ABI on a machine WITHOUT a GPU: every function here
 * has the argument list of its include/chx.h namesake, takes host pointers, ignores `stream` / `workspace`, and runs the CPU
 * restatement of the reference (oracle/chx_oracle.c, compiled into oracle/libchx_cpu.so by oracle/chx_oracle.py
 * `build_cpu_abi()`). This is test / CI infrastructure like the oracle itself: the cheetah_amd package never loads it, and
 * there is no fallback from the GPU library to it.
 *
 *   chx_build_rmatrix_cpu   chx_build_rmatrix   track_methods.py:17-77,284-382, dipole.py:372-466, cavity.py:253-358
 *   chx_compose_maps_cpu    chx_compose_maps    segment.py:534-543
 *   chx_apply_affine7_cpu   chx_apply_affine7   element.py:180-191 (the kernels' fma chain: bit-identical results)
 *   chx_moments_cpu         chx_moments         particle_beam.py:1699-1717, utils/statistics.py:4-48
 *   chx_cic_deposit_cpu     chx_cic_deposit     utils/cloud_in_cell.py:8-451 (row-major grids only)
 *   chx_track_elementwise_cpu chx_track_elementwise segment.py:571-572 (`for e in elements: beam = e.track(beam)`: E fma-chain passes)
 *   chx_cavity_coeffs_cpu   chx_cavity_coeffs   cavity.py:113-122,135-226 (coefficient rows + outgoing energy)
 *   chx_cavity_track_cpu    chx_cavity_track    cavity.py:112,135-151,220-226 (x @ R.mT, then the per-particle delta / tau update)
 *   chx_hist2d_cpu          chx_hist2d          screen.py:292-311 (torch.histogramdd on explicit edges, weight |q| * survival)
 *   chx_sc_kick_cpu         chx_sc_kick         space_charge_kick.py:477-586 (the whole kick; radix-2 convolution in double)
 *   chx_track_fused_cpu     chx_track_fused     segment.py:571-572 (the element-by-element numbers from one pass)
 *   chx_apply_affine7_bwd_cpu chx_apply_affine7_bwd element.py:180-191 under autograd: dX = dY R, dR = sum_n dY_n x_n^T
 *   chx_moments_bwd_cpu / chx_moments_bwd_w_cpu  chx_moments_bwd(_w)  the cotangent of utils/statistics.py:4-62 (rows and weights)
 *   chx_cic_deposit_bwd_cpu chx_cic_deposit_bwd utils/cloud_in_cell.py under autograd: d / d(weights), d / d(positions)
 *   chx_sc_gather_kick_cpu  chx_sc_gather_kick  space_charge_kick.py:387-475, 548-584 (trilinear gather + kick from a force grid)
 *   chx_merge_moments_cpu   chx_merge_moments   the exact pooled statistics of R shards (Chan et al.; utils/statistics.py:4-62 on the union)
 *   chx_moment_entry_cpu, chx_moments_mapped_bwd_cpu, chx_moment_entry_mapped_bwd_cpu   their namesakes: a beam property
 *                           (particle_beam.py:1672-1943) of y = R x and its gradient with respect to R from the incoming beam's moments
 *   chx_build_rmatrix_scalars_cpu, chx_run_build_compose_cpu   their namesakes: a run's element maps from scalars read where they live,
 *                           and their product (segment.py:534-543)
 * The remaining declarations of chx.h are plans, fused stretches and tuning forms of these: they are exercised by the `-m gpu`
 * tests against the oracle directly and have no host twin.
 */
#ifndef CHX_CPU_H
#define CHX_CPU_H
#include "chx.h"
#ifdef __cplusplus
extern "C" {
#endif
int chx_abi_version_cpu(void);
int chx_build_rmatrix_cpu(int kind, const void* params, const void* energy, double mass_eV, double n_charges, int64_t B,
                          int64_t Bp, int64_t Be, int dtype, void* R_out, void* stream);
int chx_compose_maps_cpu(const void* const* R_ptrs, const uint8_t* bcast, int64_t E, int64_t B, int dtype, void* R_out,
                         void* stream);
int chx_apply_affine7_cpu(const void* x_in, const void* R, void* x_out, int64_t B, int64_t Bx, int64_t BR, int64_t N, int dtype,
                          void* stream);
int chx_moments_cpu(const void* x, const void* w, int64_t B, int64_t Bx, int64_t Bw, int64_t N, int dtype, double* out,
                    void* workspace, size_t workspace_bytes, void* stream);
int chx_cic_deposit_cpu(const chx_cic_args* args, void* stream);
int chx_track_elementwise_cpu(const void* x_in, const void* R /*[E][BR][7][7]*/, void* x_out,
                              void* scratch, int64_t E, int64_t B, int64_t Bx, int64_t BR,
                              int64_t N, int dtype, void* stream);
int chx_cavity_coeffs_cpu(const void* params /*[Bp][4]*/, const void* energy /*[Be]*/, double mass_eV,
                          double n_charges, int64_t B, int64_t Bp, int64_t Be, int dtype,
                          double* coeffs /*[B][CHX_CAV_NCOEF]*/, void* energy_out /*[B] dtype*/,
                          void* stream);
int chx_cavity_track_cpu(const void* x_in, const void* R, const double* coeffs, void* x_out,
                         int64_t B, int64_t Bx, int64_t N, int dtype, void* stream);
int chx_hist2d_cpu(const chx_hist2d_args* args, void* stream);
/* chx_sc_kick on host pointers (stream / side_stream / workspace ignored; the Poisson solve in double like the oracle's) */
size_t chx_sc_kick_workspace_bytes_cpu(int64_t B, int64_t N, const int32_t* bins, int dtype);
int chx_sc_kick_cpu(const void* x_in, const void* charge, const void* survival, const void* energy, const void* length,
                    const void* grid_extent, double mass_eV, int64_t B, int64_t Bx, int64_t Bq, int64_t Bs, int64_t Bext, int64_t N,
                    const int32_t* bins, int dtype, void* x_out, void* workspace, size_t workspace_bytes, void* stream,
                    void* side_stream, const void* post_map /*[BR][7][7] or NULL*/, int64_t BR);
int chx_track_fused_cpu(const void* x_in, const void* R /*[E][BR][7][7]*/, void* x_out, int64_t E,
                        int64_t B, int64_t Bx, int64_t BR, int64_t N, int dtype, void* stream);
size_t chx_apply_bwd_workspace_bytes_cpu(int64_t B, int64_t N);
int chx_apply_affine7_bwd_cpu(const void* dY, const void* R, const void* X, void* dX, double* dR,
                              int64_t B, int64_t Bx, int64_t BR, int64_t N, int dtype,
                              void* workspace, size_t workspace_bytes, void* stream);
int chx_moments_bwd_cpu(const void* x, const void* w, const double* out, const double* d_out,
                        int64_t B, int64_t Bx, int64_t Bw, int64_t N, int dtype, void* dX, void* stream);
int chx_moments_bwd_w_cpu(const void* x, const void* w, const double* out, const double* d_out, int64_t B, int64_t Bx, int64_t Bw,
                          int64_t N, int dtype, void* dX, void* dW, void* stream);
int chx_cic_deposit_bwd_cpu(const chx_cic_args* args, const void* dgrid, void* dweight /*[B][N]*/,
                            void* dpos /*[B][N][ndim]*/, void* stream);
int chx_sc_gather_kick_cpu(const void* x_in, const void* F, const void* half, const void* cell,
                           const void* energy, const void* dt, double mass_eV, int64_t B, int64_t Bx,
                           int64_t Be, int64_t N, const int32_t* bins, int dtype, void* x_out,
                           void* stream);
int chx_merge_moments_cpu(const double* per_rank, int32_t R, int64_t B, double* out, void* stream);
int chx_moments_mapped_bwd_cpu(const double* d_out, const void* R, const double* mom_x, int64_t B, int64_t BR, int64_t Bm,
                               int dtype, double* dR, void* stream);
int chx_moment_entry_cpu(const double* mom, int64_t B, int index, int take_sqrt, int dtype, void* out, void* stream);
int chx_moment_entry_mapped_bwd_cpu(const void* grad, const double* mom_y, int index, int take_sqrt, const void* R,
                                    const double* mom_x, int64_t B, int64_t BR, int64_t Bm, int dtype, void* dR, int dR_is_double,
                                    void* stream);
int chx_build_rmatrix_scalars_cpu(const int32_t* kinds, const void* const* param_ptrs, int64_t E, const void* energy,
                                  double mass_eV, double n_charges, int dtype, void* R_out, void* stream);
int chx_run_build_compose_cpu(const int32_t* kinds, const void* const* param_ptrs, int64_t E, const void* energy, double mass_eV,
                              double n_charges, int dtype, void* maps, void* R_out, void* stream);
#ifdef __cplusplus
}
#endif
#endif

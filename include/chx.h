/* chx.h — C-ABI of libchx, the MI355X (gfx950) beam-dynamics tracking engine.
 *
 * This is the drop-in boundary of the `Segment.track(ParticleBeam)` hot path of
 * desy-ml/cheetah. The reference has no FFI of its own (it is pure Python on
 * PyTorch); its operator interface is `Element.track` /
 * `Element.first_order_transfer_map` / `Segment.track` / `ParticleBeam`
 * (cheetah/accelerator/element.py:104-193, cheetah/accelerator/segment.py:534-574,
 * cheetah/particles/particle_beam.py:60-106). Every entry point below names the
 * reference function (file:line under /root/reference/cheetah) whose arithmetic it
 * replaces. `cheetah_amd/_lib.py` binds them with ctypes; INTEGRATION.md shows the
 * stub a Cheetah maintainer would add.
 *
 * Conventions (all entry points)
 *  - return CHX_OK (0) or a negative chx_status; nothing throws across the ABI;
 *  - every pointer is a DEVICE pointer owned by the caller, except `const int32_t*`
 *    / struct arguments which are HOST memory read before the call returns;
 *  - no allocation, no host synchronisation, no global state: work is enqueued on
 *    the caller's HIP stream (`stream` = hipStream_t, NULL = default stream) and the
 *    call returns immediately; calls on different streams are re-entrant;
 *  - dtype: CHX_F32 / CHX_F64 selects the element type of all `void*` floating
 *    buffers of that call; `double*` buffers are always fp64;
 *  - phase-space layout is the reference's: particles[B][N][7] row-major, row =
 *    (x, px, y, py, tau, delta, 1) (element.py:104-131), transfer maps [B][7][7]
 *    row-major with out_i = sum_j R[i][j] * in_j (element.py:182);
 *  - "B?" arguments (Bx, BR, Bw, ...) are broadcast extents: each must be 1 or B.
 *  - particle buffers must be 16-byte aligned (CHX_ERR_MISALIGNED otherwise).
 */
#ifndef CHX_H
#define CHX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CHX_ABI_VERSION 9 /* 2: ldz argument of chx_sc_igf / chx_sc_gradient; 3: post_map arguments of chx_sc_kick;
                             4: s_in / s_out arguments of chx_run_map / chx_run_track;
                             5: s_in / s_out arguments of chx_cavity_prepare_scalars / chx_cavity_track_scalars;
                             6: chx_lattice_track_diag (items of type 2 / 3 in the table of a lattice stretch);
                             7: chx_lattice_track_screens / chx_parameter_lattice_track_screens (items of type 4: active Screens);
                             8: chx_lattice_screen.mom_partials, chx_table_store; 9: chx_run_vjp_entry */

typedef enum chx_status {
    CHX_OK = 0,
    CHX_ERR_INVALID_ARG = -1,
    CHX_ERR_DTYPE = -2,
    CHX_ERR_MISALIGNED = -3,
    CHX_ERR_LAUNCH = -4,
    CHX_ERR_WORKSPACE = -5,
    CHX_ERR_NO_DEVICE = -6
} chx_status;

typedef enum chx_dtype { CHX_F32 = 0, CHX_F64 = 1 } chx_dtype;

/* Element kinds understood by chx_build_rmatrix; params layout [P] per batch row. */
typedef enum chx_kind {
    CHX_IDENTITY = 0,   /* P=0  Marker / inactive Screen, BPM, Aperture (marker.py:45-57) */
    CHX_DRIFT = 1,      /* P=1  [L]                                  track_methods.py:284-299 */
    CHX_QUADRUPOLE = 2, /* P=5  [L,k1,tilt,mis_x,mis_y]              quadrupole.py:93-110     */
    CHX_DIPOLE = 3,     /* P=9  [L,angle,k1,e1,e2,tilt,fint,fint_exit,gap]  dipole.py:372-394,430-466 */
    CHX_HCOR = 4,       /* P=2  [L,angle]                            horizontal_corrector.py:60-78 */
    CHX_VCOR = 5,       /* P=2  [L,angle]                            vertical_corrector.py:61-78 */
    CHX_CCOR = 6,       /* P=3  [L,hangle,vangle]                    combined_corrector.py:77-98 */
    CHX_CAVITY_SW = 7,  /* P=4  [L,voltage,phase_deg,frequency]      cavity.py:253-358 (standing wave) */
    CHX_CAVITY_TW = 8,  /* P=4  same, traveling wave                 cavity.py:310-335 */
    CHX_SOLENOID = 9,   /* P=4  [L,k,mis_x,mis_y]                    solenoid.py:75-116  (row f3) */
    CHX_UNDULATOR = 10, /* P=4  [L,kx,ky,period]                     undulator.py:79-125 (row f3) */
    CHX_KIND_COUNT = 11
} chx_kind;

#define CHX_MAX_PARAMS 9
/* number of params of a kind, or -1 */
int chx_kind_num_params(int kind);
int chx_abi_version(void);
/* human-readable status */
const char* chx_status_string(int status);

/* ---- map builders (a3-a8; track_methods.py:17-77,284-382; dipole.py:430-466; cavity.py:253-358)
 * Builds R_out[B][7][7] (dtype) from params[Bp][P] and energy[Be] (dtype). Arithmetic is
 * carried out in fp64 on device and rounded once. mass_eV / n_charges describe the Species
 * (species.py:30-36). */
int chx_build_rmatrix(int kind, const void* params, const void* energy, double mass_eV,
                      double n_charges, int64_t B, int64_t Bp, int64_t Be, int dtype,
                      void* R_out, void* stream);

/* Vector-Jacobian product of chx_build_rmatrix: given dR[B][7][7] (dtype) returns
 * dparams[B][P] and denergy[B] (dtype), d(sum dR*R)/d(param). Forward-mode dual numbers
 * inside the kernel; replaces autograd through track_methods.py / utils/autograd.py:77-146. */
int chx_build_rmatrix_vjp(int kind, const void* params, const void* energy, double mass_eV,
                          double n_charges, const void* dR, int64_t B, int64_t Bp, int64_t Be,
                          int dtype, void* dparams, void* denergy, void* stream);

/* Maps of E elements whose parameters are SCALARS, read where they live: kinds[E] and param_ptrs[E][CHX_MAX_PARAMS]
 * (host arrays; entry k of element e is a DEVICE pointer to one `dtype` value, in the order chx_build_rmatrix documents;
 * unused entries NULL), energy = device pointer to one value -> R_out[E][7][7]. Same arithmetic and rounding as E calls
 * of chx_build_rmatrix with B = 1; made for control loops that change a few settings and re-track (one call instead of a
 * packed tensor, an allocation and a launch per element). */
int chx_build_rmatrix_scalars(const int32_t* kinds, const void* const* param_ptrs, int64_t E, const void* energy,
                              double mass_eV, double n_charges, int dtype, void* R_out, void* stream);
/* Persistent map of a run of scalar-parameter elements (Segment.track's steady state, segment.py:545-574): `state`
 * (chx_run_state_bytes(E) bytes of device memory owned by the caller, filled with 0xFF bytes once) remembers the parameter
 * values, the element maps and the composed map R of the last call. chx_run_map launches ONE workgroup that re-reads every
 * parameter through its pointer and returns at once if none changed, else rebuilds and recomposes — bit-identical to
 * chx_build_rmatrix_scalars + chx_compose_maps. *R_out (if not NULL) receives the device address of R[7][7] inside `state`.
 * chx_run_track = chx_run_map + chx_apply_affine7(x_in, R, x_out) for one beam of N particles: a whole merged
 * Segment.track in one call, with no host-side validation of the settings. E <= 192 elements, <= 400 parameters in total
 * (CHX_ERR_INVALID_ARG beyond: use the two-call form). Identity elements are left out by the caller.
 * s_in / s_out (both NULL or both device pointers to one `dtype` value): the path length behind the run,
 * *s_out = *s_in + (((L_0 + L_1) + L_2) + ...) with the lengths read from the settings on every call (the first parameter of
 * every kind), like `incoming.s + segment.length` (segment.py:54-58): no host copy that an edited length could leave stale. */
/* Prefix products of a run: out[e][b] = maps[e][b] ... maps[1][b] maps[0][b] for every e (maps[E][Bm][7][7], Bm in {1, B};
 * out[E][B][7][7]); fp64 accumulation carried along, each prefix rounded once. With chx_track_moments on the E prefixes the
 * beam moments after every element of a lattice (segment.py:658-700 `get_beam_attrs_along_segment`) cost one pass over the
 * particles instead of E tracking passes and E reductions. */
int chx_compose_prefix(const void* maps, int64_t E, int64_t B, int64_t Bm, int dtype, void* out, void* stream);
/* Backward of chx_build_rmatrix_scalars + chx_compose_maps for a run whose settings carry gradients (autograd through
 * segment.py:534-574 for scalar parameters): given the element maps maps[E][7][7] of the forward call and dT[7][7] =
 * dL/d(composed map) (both `dtype`), writes dinputs[E][CHX_MAX_PARAMS + 1] (`dtype`): column k < P_e is dL/d(parameter k of
 * element e), column CHX_MAX_PARAMS is element e's contribution to dL/d(energy), the other columns are zero. Two launches
 * (a one-wave prefix / suffix sweep in fp64, then dual-number builders), workspace chx_run_vjp_workspace_bytes(E). */
size_t chx_run_vjp_workspace_bytes(int64_t E);
int chx_run_vjp(const int32_t* kinds, const void* const* param_ptrs, int64_t E, const void* energy, double mass_eV,
                double n_charges, int dtype, const void* maps, const void* dT, void* dinputs, void* workspace,
                size_t workspace_bytes, void* stream);
/* The same with a mask: need[E] (NULL = everything), bit k of need[e] set = slot k of element e is wanted (bit
 * CHX_MAX_PARAMS: the energy); unwanted slots are written as zeros without evaluating their dual-number builder (a loss on
 * ONE quadrupole strength of a 100-element run evaluates one builder, not 250). */
int chx_run_vjp_masked(const int32_t* kinds, const void* const* param_ptrs, int64_t E, const void* energy, double mass_eV,
                       double n_charges, int dtype, const void* maps, const void* dT, const uint16_t* need, void* dinputs,
                       void* workspace, size_t workspace_bytes, void* stream);
/* ABI 9. The same when dL/d(composed map) comes from ONE property of the beam y = C x the run's map produced — entry `index` (2..28)
 * of chx_moments(y) = mom_y[29], or its square root (`take_sqrt`), with the gradient grad[1] (`dtype`) — instead of a tensor dT:
 * every wave of the builders' launch forms chx_moment_entry_mapped_bwd's 49 values itself (C[7][7] `dtype`: the composed map,
 * mom_x[29]: the incoming beam's moments), rounded to `dtype` as that call writes them. The backward pass of
 * d sigma_x(screen) / d k1 (tests/test_differentiable.py:10-32; screen.py:187-239, particle_beam.py:1672-1943) is then ONE launch.
 * Runs of more than 16 elements fall back to the two launches. Workspace: chx_run_vjp_entry_workspace_bytes(E). */
size_t chx_run_vjp_entry_workspace_bytes(int64_t E);
int chx_run_vjp_entry(const int32_t* kinds, const void* const* param_ptrs, int64_t E, const void* energy, double mass_eV,
                      double n_charges, int dtype, const void* maps, const uint16_t* need, const void* grad, const double* mom_y,
                      int index, int take_sqrt, const void* C, const double* mom_x, void* dinputs, void* workspace,
                      size_t workspace_bytes, void* stream);
/* A merged run whose settings are VECTORISED over B lattice settings (segment.py:534-547 with (B,) parameters: a k1 scan, a batched
 * environment, an orbit response): all B composed maps R_out[B][7][7] in one launch — per row the element maps of chx_build_rmatrix
 * and the product of chx_compose_maps, bit-identical to those calls. batched[E][CHX_MAX_PARAMS]: 1 = the pointer addresses a
 * contiguous (B,) array of `dtype`, 0 = a scalar; every length (parameter 0) is a scalar; energy_rows != 0: `energy` is a (B,)
 * array as well (a scan of beam energies, with or without vectorised settings), else a scalar.
 * workspace: chx_run_map_batched_workspace_bytes(E, B, dtype) bytes (0 when a row's element maps fit the LDS). */
size_t chx_run_map_batched_workspace_bytes(int64_t E, int64_t B, int dtype);
int chx_run_map_batched(const int32_t* kinds, const void* const* param_ptrs, const uint8_t* batched, int64_t E, int64_t B,
                        const void* energy, int energy_rows, double mass_eV, double n_charges, int dtype, void* workspace,
                        size_t workspace_bytes, void* R_out, void* stream);
/* Forward of the same run in one call: chx_build_rmatrix_scalars into maps[E][7][7] (kept for chx_run_vjp) followed by
 * chx_compose_maps of that stack into R_out[7][7]; bit-identical to the two calls (segment.py:534-543). E <= 4096. */
int chx_run_build_compose(const int32_t* kinds, const void* const* param_ptrs, int64_t E, const void* energy, double mass_eV,
                          double n_charges, int dtype, void* maps, void* R_out, void* stream);
size_t chx_run_state_bytes(int64_t E);
int chx_run_map(const int32_t* kinds, const void* const* param_ptrs, int64_t E, const void* energy, double mass_eV,
                double n_charges, int dtype, void* state, size_t state_bytes, void** R_out, const void* s_in, void* s_out,
                void* stream);
int chx_run_track(const int32_t* kinds, const void* const* param_ptrs, int64_t E, const void* energy, double mass_eV,
                  double n_charges, int dtype, void* state, size_t state_bytes, const void* x_in, void* x_out, int64_t N,
                  const void* s_in, void* s_out, void* stream);
/* ---- segment composition (a2; segment.py:534-543): R_out[b] = R_{E-1}[b] ... R_1[b] R_0[b].
 * R_ptrs is a HOST array of E device pointers, one map buffer per element (each element owns
 * its cached map); buffer e is [1][7][7] if bcast[e] (HOST array) else [B][7][7]. The pointers are
 * forwarded by value in the kernel arguments (no device-side table, no copy). fp64 accumulation,
 * rounded once to dtype. */
int chx_compose_maps(const void* const* R_ptrs, const uint8_t* bcast, int64_t E, int64_t B,
                     int dtype, void* R_out, void* stream);
/* Backward of chx_compose_maps (autograd through segment.py:534-543 for vectorised maps): given dT[B][7][7] = dL/d(composed map)
 * writes dM[E][B][7][7] = dL/d(element map e of batch row b) (dtype; the caller sums the rows of a broadcast map). One wave per
 * batch row, fp64 prefix / suffix sweep. E <= 192; workspace chx_compose_maps_vjp_workspace_bytes(E, B). */
size_t chx_compose_maps_vjp_workspace_bytes(int64_t E, int64_t B);
int chx_compose_maps_vjp(const void* const* R_ptrs, const uint8_t* bcast, int64_t E, int64_t B, int dtype, const void* dT, void* dM,
                         void* workspace, size_t workspace_bytes, void* stream);

/* ---- linear apply (a1; element.py:180-191): x_out[b][n][:] = R[b] . x_in[b][n][:]. */
int chx_apply_affine7(const void* x_in, const void* R, void* x_out, int64_t B, int64_t Bx,
                      int64_t BR, int64_t N, int dtype, void* stream);

/* Backward of the apply: dX[b][n][j] = sum_i dY[b][n][i] R[b][i][j]  (dX may be NULL) and
 * dR[b][i][j] = sum_n dY[b][n][i] X[b][n][j] (double dR[B][49]; may be NULL; needs workspace
 * of chx_apply_bwd_workspace_bytes(B,N)). */
size_t chx_apply_bwd_workspace_bytes(int64_t B, int64_t N);
int chx_apply_affine7_bwd(const void* dY, const void* R, const void* X, void* dX, double* dR,
                          int64_t B, int64_t Bx, int64_t BR, int64_t N, int dtype,
                          void* workspace, size_t workspace_bytes, void* stream);

/* Element-by-element tracking of a run of E linear elements WITHOUT merging the maps
 * (`for e in elements: beam = e.track(beam)`, segment.py:571-572): E apply passes
 * pass 0 writes x_out, later passes update x_out in place (tile-local read-then-write); `scratch` is ignored (kept for
 * ABI stability, may be NULL). x_out must not alias x_in. */
int chx_track_elementwise(const void* x_in, const void* R /*[E][BR][7][7]*/, void* x_out,
                          void* scratch, int64_t E, int64_t B, int64_t Bx, int64_t BR,
                          int64_t N, int dtype, void* stream);

/* Same result as chx_track_elementwise (bit-identical per-element rounding) but one pass
 * over HBM: each particle stays in registers while the E maps are applied in order. */
int chx_track_fused(const void* x_in, const void* R /*[E][BR][7][7]*/, void* x_out, int64_t E,
                    int64_t B, int64_t Bx, int64_t BR, int64_t N, int dtype, void* stream);

/* ---- cavity (a8; cavity.py:100-251).
 * coefficients per batch row, computed on device (no host branch: the reference's
 * `if (delta_energy > 0).any()` (cavity.py:157) is evaluated over the batch in-kernel). */
#define CHX_CAV_NCOEF 8 /* [a, b, kbeta0, phi, cosphi, T566, T556, T555] */
int chx_cavity_coeffs(const void* params /*[Bp][4]*/, const void* energy /*[Be]*/, double mass_eV,
                      double n_charges, int64_t B, int64_t Bp, int64_t Be, int dtype,
                      double* coeffs /*[B][CHX_CAV_NCOEF]*/, void* energy_out /*[B] dtype*/,
                      void* stream);
/* x_out = R.x_in, then delta' and tau' per particle (cavity.py:135-151,220-226). */
int chx_cavity_track(const void* x_in, const void* R, const double* coeffs, void* x_out,
                     int64_t B, int64_t Bx, int64_t N, int dtype, void* stream);

/* Backward of the per-particle part of chx_cavity_track that is not the matrix apply: given dY = dL/d(x_out), adds
 * the delta' / tau' rewrite's contribution to dX[B][N][7] (columns tau, delta; dX may be NULL) and writes
 * dcoeffs[B][CHX_CAV_NCOEF] = dL/d(coeffs). The matrix part is chx_apply_affine7_bwd with row 5 of R zeroed.
 * workspace: chx_moments_workspace_bytes(B, N). */
int chx_cavity_track_bwd(const void* dY, const void* X, const double* coeffs, void* dX, double* dcoeffs, int64_t B,
                         int64_t Bx, int64_t N, int dtype, void* workspace, size_t workspace_bytes, void* stream);

/* ---- weighted beam moments (a10; particle_beam.py:1672-1943, utils/statistics.py:4-62).
 * sums[b] = { W=sum w, sum w^2, sum w x_0..5 }  (8 doubles)
 * m2[b]   = upper triangle (a<=b, row-major, 21 doubles) of sum w (x_a-mu_a)(x_b-mu_b),
 *           mu = sums[2..7]/sums[0] read from `sums` on device (multi-GPU: all-reduce sums first)
 * out[b]  = { W, W2, mu[6], cov[21] } with cov = m2 / (W - W2/W)          (29 doubles)
 * w may be NULL (all ones). Deterministic two-stage reductions (no float atomics). */
#define CHX_MOM_NSUMS 8
#define CHX_MOM_NM2 21
#define CHX_MOM_NOUT 29
size_t chx_moments_workspace_bytes(int64_t B, int64_t N);
int chx_moment_sums(const void* x, const void* w, int64_t B, int64_t Bx, int64_t Bw, int64_t N,
                    int dtype, double* sums, void* workspace, size_t workspace_bytes, void* stream);
int chx_moment_centred(const void* x, const void* w, const double* sums, int64_t B, int64_t Bx,
                       int64_t Bw, int64_t N, int dtype, double* m2, void* workspace,
                       size_t workspace_bytes, void* stream);
int chx_moment_finalize(const double* sums, const double* m2, int64_t B, double* out, void* stream);
/* convenience: the three calls above on one stream */
int chx_moments(const void* x, const void* w, int64_t B, int64_t Bx, int64_t Bw, int64_t N,
                int dtype, double* out, void* workspace, size_t workspace_bytes, void* stream);
/* chx_moments plus ONE entry of the vector in the beam dtype — what a beam property reads (particle_beam.py:1672-1943: mu_*,
 * sigma_* = sqrt of the variance, cov_*): entry_out[B] (dtype) = out[b][index] or its square root, written by the same finalize
 * launch (index < 0: plain chx_moments). */
int chx_moments_entry(const void* x, const void* w, int64_t B, int64_t Bx, int64_t Bw, int64_t N, int dtype, double* out,
                      int index, int take_sqrt, void* entry_out, void* workspace, size_t workspace_bytes, void* stream);
/* Multi-GPU: exact merge of the chx_moments outputs of R particle shards, per_rank[R][B][29] -> out[B][29] (the reference's
 * weighted statistics over the union of the shards, utils/statistics.py:30-48; shards with zero weight are skipped). */
int chx_merge_moments(const double* per_rank, int32_t R, int64_t B, double* out, void* stream);
/* backward of chx_moments wrt x: given d_out[B][29] (double; entries 0,1 ignored) */
int chx_moments_bwd(const void* x, const void* w, const double* out, const double* d_out,
                    int64_t B, int64_t Bx, int64_t Bw, int64_t N, int dtype, void* dX, void* stream);
/* The same with the gradient of the WEIGHTS as well (utils/statistics.py:4-62 is differentiable in the survival
 * probabilities: an Aperture with trainable edges upstream): dW[B][N] (dtype; the caller sums the rows of a broadcast weight
 * array), dX and dW may each be NULL. d_out entries 0 (W) and 1 (sum w^2) are honoured here. */
int chx_moments_bwd_w(const void* x, const void* w, const double* out, const double* d_out, int64_t B, int64_t Bx, int64_t Bw,
                      int64_t N, int dtype, void* dX, void* dW, void* stream);
/* Backward of chx_moments(y), y_n = R x_n, with respect to the MAP R[BR][7][7] (dtype) when the particles x carry no
 * gradient: mu' = A mu + b, cov' = A C A^T (element.py:180-191 + utils/statistics.py:4-62), so
 * dR[B][7][7] (double) = [2 G A C + g_mu mu^T | g_mu; 0] from d_out[B][29] and the INCOMING beam's chx_moments
 * mom_x[Bm][29] alone — no pass over the particles (tests/test_differentiable.py:10-32: d sigma_x(screen) / d k1). */
int chx_moments_mapped_bwd(const double* d_out, const void* R, const double* mom_x, int64_t B, int64_t BR, int64_t Bm,
                           int dtype, double* dR, void* stream);
/* One entry of the moment vector as a beam property reads it (particle_beam.py:1672-1943: mu_*, sigma_* = sqrt of the
 * variance, cov_*): out[b] (dtype) = mom[b][index] or its square root — and its backward for a linearly tracked beam in one
 * launch: grad[B] (dtype) of that entry -> dR[B][7][7] (double, or dtype when dR_is_double = 0) through
 * chx_moments_mapped_bwd's algebra; mom_y[B][29] = chx_moments of the tracked beam (for the square root). */
int chx_moment_entry(const double* mom, int64_t B, int index, int take_sqrt, int dtype, void* out, void* stream);
int chx_moment_entry_mapped_bwd(const void* grad, const double* mom_y, int index, int take_sqrt, const void* R,
                                const double* mom_x, int64_t B, int64_t BR, int64_t Bm, int dtype, void* dR, int dR_is_double,
                                void* stream);
/* Fused observables: out[b] = chx_moments of the tracked beam R[b] x without ever
 * writing it (element.py:180-191 followed by particle_beam.py:1672-1943), e.g. sigma_x(k1) over a scan of B
 * settings on one shared beam. One pass; second moments are accumulated about c_b = R[b] centre, where
 * centre[Bx][6] (double, device; e.g. the incoming means from chx_moment_sums) must lie within a few sigma of the
 * incoming mean (NULL = origin). y = R x is evaluated exactly like chx_apply_affine7 (same fma chain).
 * With a shared beam (Bx == Bw == 1, BR == B >= 64) each lane owns one batch row: no reduction across lanes. */
size_t chx_track_moments_workspace_bytes(int64_t B, int64_t N);
int chx_track_moments(const void* x_in, const void* w, const void* R, const double* centre, int64_t B, int64_t Bx,
                      int64_t BR, int64_t Bw, int64_t N, int dtype, double* out, void* workspace,
                      size_t workspace_bytes, void* stream);

/* ---- cloud-in-cell deposition (a12; utils/cloud_in_cell.py:8-451) and Screen images
 * (a11; screen.py:241-344). Positions are read straight out of the particle array:
 * pos_d = (x[b][n][col[d]] * scale[b][d]) - shift[b][d], each step rounded in `dtype`
 * exactly like the reference's tensor ops (screen.py:200-212 misalignment subtraction,
 * particle_beam.py:1333 z = tau * -beta). Weight = charge * survival (|charge| if abs_charge).
 * Index arithmetic follows cloud_in_cell.py:150-211 in `dtype`, bit-exact. */
typedef struct chx_cic_args {
    int32_t ndim;            /* 1..3 */
    int32_t cols[3];         /* column of the 7-vector used for each axis */
    int32_t bins[3];         /* grid shape */
    int64_t grid_strides[3]; /* element strides of the output grid per axis (row-major default if 0) */
    int64_t grid_batch_stride;
    int64_t B, Bx, Bq, Bs, Be, Bsc, Bsh, N;
    int32_t dtype;
    int32_t abs_charge;
    const void* x;        /* [Bx][N][7] */
    const void* charge;   /* [Bq][N] or NULL (=1) */
    const void* survival; /* [Bs][N] or NULL (=1) */
    const void* extent;   /* [Be][ndim][2] (left,right) */
    const void* scale;    /* [Bsc][ndim] or NULL */
    const void* shift;    /* [Bsh][ndim] or NULL */
    void* grid;           /* [B] grids, dtype, accumulated into (caller zeroes) */
} chx_cic_args;
int chx_cic_deposit(const chx_cic_args* args, void* stream);
/* Fused "track, then deposit" (SURVEY section 8 row f2): the position of particle n in batch row b is taken from R[b] x
 * (R[BR][7][7], BR in {1, B}; coordinate cols[d] of the product, the fma chain of chx_apply_affine7 — bit-identical to
 * chx_apply_affine7 followed by chx_cic_deposit) without the (B, N, 7) tracked array ever being written: Screen images
 * for a scan of B lattice settings over one shared beam (element.py:180-191 + screen.py:327-339). */
int chx_cic_deposit_mapped(const chx_cic_args* args, const void* R, int64_t BR, void* stream);
/* Same result as chx_cic_deposit (identical addends, different summation order) for ndim 2 or 3 and large
 * N: particles are counting-sorted by grid tile, accumulated in LDS (ds_add) and flushed once per tile,
 * instead of 2^ndim global float atomics per particle (which saturate at ~21 G atomics/s on MI355X). */
size_t chx_cic_sorted_workspace_bytes(const chx_cic_args* args);
int chx_cic_deposit_sorted(const chx_cic_args* args, void* workspace, size_t workspace_bytes, void* stream);
/* The same deposit with grid = deposit instead of grid += deposit: every cell of every grid row is stored by the tile
 * that owns it (zeros included), so the caller does not zero the grid first and the flush carries no dependent load.
 * Used by chx_sc_kick (space_charge_kick.py:556-563 deposits into a fresh zero tensor). */
int chx_cic_deposit_sorted_overwrite(const chx_cic_args* args, void* workspace, size_t workspace_bytes, void* stream);
/* Test/diagnostic twin: writes the integer cell index i_d=floor(p_d) (int32 [B][N][ndim])
 * and fractional part f_d (dtype [B][N][ndim]) instead of depositing. */
int chx_cic_indices(const chx_cic_args* args, int32_t* idx_out, void* frac_out, void* stream);
/* Backward of the deposit wrt charges and positions (cloud_in_cell.py is differentiable):
 * dq[b][n] = sum_corners dgrid*weight ; dpos[b][n][d] via d(weight)/d(f_d) / bin_width. */
int chx_cic_deposit_bwd(const chx_cic_args* args, const void* dgrid, void* dweight /*[B][N]*/,
                        void* dpos /*[B][N][ndim]*/, void* stream);

/* Screen "histogram" method (screen.py:292-311 -> torch.histogramdd): bin j iff
 * e_j <= v < e_{j+1}, last bin right-inclusive, outside dropped; edges from torch.linspace. */
typedef struct chx_hist2d_args {
    int64_t B, Bx, Bq, Bs, Bsh, N;
    int32_t nx, ny; /* number of bins; edges have nx+1 / ny+1 entries */
    int32_t dtype;
    const void* x;        /* [Bx][N][7]; x = col 0, y = col 2 */
    const void* charge;   /* |charge| * survival is the weight */
    const void* survival;
    const void* shift;    /* [Bsh][2] screen misalignment or NULL */
    const void* edges_x;  /* [nx+1] dtype */
    const void* edges_y;  /* [ny+1] dtype */
    void* image;          /* [B][ny][nx] dtype (already transposed like screen.py:311), caller zeroes */
} chx_hist2d_args;
int chx_hist2d(const chx_hist2d_args* args, void* stream);
int chx_hist2d_indices(const chx_hist2d_args* args, int32_t* ij_out /*[B][N][2], -1 = dropped*/,
                       void* stream);

/* Derivatives of the space-charge kick (the reference differentiates space_charge_kick.py with torch autograd;
 * tests/test_space_charge_kick.py:202-327). Together with chx_cic_deposit_bwd, chx_moments' backward and the
 * self-adjoint chx_sc_convolve they make SpaceChargeKick.track differentiable on power-of-two grids.
 *  - chx_sc_igf_table_grad: tables[3][B][(gx+1)(gy+1)(gz+1)] (double) = d(corner table of chx_sc_igf_table) /
 *    d(cell_x, cell_y, cell_z * gamma); feeding each through chx_sc_green_spectrum + chx_sc_convolve gives
 *    d phi / d(that cell size), so dL/dcell = sum(dphi * that).
 *  - chx_sc_gradient_bwd: dphi[B][gx][gy][gz] = adjoint of chx_sc_gradient (compact phi) applied to dF[B][g^3][4].
 *  - chx_sc_gather_kick_bwd: dx[B][N][7] (may be NULL); dF[B][g^3][4] (may be NULL; ACCUMULATED with atomics — the
 *    caller zero-fills); partials (may be NULL) = chx_sc_gather_kick_bwd_partials_count() doubles
 *    [B][ceil(N/256)][8] = per-workgroup sums of dY . d x_out / d (half[3], cell[3], dt, energy). */
int chx_sc_igf_table_grad(const void* cell, const void* gamma, int64_t B, const int32_t* bins, int dtype, double* tables,
                          void* stream);
int chx_sc_gradient_bwd(const void* dF, const void* cell, const void* gamma, int64_t B, const int32_t* bins, int dtype,
                        void* dphi, void* stream);
int64_t chx_sc_gather_kick_bwd_partials_count(int64_t B, int64_t N);
int chx_sc_gather_kick_bwd(const void* x_in, const void* F, const void* half, const void* cell, const void* energy,
                           const void* dt, const void* dY, double mass_eV, int64_t B, int64_t Bx, int64_t Be, int64_t N,
                           const int32_t* bins, int dtype, void* dx, void* dF, double* partials, void* stream);

/* Screen "kde" method (screen.py:312-326, utils/kde.py:4-77): Gaussian kernel values of particles n0 .. n0 + nchunk - 1
 * against the bin centres,  out[B][nchunk][nbins] = max(w exp(-((v - c_i) / sigma)^2 / 2) / sqrt(2 pi sigma^2), tiny),
 * v = x[..][col] - shift (col 0: x, col 2: y), w = |charge| * survival (either may be NULL = 1). The joint density is
 * the GEMM of two such arrays over the particle axis, left to the BLAS library by the host layer. */
int chx_kde_values(const void* x, const void* charge, const void* survival, const void* shift, const void* centres,
                   const void* sigma, int col, int64_t B, int64_t Bx, int64_t Bq, int64_t Bs, int64_t Bsh, int64_t N,
                   int64_t n0, int64_t nchunk, int32_t nbins, int dtype, void* out, void* stream);

/* ---- space charge (a13; space_charge_kick.py:103-586, particle_beam.py:1262-1346) */
/* Integrated Green function on the doubled grid (space_charge_kick.py:163-291).
 * cell[B][3] (dtype) = cell sizes (hx,hy,htau); gamma[B] (dtype). fp64 inside; the workspace
 * holds the (gx+1)(gy+1)(gz+1) corner table of the primitive per batch row. */
size_t chx_sc_igf_workspace_bytes(int64_t B, const int32_t* bins);
/* ldz = distance between rows of the last axis of G_out: 2gz (0 = default), or 2gz + 2 for the padded layout of
 * an in-place real-to-complex transform (chx_sc_fft_exec). */
int chx_sc_igf(const void* cell, const void* gamma, int64_t B, const int32_t* bins, int dtype,
               void* G_out /*[B][2gx][2gy][ldz]*/, int64_t ldz, void* workspace, size_t workspace_bytes,
               void* stream);
/* The same doubled array from a corner table that already exists (chx_sc_igf_table, or one of the three derivative tables of
 * chx_sc_igf_table_grad — the backward pass of the Poisson stage on grids outside chx_sc_pruned_supported). */
int chx_sc_igf_from_table(const double* table, int64_t B, const int32_t* bins, int dtype, void* G_out, int64_t ldz, void* stream);
/* 3-D FFTs of the Hockney convolution (space_charge_kick.py:306-314) through hipFFT, in place on the padded real
 * layout [B][2gx][2gy][2gz + 2] (= complex [B][2gx][2gy][gz + 1]), unnormalised. direction 0 / 1: forward with
 * plan 0 / 1 (two plans, so rho and the Green function can be transformed concurrently on two streams),
 * 2: inverse. Plan creation is a host-side call that may allocate hipFFT work areas; exec only enqueues. */
int chx_sc_fft_plan_create(int64_t B, const int32_t* bins, int dtype, void** plan_out);
int chx_sc_fft_plan_destroy(void* plan);
int chx_sc_fft_exec(void* plan, int direction, void* data, void* stream);
/* Pruned, symmetry-aware Poisson solve (csrc/chx_fft.hip) for power-of-two grids (chx_sc_pruned_supported):
 *  - chx_sc_igf_table: the (gx+1)(gy+1)(gz+1) corner table of the Green-function primitive (double);
 *  - chx_sc_green_spectrum: real, even spectrum of the integrated Green function on the doubled grid, stored on
 *    (gx+1)(gy+1)(gz+1) points (the spectrum at index k and 2g-k is the same number);
 *  - chx_sc_convolve: phi[B][gx][gy][gz] = crop(ifft(fft(pad(rho)) * Ghat * scale[b])) from the COMPACT rho[B][gx][gy][gz]:
 *    the zero padding is implicit (the zero halves of the lines are never read) and only the first octant of the
 *    result is ever written; transforms unnormalised (fold 1/(8 gx gy gz) into scale). */
int chx_sc_pruned_supported(const int32_t* bins, int dtype);
int chx_sc_igf_table(const void* cell, const void* gamma, int64_t B, const int32_t* bins, int dtype, double* table,
                     void* stream);
size_t chx_sc_green_workspace_bytes(int64_t B, const int32_t* bins, int dtype);
/* chx_sc_igf_table + chx_sc_green_spectrum in one call from cell[B][3] / gamma[B] (dtype). fp32 grids evaluate the
 * 6-transcendental primitive (space_charge_kick.py:103-123) only for cells closer than 8 x the largest cell edge and use
 * the 4th-order multipole expansion of the cell integral of 1/r beyond (relative error <= 1e-8, below fp32 rounding);
 * fp64 grids are exact everywhere. */
size_t chx_sc_green_fast_workspace_bytes(int64_t B, const int32_t* bins, int dtype);
int chx_sc_green_spectrum_fast(const void* cell, const void* gamma, int64_t B, const int32_t* bins, int dtype, void* Ghat,
                               void* workspace, size_t workspace_bytes, void* stream);
int chx_sc_green_spectrum(const double* table, int64_t B, const int32_t* bins, int dtype, void* Ghat, void* workspace,
                          size_t workspace_bytes, void* stream);
size_t chx_sc_convolve_workspace_bytes(int64_t B, const int32_t* bins, int dtype);
int chx_sc_convolve(const void* rho, const void* Ghat, const double* scale, int64_t B, const int32_t* bins, int dtype,
                    void* phi, void* workspace, size_t workspace_bytes, void* stream);
/* The same convolution with the potential stored inside a halo of 2 nodes: phi_halo[B][gx+4][gy+4][gz+4] (dtype,
 * chx_sc_phi_halo_elements() elements), node (i, j, k) at [i+2][j+2][k+2]. The halo itself is NOT written; its content is
 * undefined and chx_sc_gather_kick_phi reads it only where the value is discarded. Workspace as chx_sc_convolve. */
size_t chx_sc_phi_halo_elements(int64_t B, const int32_t* bins);
int chx_sc_convolve_halo(const void* rho, const void* Ghat, const double* scale, int64_t B, const int32_t* bins, int dtype,
                         void* phi_halo, void* workspace, size_t workspace_bytes, void* stream);
/* The same convolution (space_charge_kick.py:262-322: FFT of the padded charge, product with the Green spectrum, inverse FFT,
 * crop), for a Green spectrum that another stream is still computing: `ghat_ready_event` (a hipEvent_t recorded on that
 * stream behind the spectrum, or NULL) is waited for on `stream` in front of the pass that first reads Ghat — the forward
 * x and y passes of rho do not wait. */
int chx_sc_convolve_halo_after(const void* rho, const void* Ghat, const double* scale, int64_t B, const int32_t* bins, int dtype,
                               void* phi_halo, void* workspace, size_t workspace_bytes, void* stream, void* ghat_ready_event);
/* Grid geometry of a kick from the beam moments (space_charge_kick.py:531-550,110-130), one launch instead of ~25
 * tensor ops: moments[Bm][29] (chx_moments layout), grid_extent[Bext][3] (in sigmas), energy[Be], length[Bl] ->
 * half[B][3] = extent*sigma, cell[B][3] = 2 half / bins, gamma[B], dt[B] = L/(c beta), scale[B][3] = (1, 1, -beta),
 * extent[B][3][2] = (-half, half) (all `dtype`, rounded step by step like the reference's tensor expressions) and
 * pot_scale[B] (double) = pot_factor / prod(cell). */
int chx_sc_geometry(const double* moments, const void* grid_extent, const void* energy, const void* length,
                    double mass_eV, double pot_factor, int64_t B, int64_t Bm, int64_t Bext, int64_t Be, int64_t Bl,
                    const int32_t* bins, int dtype, void* half, void* cell, void* gamma, void* dt, void* scale,
                    void* extent, double* pot_scale, void* stream);
/* rho_hat *= G_hat * scale[b]  (complex multiply; space_charge_kick.py:313-316);
 * n_complex = complex elements per batch row; scale (double[B]) folds 1/(4 pi eps0), 1/cell volume and the FFT
 * normalisation. */
int chx_sc_spectral_mul(void* rho_hat, const void* G_hat, const double* scale, int64_t B,
                        int64_t n_complex, int dtype, void* stream);
/* E+vxB force field from the potential (space_charge_kick.py:324-365): central differences, x -1/gamma^2;
 * phi is the doubled array phi[B][2gx][2gy][ldz] (cropped on the fly, phi_doubled = 1; ldz = 0 means 2gz) or the
 * compact phi[B][gx][gy][gz] (phi_doubled = 0); F_out[B][gx][gy][gz][4]. */
int chx_sc_gradient(const void* phi, const void* cell, const void* gamma, int64_t B,
                    const int32_t* bins, int phi_doubled, int64_t ldz, int dtype, void* F_out, void* stream);
/* Fused: to_xyz_pxpypz -> trilinear node-based gather -> p += F*dt -> from_xyz_pxpypz
 * (space_charge_kick.py:387-475,548-584; particle_beam.py:1262-1346).
 * half[B][3] grid half-widths, cell[B][3], energy[Be], dt[B] (all dtype). */
int chx_sc_gather_kick(const void* x_in, const void* F, const void* half, const void* cell,
                       const void* energy, const void* dt, double mass_eV, int64_t B, int64_t Bx,
                       int64_t Be, int64_t N, const int32_t* bins, int dtype, void* x_out,
                       void* stream);
/* chx_sc_gather_kick followed by chx_apply_affine7 with post_map[BR][7][7] (BR in {1, B}) in ONE pass: the linear run that
 * follows a SpaceChargeKick inside a Segment (segment.py:545-574) is applied while the kicked particle is still in registers
 * (the kicked coordinates are rounded to dtype first, then the apply kernel's fma chain: bit-identical to the two passes).
 * chx_sc_kick takes the same optional map as its last arguments (NULL = kick only). */
int chx_sc_gather_kick_mapped(const void* x_in, const void* F, const void* half, const void* cell, const void* energy,
                              const void* dt, double mass_eV, int64_t B, int64_t Bx, int64_t Be, int64_t N, const int32_t* bins,
                              int dtype, const void* post_map, int64_t BR, void* x_out, void* stream);
/* chx_sc_gradient + chx_sc_gather_kick(_mapped) in ONE pass, without the force grid: every particle takes the central
 * differences of chx_sc_gradient (space_charge_kick.py:324-385; zero normal component on the boundary nodes) on the 32
 * potential values around its cell — phi_halo from chx_sc_convolve_halo, gamma[B] and cell[B][3] as for chx_sc_gradient —
 * in the same arithmetic and order, then interpolates and kicks: bit-identical to the two-kernel form, which writes and
 * re-reads a [g^3][4] force grid four times the size of the potential. post_map may be NULL (kick only). */
int chx_sc_gather_kick_phi(const void* x_in, const void* phi_halo, const void* half, const void* cell, const void* gamma,
                           const void* energy, const void* dt, double mass_eV, int64_t B, int64_t Bx, int64_t Be, int64_t N,
                           const int32_t* bins, int dtype, const void* post_map, int64_t BR, void* x_out, void* stream);
/* chx_moments + chx_sc_geometry for the kick: only the three variances the grid needs (sigma_x, sigma_y, sigma_tau;
 * space_charge_kick.py:531-538) are accumulated (8 fp64 sums per lane instead of 29, same shifted one-pass formulas and
 * rounding), and the partial sums are finalised inside the geometry kernel: two launches. Outputs as chx_sc_geometry. */
size_t chx_sc_beam_geometry_workspace_bytes(int64_t B, int64_t N);
int chx_sc_beam_geometry(const void* x, const void* w, const void* grid_extent, const void* energy, const void* length,
                         double mass_eV, double pot_factor, int64_t B, int64_t Bx, int64_t Bw, int64_t Bext, int64_t Be,
                         int64_t Bl, int64_t N, const int32_t* bins, int dtype, void* half, void* cell, void* gamma, void* dt,
                         void* scale, void* extent, double* pot_scale, void* workspace, size_t workspace_bytes, void* stream);
/* A whole SpaceChargeKick.track in ONE call (space_charge_kick.py:477-586) for grids chx_sc_pruned_supported() accepts:
 * chx_sc_beam_geometry -> [side stream: chx_sc_green_spectrum_fast] -> chx_cic_deposit_sorted_overwrite (N >= 65536; zero +
 * chx_cic_deposit below) -> chx_sc_convolve_halo -> chx_sc_gather_kick_phi, all intermediates in `workspace`
 * (chx_sc_kick_workspace_bytes). x_in[Bx][N][7], charge[Bq][N], survival[Bs][N], energy[B], length[B],
 * grid_extent[Bext][3] (in sigmas) -> x_out[B][N][7]. `side_stream` may be NULL (everything on `stream`); two events are
 * created and destroyed per call, nothing else is allocated. Saves ~20 foreign-function calls per kick: at the
 * reference's default 32^3 grid the host, not the GPU, was the limit. */
size_t chx_sc_kick_workspace_bytes(int64_t B, int64_t N, const int32_t* bins, int dtype);
int chx_sc_kick(const void* x_in, const void* charge, const void* survival, const void* energy, const void* length,
                const void* grid_extent, double mass_eV, int64_t B, int64_t Bx, int64_t Bq, int64_t Bs, int64_t Bext, int64_t N,
                const int32_t* bins, int dtype, void* x_out, void* workspace, size_t workspace_bytes, void* stream,
                void* side_stream, const void* post_map /*[BR][7][7] or NULL*/, int64_t BR);
/* A CHAIN of kicks inside one Segment.track on a tile-ordered beam (csrc/chx_sc_tiles.h; space_charge_kick.py:477-586 per kick):
 * between two kicks the particles barely move against the grid (its extent follows the beam sigmas), so the counting sort by
 * 8^3 deposit tile is done by the FIRST kick only; every kick then deposits straight from the ordered rows (LDS block per tile,
 * +1 layer handed to the neighbours through per-tile face buffers) and gathers with the potential block of each tile staged in
 * LDS; particles that left the tile of their slot are handled exactly on a slow path (float atomics / global loads), and when more
 * than 1/16 of the beam is misfiled the gather of that kick writes its rows in the new tile order (device-side decision, no host
 * synchronisation, no extra launch). B = 1 (one beam), grids as chx_sc_kick.
 *  - state: chx_sc_tile_state_bytes() bytes that live as long as the chain (header, tile starts, permutation, ordered
 *    weights / charges, row buffer, face buffers, the misfiled particles' grid); workspace: per-kick scratch, chx_sc_kick_sorted_workspace_bytes();
 *  - flags: CHX_SC_FIRST (1) x_in, charge, survival are the caller's arrays in the caller's order — sort; otherwise x_in is the
 *    x_out of the previous kick of the chain (possibly mapped through linear elements) and charge / survival are ignored;
 *    CHX_SC_LAST (2) x_out is written in the caller's particle order (else in tile order);
 *    CHX_SC_INDEX(i) (bits 8..): the kick's position in its chain, 0 for the first. With it (fp32 beams, grid edges up to 128 along
 *    x) the kicks from the second one on launch NO one-workgroup kernels: the grid geometry is formed inside the deposit and the
 *    Green function's corner-table launches from sums the previous kick's gather pass adds up with fp64 atomics, and the deposit's
 *    bookkeeping rides in the convolution's first FFT pass (csrc/chx_sc_geom_dev.h: 13 launches per kick instead of 15). The
 *    sums' last bits depend on the order of the atomics, like the charge grid's. Without an index (0) every kick launches the
 *    geometry and bookkeeping kernels. An indexed kick must not be handed beam_moments (chx_sc_kick_sorted_begin);
 *  - post_map (may be NULL): the linear run behind the kick, applied in the same particle pass (as chx_sc_kick).
 * The pieces are entry points of their own: chx_sc_beam_geometry_tiles (= chx_sc_beam_geometry + the header update),
 * chx_sc_tile_sort, chx_sc_tile_deposit (rho[gx][gy][gz], every cell stored: no zero fill), chx_sc_tile_gather_kick. */
#define CHX_SC_FIRST 1
#define CHX_SC_LAST 2
#define CHX_SC_INDEX(i) ((int)(i) << 8)
size_t chx_sc_tile_state_bytes(int64_t N, const int32_t* bins, int dtype);
size_t chx_sc_kick_sorted_workspace_bytes(int64_t N, const int32_t* bins, int dtype);
int chx_sc_kick_sorted(const void* x_in, const void* charge, const void* survival, const void* energy, const void* length,
                       const void* grid_extent, double mass_eV, int64_t N, const int32_t* bins, int dtype, void* x_out,
                       void* workspace, size_t workspace_bytes, void* state, size_t state_bytes, int flags, void* stream,
                       void* side_stream, const void* post_map);
/* chx_sc_kick_sorted in two halves, for a beam whose particles are spread over several GPUs (one process per GPU,
 * cheetah_amd/sharding.py; the reference's kick, space_charge_kick.py:477-586, needs the beam sizes of ALL particles :531-538 and
 * the charge of ALL particles on the grid :556-563): the two exchanges sit between the halves and the rows never leave the
 * tile order.
 *  begin : grid geometry — from beam_moments when non-NULL (chx_moments layout; moment_rows = 0: the 29 moments of the WHOLE
 *          beam, moment_rows = R > 0: the [R][29] moments of its shards exactly as the ranks all-gathered them, merged inside the
 *          geometry kernel with chx_merge_moments' arithmetic), else from this process's own rows as chx_sc_kick_sorted does — then
 *          [side stream: Green spectrum], [CHX_SC_FIRST: tile sort], tile deposit. *rho_out = the compact charge grid
 *          [gx][gy][gz] (dtype) inside `state` (the chain's accumulation grid): sum it over the ranks in place.
 *  finish: convolution (joins the side stream) -> gather + kick (+ post_map) into x_out.
 * Same workspace / state / flags / streams for both halves; the workspace must not be touched in between.
 * chx_sc_tile_beam_moments: this process's share of the moments the NEXT kick's grid needs — the sums the last
 * gather pass left in `state`, as a chx_moments row (entries of x, y, tau; zeros elsewhere) for the all-gather + merge. */
int chx_sc_kick_sorted_begin(const void* x_in, const void* charge, const void* survival, const void* energy, const void* length,
                             const void* grid_extent, double mass_eV, int64_t N, const int32_t* bins, int dtype, void* workspace,
                             size_t workspace_bytes, void* state, size_t state_bytes, int flags, const double* beam_moments,
                             int32_t moment_rows, void** rho_out, void* stream, void* side_stream);
int chx_sc_kick_sorted_finish(const void* x_in, const void* energy, double mass_eV, int64_t N, const int32_t* bins, int dtype,
                              void* x_out, void* workspace, size_t workspace_bytes, void* state, size_t state_bytes, int flags,
                              void* stream, void* side_stream, const void* post_map);
int chx_sc_tile_beam_moments(const void* state, size_t state_bytes, int64_t N, const int32_t* bins, int dtype, double* moments_out,
                             void* stream);
/* partials[8][nblk] (the sums of chx_sc_tile_gather_kick: W, W2, sum w x / y / tau, sum w x^2 / y^2 / tau^2 about the origin)
 * -> one chx_moments row [29] with the x, y, tau means and variances filled (statistics.py:30-48), zeros elsewhere */
int chx_sc_partials_moments(const double* partials, int64_t nblk, double* moments_out, void* stream);
/* chx_sc_geometry (B = 1 in a chain) + the header update of a chain of tile-ordered kicks (tile_header = start of the chain's
 * state for every kick but the first, NULL otherwise) */
int chx_sc_geometry_tiles(const double* moments, const void* grid_extent, const void* energy, const void* length, double mass_eV,
                          double pot_factor, int64_t B, int64_t Bm, int64_t Bext, int64_t Be, int64_t Bl, const int32_t* bins,
                          int dtype, void* half, void* cell, void* gamma, void* dt, void* scale, void* extent, double* pot_scale,
                          void* tile_header, int32_t merge_rows /*0, or B = 1 and moments[merge_rows][29] = shards to merge*/, void* stream);
int chx_sc_beam_geometry_tiles(const void* x, const void* w, const void* grid_extent, const void* energy, const void* length,
                               double mass_eV, double pot_factor, int64_t B, int64_t Bx, int64_t Bw, int64_t Bext, int64_t Be,
                               int64_t Bl, int64_t N, const int32_t* bins, int dtype, void* half, void* cell, void* gamma, void* dt,
                               void* scale, void* extent, double* pot_scale, void* workspace, size_t workspace_bytes,
                               void* tile_header, int tile_first, void* stream);
/* the geometry kernel of chx_sc_beam_geometry alone, from partial sums partials[8][nblk] (double) that chx_sc_tile_gather_kick
 * accumulated over the rows it wrote (B = 1) */
int chx_sc_geometry_from_partials(const double* partials, int64_t nblk, const void* grid_extent, const void* energy,
                                  const void* length, double mass_eV, double pot_factor, const int32_t* bins, int dtype, void* half,
                                  void* cell, void* gamma, void* dt, void* scale, void* extent, double* pot_scale, void* tile_header,
                                  void* stream);
int chx_sc_tile_sort(const void* x_in, const void* charge, const void* survival, const void* extent, const void* scale, int64_t N,
                     const int32_t* bins, int dtype, void* state, size_t state_bytes, void* stream);
/* rows: the tile-ordered rows (NULL = the state's own row buffer, which chx_sc_tile_sort filled); allow_reorder: may the crosser
 * pass order a re-sort by this kick's gather (0 for the last kick of a chain, which restores the caller's order instead). */
int chx_sc_tile_deposit(const void* rows, const void* extent, const void* scale, int64_t N, const int32_t* bins, int dtype,
                        void* state, size_t state_bytes, void* grid, int allow_reorder, void* stream);
/* chx_sc_tile_deposit with the charge LEFT in the chain's accumulation grid (*acc_out: [gx][gy][gz] of dtype inside `state`, all
 * zero between two kicks): what chx_sc_kick_sorted uses. Whoever consumes the grid must leave zeros behind —
 * chx_sc_convolve_halo_consume (= chx_sc_convolve_halo_after whose first FFT pass writes zeros behind its loads) does. */
int chx_sc_tile_deposit_acc(const void* rows, const void* extent, const void* scale, int64_t N, const int32_t* bins, int dtype,
                            void* state, size_t state_bytes, int allow_reorder, void** acc_out, void* stream);
int chx_sc_convolve_halo_consume(void* rho, const void* Ghat, const double* scale, int64_t B, const int32_t* bins, int dtype,
                                 void* phi_halo, void* workspace, size_t workspace_bytes, void* stream, void* ghat_ready_event);
int chx_sc_tile_gather_kick(const void* rows, const void* phi_halo, const void* half, const void* cell, const void* gamma,
                            const void* energy, const void* dt, double mass_eV, int64_t N, const int32_t* bins, int dtype,
                            const void* post_map, void* state, size_t state_bytes, int unpermute, void* x_out, void* stream);
/* A STRETCH of lattice — [run of skippable elements | active Cavity]+ with scalar device settings, ONE beam — in two launches:
 * what Segment.track's element-by-element walk (segment.py:545-574; cavity.py:100-251 per cavity) does with two launches and
 * ~8 us of host time per item. Launch 1 (chx_lattice_prepare, a workgroup per item): every run's composed map, every cavity's
 * map and coefficient row for the reference energy it receives (walked through the cavities in front of it), the outgoing
 * energy and path length. Launch 2: every particle through all items in registers. Bit-identical to tracking the items one by
 * one with chx_run_track / chx_cavity_track_scalars.
 *  table (device, int64 words): items[n_items][4] = {0 run | 1 cavity, elements E, first element, 0} (| 2, 3: see
 *    chx_lattice_track_diag), elem_kind[n_elems],
 *    elem_poff[n_elems] (index of the element's first pointer), ptrs[n_ptrs] (device addresses of the settings, each kind's
 *    parameters in chx_build_rmatrix order; a cavity: length, voltage, phase, frequency);
 *  state: chx_lattice_state_bytes(n_items, n_elems) bytes of device scratch that belong to the plan;
 *  energy, energy_out, s_in, s_out: one value of `dtype` each (s_in / s_out may both be NULL). */
size_t chx_lattice_state_bytes(int64_t n_items, int64_t n_elems);
int chx_lattice_prepare(const int64_t* table, int64_t n_items, int64_t n_elems, int64_t n_ptrs, const void* energy, double mass_eV,
                        double n_charges, int dtype, void* state, size_t state_bytes, void* energy_out, const void* s_in, void* s_out,
                        void* stream);
int chx_lattice_track(const int64_t* table, int64_t n_items, int64_t n_elems, int64_t n_ptrs, const void* energy, double mass_eV,
                      double n_charges, int dtype, void* state, size_t state_bytes, const void* x_in, void* x_out, int64_t N,
                      void* energy_out, const void* s_in, void* s_out, void* stream);
/* The same stretch with ACTIVE beam position monitors and apertures in it — elements that read or thin the beam and let the
 * particles pass, each of which costs the element-by-element walk (segment.py:545-574) a stop and several launches:
 *  {2, 0, q, slot}: a BPM (bpm.py:77-87): ptrs[q] = address of its misalignment [2]; readings[slot] = (dtype)(sum w x / sum w,
 *     sum w y / sum w) - misalignment of the beam AT that point (fp64 sums per wave in the particle pass, one more launch). In a big
 *     float32 scan (Bx = 1, Bm = B >= 8, B * N >= 8e6, rows on 16-byte boundaries, one row of weights) a monitor with nothing but maps
 *     and monitors in front of it is evaluated by taking the beam's weighted mean through the row's maps in fp64 instead — the same
 *     number up to the per-item float32 rounding of the individual particles, which averages out (~1e-10 of the beam size);
 *  {3, 0 rectangular | 1 elliptical, q, 0}: an aperture (aperture.py:90-135): ptrs[q], ptrs[q + 1] = addresses of x_max, y_max;
 *     survival *= inside(x, y), the arithmetic of chx_aperture_mask; monitors behind it weigh with the reduced probabilities.
 * B beams of N particles each (x_in / x_out [B][N][7]: a vectorised ParticleBeam under ONE lattice setting and energy — the same
 * maps for all beams; chx_lattice_track: B = 1). Bx = 1: x_in is ONE beam [N][7] shared by the B rows; Bm = B: the lattice
 * settings are vectorised over the rows (tagged addresses, state of chx_lattice_state_bytes_batched(.., Bm); small_runs as for
 * chx_lattice_prepare_rows): row b of the output = the beam through row b of the settings. survival: [Bw][N] (Bw = 1: one row of weights for
 * all rows), survival_out [B][N] of `dtype` or NULL (= 1); survival_out: [B][N], required when
 * the stretch holds an aperture, else NULL; readings[n_bpm][B][2]; workspace: chx_lattice_diag_workspace_bytes(N, B, n_bpm)
 * bytes. Particles, energy and path length are those of chx_lattice_track bit for bit. */
size_t chx_lattice_diag_workspace_bytes(int64_t N, int64_t B, int64_t n_bpm);
int chx_lattice_track_diag(const int64_t* table, int64_t n_items, int64_t n_elems, int64_t n_ptrs, const void* energy, double mass_eV,
                           double n_charges, int dtype, void* state, size_t state_bytes, const void* x_in, void* x_out, int64_t N,
                           int64_t B, int64_t Bx, int64_t Bm, int64_t Bw, int small_runs, void* energy_out, const void* s_in,
                           void* s_out, const void* survival, void* survival_out, int64_t n_bpm, void* readings, void* workspace,
                           size_t workspace_bytes, void* stream);
/* The same stretch with ACTIVE SCREENS in it (screen.py:187-239: the screen records a copy of the beam that reaches it and lets
 * the beam pass; screen.py:241-344: its image). An RL control step is `Segment.track` on a small beam followed by the screen's
 * reading: element by element that is the merged track, a copy of five tensors, a memset and the deposit — four launches and
 * ~50 us of host time around ~10 us of work. Here the screen is an item of the stretch:
 *  {4, flags, q, slot}: ptrs[q] = address of its misalignment [2], ptrs[q + 1] = address of its pixel_size [2] (both `dtype`),
 *     ptrs[q + 2 .. q + 5] = the integers resolution_x, resolution_y, bins_x, bins_y (bins = resolution // binning);
 *     flags bit 0: deposit the cloud-in-cell image (screen.py:327-339) in the particle pass.
 * screens[slot] (HOST array, read before the call returns) holds this call's output buffers of the screen:
 *  rows [N][7], charges [N], survival [N]: the record of the beam AT the screen (the coordinates UNSHIFTED: the misalignment is
 *     subtracted where the image is formed and by the caller's `get_read_beam`), written by the particle pass; each may be NULL;
 *  energy, s: one value of `dtype` each — the reference energy and path length at the screen, written by the preparation launch;
 *  image [bins_y][bins_x] (`dtype`; NULL = none) of image_bytes bytes: ZEROED by the preparation launch (spare workgroups: no
 *     memset launch), then every particle adds |charge| * survival to its four pixels with the index arithmetic of
 *     chx_cic_deposit on (x - misalignment_x, y - misalignment_y) and the extent (-+ resolution * pixel_size / 2) evaluated from
 *     the pixel size as torch evaluates screen.py:139-148 — the same cells, fractions and addends as
 *     chx_apply_affine7 + chx_cic_deposit on the recorded rows (float atomics: the same sum to the order of the additions);
 *  ParameterBeam variant (chx_parameter_lattice_track_screens): mu [7], cov [49] = the moments at the screen; image
 *     [height][width] = chx_screen_gaussian of them (geom: [left, hstep, bottom, vstep], shift: the misalignment; one more launch).
 * One beam, scalar settings (B = Bx = Bm = Bw = 1) when n_screens > 0; `charge` [N] (may be NULL = 1) feeds the image weights
 * and the record. */
#define CHX_LATTICE_MAX_SCREENS 4
typedef struct chx_lattice_screen {
    void* rows;
    void* charges;
    void* survival;
    void* energy;
    void* s;
    void* image;
    int64_t image_bytes;
    void* map;              /* [7][7] (`dtype`) or NULL: the composed map of the RUN right in front of the screen, as the particle
                               pass applies it (what a differentiable caller hangs the screen's beam properties on) */
    void* element_maps;     /* [E][7][7] (`dtype`) or NULL: that run's element maps in tracking order (what chx_run_vjp_masked takes) */
    /* ParameterBeam variant only */
    void* mu;
    void* cov;
    const void* geom;
    const void* shift;      /* the screen's misalignment [2] (`dtype`, device) */
    const void* total_charge; /* the beam's total charge (one value of `dtype`) and where the record's copy of it goes */
    void* total_charge_out;
    int32_t width, height;
    /* ParticleBeam variant, round 6 (ABI 8): the particle pass also adds the one-pass sums of the recorded beam's 29 moments
     * (particle_beam.py:1699-1717; utils/statistics.py:4-62) — about the beam's first row as it stands at the screen — into
     * mom_partials: CHX_LATTICE_MOMENT_DOUBLES doubles = n sets (n = chx_lattice_moment_blocks(N, B) <= 64) of {W, W2, s[6], m[21]} as
     * [29][n], the centre[6] behind them; the preparation launch zeroes the buffer — or NULL. chx_lattice_screen_moments
     * adds the sets and finalises: a beam property of the screen's beam costs one small launch, no pass over its rows.
     * Not together with `image` in one call. */
    void* mom_partials;
} chx_lattice_screen;
#define CHX_LATTICE_MOMENT_DOUBLES (29 * 64 + 6)
/* the sets of partial sums the particle pass of chx_lattice_track_screens uses for N particles x B beams (at most 64 per beam) */
int64_t chx_lattice_moment_blocks(int64_t N, int64_t B);
/* mom_partials (n_blocks = chx_lattice_moment_blocks sets) -> out[29] = {W, W2, mu[6], cov[21]} (what chx_moments gives for the
 * recorded rows, to the rounding of another summation order) and, index >= 0, entry_out[1] (`dtype`) = out[index] or its square
 * root (chx_moments_entry). */
int chx_lattice_screen_moments(const double* mom_partials, int64_t n_blocks, int dtype, double* out, int index, int take_sqrt,
                               void* entry_out, void* stream);
/* (ABI 9: B > 1 beams with screens when Bm == 1 — records [B][N][7] / [B][N] / [B][N], images [B][height][width], `charge` one row of N
 * shared by the beams; mom_partials only with B == 1.) */
int chx_lattice_track_screens(const int64_t* table, int64_t n_items, int64_t n_elems, int64_t n_ptrs, const void* energy,
                              double mass_eV, double n_charges, int dtype, void* state, size_t state_bytes, const void* x_in,
                              void* x_out, int64_t N, int64_t B, int64_t Bx, int64_t Bm, int64_t Bw, int small_runs,
                              void* energy_out, const void* s_in, void* s_out, const void* survival, void* survival_out,
                              int64_t n_bpm, void* readings, void* workspace, size_t workspace_bytes, const void* charge,
                              const chx_lattice_screen* screens, int64_t n_screens, void* stream);
/* A stretch's table (the `table` argument of the chx_lattice_* calls: device memory) written from HOST words without a staging
 * buffer — the words travel in the arguments of a one-workgroup launch on `stream`, in front of the preparation launch that reads
 * them: n <= chx_table_store_max_words(). For a host whose control loop assigns new setting tensors every step (README.md:73-77):
 * the table's layout stands, a few addresses change. */
int64_t chx_table_store_max_words(void);
int chx_table_store(const int64_t* host_words, int64_t n, void* table, void* stream);
/* chx_lattice_prepare_rows that also writes screens[slot].energy / .s and zeroes screens[slot].image (rows = 1 when n_screens > 0) */
int chx_lattice_prepare_screens(const int64_t* table, int64_t n_items, int64_t n_elems, int64_t n_ptrs, int64_t rows, int small_runs,
                                const void* energy, double mass_eV, double n_charges, int dtype, void* state, size_t state_bytes,
                                void* energy_out, const void* s_in, void* s_out, const chx_lattice_screen* screens,
                                int64_t n_screens, void* stream);
/* out[4] (`dtype`, device) = the extent (left, right, bottom, top) the stretch kernels derive from pixel_size[2] (`dtype`, device)
 * and the resolution — for checking it bit for bit against the reference's tensor expression (screen.py:139-148) */
int chx_screen_extent(const void* pixel_size, int32_t resolution_x, int32_t resolution_y, int dtype, void* out, void* stream);
/* The same stretch for a ParameterBeam (element.py:167-179, cavity.py:127-133,202-218, bpm.py:77-87): mu [Bmu][7], cov [Bcov][49]
 * (Bmu, Bcov in {1, B}) through [run | active Cavity | active BPM]+ by one wavefront per batch row after the same preparation
 * launch — mu' = R mu, cov' = R cov R^T item by item (fp64 inside, rounded to `dtype` between items like chx_parameter_track), a
 * cavity's moment updates, readings[n_bpm][B][2] = (mu_x, mu_y) - misalignment at every monitor. Items of type 3 are skipped. */
/* Bm = rows of lattice settings (1, or B: the settings are vectorised — ptrs entries with the lowest bit set address (B,) arrays of
 * `dtype`, like chx_run_map_batched; lengths, cavity settings and the energy stay scalars): row b of the beam goes through row b of
 * the lattice; state: chx_lattice_state_bytes_batched(n_items, n_elems, Bm) bytes (chx_lattice_prepare_batched fills it). */
size_t chx_lattice_state_bytes_batched(int64_t n_items, int64_t n_elems, int64_t rows);
int chx_lattice_prepare_batched(const int64_t* table, int64_t n_items, int64_t n_elems, int64_t n_ptrs, int64_t rows, const void* energy,
                                double mass_eV, double n_charges, int dtype, void* state, size_t state_bytes, void* energy_out,
                                const void* s_in, void* s_out, void* stream);
/* small_runs is a set of flags. Bit 0 (CHX_LATTICE_SMALL_RUNS): the caller vouches that the stretch holds no cavity and no run of
 * more than 64 elements — a wave per (item, row) prepares the maps instead of a workgroup (300 000 of them for 75 items x 4096
 * rows); the same results. Bit 1 (CHX_LATTICE_ENERGY_ROWS): `energy` and `energy_out` are (rows,) arrays — a scan of BEAM ENERGIES,
 * row r of the maps, coefficient rows and outgoing energies belongs to energy r (cavity.py:113-122 per row; the second-order
 * path-length switch of cavity.py:157 depends on the cavity's settings only, which are scalars here). The same flags travel through
 * the `small_runs` argument of chx_lattice_track_diag and chx_parameter_lattice_track. Bit 2 (CHX_LATTICE_ENERGY_OUT_ROWS):
 * `energy_out` alone is a (rows,) array — a cavity whose voltage or phase is vectorised (tagged addresses) hands on one energy per
 * row; cavity.py:157's `(delta_energy > 0).any()` is then taken over the rows of that cavity. */
#define CHX_LATTICE_SMALL_RUNS 1
#define CHX_LATTICE_ENERGY_ROWS 2
#define CHX_LATTICE_ENERGY_OUT_ROWS 4
/* bit 3: the caller vouches that every run holds at most 64 elements (cavities allowed): with rows > 1 a wave per (item, row) prepares
 * the stretch, cavity arithmetic included (a 16-cell linac at 64 energies: 2048 workgroups of four waves otherwise) */
#define CHX_LATTICE_SHORT_RUNS 8
int chx_lattice_prepare_rows(const int64_t* table, int64_t n_items, int64_t n_elems, int64_t n_ptrs, int64_t rows, int small_runs,
                             const void* energy, double mass_eV, double n_charges, int dtype, void* state, size_t state_bytes,
                             void* energy_out, const void* s_in, void* s_out, void* stream);
int chx_parameter_lattice_track(const int64_t* table, int64_t n_items, int64_t n_elems, int64_t n_ptrs, const void* energy,
                                double mass_eV, double n_charges, int dtype, void* state, size_t state_bytes, const void* mu,
                                const void* cov, int64_t B, int64_t Bmu, int64_t Bcov, int64_t Bm, int small_runs, void* mu_out,
                                void* cov_out, void* energy_out, const void* s_in, void* s_out, int64_t n_bpm, void* readings,
                                void* stream);
int chx_parameter_lattice_track_screens(const int64_t* table, int64_t n_items, int64_t n_elems, int64_t n_ptrs, const void* energy,
                                        double mass_eV, double n_charges, int dtype, void* state, size_t state_bytes, const void* mu,
                                        const void* cov, int64_t B, int64_t Bmu, int64_t Bcov, int64_t Bm, int small_runs, void* mu_out,
                                        void* cov_out, void* energy_out, const void* s_in, void* s_out, int64_t n_bpm, void* readings,
                                        const chx_lattice_screen* screens, int64_t n_screens, void* stream);
/* Cavity.track (cavity.py:100-251) for ONE beam and a cavity whose four settings are device scalars of `dtype`:
 * param_ptrs[4] = device pointers to length, voltage, phase [deg], frequency; energy = device pointer to one value;
 * kind = CHX_CAVITY_SW / CHX_CAVITY_TW. chx_cavity_prepare_scalars writes the map R_out[7][7] (dtype, as chx_build_rmatrix),
 * the coefficient row coeffs[CHX_CAV_NCOEF] (as chx_cavity_coeffs) and the outgoing energy with one thread of one launch;
 * chx_cavity_track_scalars adds the particle pass of chx_cavity_track: the whole element in one call, two launches,
 * bit-identical to chx_build_rmatrix + chx_cavity_coeffs + chx_cavity_track. */
int chx_cavity_prepare_scalars(const void* const* param_ptrs, const void* energy, int kind, double mass_eV, double n_charges,
                               int dtype, void* R_out, double* coeffs, void* energy_out, const void* s_in /*may be NULL*/,
                               void* s_out /*NULL iff s_in is: *s_out = *s_in + length*/, void* stream);
size_t chx_cavity_track_scalars_workspace_bytes(void);
int chx_cavity_track_scalars(const void* x_in, const void* const* param_ptrs, const void* energy, int kind, double mass_eV,
                             double n_charges, int64_t N, int dtype, void* x_out, void* energy_out, const void* s_in,
                             void* s_out, void* workspace, size_t workspace_bytes, void* stream);
/* n <= 8 device arrays copied by one launch: dst[k][0 .. bytes[k]) = src[k][...] (host arrays of device pointers and byte
 * counts). What a Screen's record of the incoming beam costs (screen.py:190 `incoming.clone()`: five tensors). */
int chx_copy_arrays(const void* const* src, void* const* dst, const int64_t* bytes, int32_t n, void* stream);
/* SI conversion on its own (particle_beam.py:1262-1346), used by to_xyz_pxpypz/from_xyz_pxpypz */
int chx_to_xyz_pxpypz(const void* x_in, const void* energy, double mass_eV, int64_t B, int64_t Bx,
                      int64_t Be, int64_t N, int dtype, void* xp_out, void* stream);
int chx_from_xyz_pxpypz(const void* xp_in, const void* energy, double mass_eV, int64_t B,
                        int64_t Bx, int64_t Be, int64_t N, int dtype, void* x_out, void* stream);

/* ---- ParameterBeam path (element.py:167-179, cavity.py:127-133,202-218,
 * screen.py:255-291). mu[Bmu][7], cov[Bcov][7][7], R[BR][7][7]: mu' = R mu, cov' = R cov R^T (fp64 inside).
 * With cavity_coeffs (double [B][CHX_CAV_NCOEF] from chx_cavity_coeffs) the reference's cavity moment
 * updates are applied on top (mu_tau, mu_delta, cov_tautau, cov_taudelta, cov_deltadelta). */
int chx_parameter_track(const void* mu, const void* cov, const void* R, const double* cavity_coeffs, int64_t B,
                        int64_t Bmu, int64_t Bcov, int64_t BR, int dtype, void* mu_out, void* cov_out,
                        void* stream);
/* Backward of the linear part (no cavity coefficients): g_mu[B][7], g_cov[B][7][7] (dtype; either may be NULL = zero) ->
 * d_mu[B][7] = R^T g, d_cov[B][7][7] = R^T G R, d_R[B][7][7] = g mu^T + G R cov^T + G^T R cov, one row per batch row (the
 * caller sums the rows of a broadcast input); each output may be NULL. What autograd derives from element.py:167-179
 * (tests/test_differentiable.py:58-75). */
int chx_parameter_track_bwd(const void* g_mu, const void* g_cov, const void* mu, const void* cov, const void* R, int64_t B,
                            int64_t Bmu, int64_t Bcov, int64_t BR, int dtype, void* d_mu, void* d_cov, void* d_R,
                            void* stream);
/* Screen reading of a ParameterBeam: bivariate normal density of (x - shift_x, y - shift_y) sampled at
 * (left + ix*hstep, bottom + iy*vstep); geom = [left, hstep, bottom, vstep] (dtype, device);
 * positions_fp32 != 0 rounds the sample positions to fp32 (the reference's torch.arange grid is created in
 * torch's default dtype, screen.py:284-287); image[B][height][width]. */
int chx_screen_gaussian(const void* mu, const void* cov, const void* shift, const void* geom, int64_t B,
                        int64_t Bmu, int64_t Bcov, int64_t Bsh, int32_t width, int32_t height,
                        int positions_fp32, int dtype, void* image, void* stream);

/* ---- non-linear per-particle tracking.
 * Drift-kick-drift (Bmad-X) tracking: drift.py:106-154, quadrupole.py:168-251, dipole.py:183-370,
 * transverse_deflecting_cavity.py:122-209 over utils/bmadx.py. Cheetah -> Bmad canonical coordinates,
 * the element's exact/symplectic map, and back, in one pass; column 6 of x_out is 1.
 * params[Bp][P] (dtype):
 *   CHX_DKD_DRIFT       [length]
 *   CHX_DKD_QUADRUPOLE  [length, k1, tilt, misalignment_x, misalignment_y]   (+ num_steps)
 *   CHX_DKD_DIPOLE      [length, angle, e1, e2, tilt, fint, fint_exit, gap, gap_exit]
 *                       (+ fringe_at: bit 0 = entrance, bit 1 = exit; linear_edge fringe)
 *   CHX_DKD_TDC         [length, voltage, phase, frequency, tilt, misalignment_x, misalignment_y]
 * energy_out[B] (may be NULL) receives the reference energy recomputed from p0c (bmadx.py:49). */
enum chx_dkd_kind { CHX_DKD_DRIFT = 0, CHX_DKD_QUADRUPOLE = 1, CHX_DKD_DIPOLE = 2, CHX_DKD_TDC = 3,
                    CHX_DKD_LINEAR = 4 /* only in chx_dkd_chain_mixed / chx_dkd_energy_chain: a merged run of linear elements */ };
int chx_dkd_num_params(int kind);
int chx_dkd_track(int kind, const void* x_in, const void* params, const void* energy, double mass_eV,
                  double n_charges, int32_t num_steps, int32_t fringe_at, int64_t B, int64_t Bx, int64_t Bp,
                  int64_t Be, int64_t N, int dtype, void* x_out, void* energy_out, void* stream);
/* The same with the arithmetic width as an argument: storage_precision = 0 evaluates the per-particle map in float64 whatever
 * the storage dtype (chx_dkd_track: agreement with Bmad-X to 1e-14 in float64, ~1e-7 of a coordinate in float32, but fp64-VALU
 * bound: 20-42 us per element at 1e6 particles); 1 evaluates float32 beams in float32 like the reference's own tensor code
 * (cheetah/utils/bmadx.py runs in the beam dtype) — one HBM-bound pass. float64 beams ignore the flag. */
int chx_dkd_track_p(int kind, const void* x_in, const void* params, const void* energy, double mass_eV, double n_charges,
                    int32_t num_steps, int32_t fringe_at, int64_t B, int64_t Bx, int64_t Bp, int64_t Be, int64_t N, int dtype,
                    int storage_precision, void* x_out, void* energy_out, void* stream);
/* A run of E drift-kick-drift elements (a lattice tracked with tracking_method = "drift_kick_drift": every element is its own,
 * non-mergeable map, e.g. quadrupole.py:174-240) on one beam x_in[N][7] with scalar settings: HOST arrays kinds[E],
 * params[E] (device pointers to each element's parameter array), num_steps[E], fringe_at[E], storage_precision[E]; the
 * reference energy travels from element to element on the device (energies[E] (dtype): what each element leaves; energy_in one
 * scalar). x_tmp[N][7] is scratch (may be NULL for E = 1); x_out receives the last element's particles. Drifts, Quadrupoles and
 * Dipoles (float32 beams: of one storage_precision): two launches per 320 elements, the particles in registers across the run (x_tmp
 * holds the elements' constants); otherwise E launches. Either way one call and the bits of E separate chx_dkd_track_p calls.
 * s_in / s_out (one scalar of dtype each, both or neither): the path length, s_out = (((s_in + l_0) + l_1) + ...) with the
 * lengths added one by one in dtype like the reference's `s=incoming.s + self.length` per element. */
int chx_dkd_chain(const int32_t* kinds, const void* const* params, const int32_t* num_steps, const int32_t* fringe_at,
                  const int32_t* storage_precision, int64_t E, const void* x_in, const void* energy_in, double mass_eV,
                  double n_charges, int64_t N, int dtype, void* x_out, void* x_tmp, void* energies, const void* s_in, void* s_out,
                  void* stream);
/* The same run with merged runs of linear elements in between (a lattice whose drifts are tracked linearly and whose magnets with
 * the Bmad-X maps; segment.py:545-574 tracks such a run with its composed map): kinds[e] = CHX_DKD_LINEAR marks params[e] as a
 * [7][7] first-order map (dtype), applied with the arithmetic of chx_apply_affine7, and lengths[e] as the device scalar holding
 * that run's summed length (lengths may be NULL when no item is linear; the entry of a drift-kick-drift element may be NULL: its
 * first parameter). The reference energy passes a linear run unchanged. One pass over the beam for the whole stretch. */
int chx_dkd_chain_mixed(const int32_t* kinds, const void* const* params, const void* const* lengths, const int32_t* num_steps,
                        const int32_t* fringe_at, const int32_t* storage_precision, int64_t E, const void* x_in, const void* energy_in,
                        double mass_eV, double n_charges, int64_t N, int dtype, void* x_out, void* x_tmp, void* energies,
                        const void* s_in, void* s_out, void* stream);
/* energies[E] (dtype) = the reference energy behind every item of such a run without tracking anything (a caller that builds the
 * maps of the linear runs needs the energy in front of each of them). kinds_scratch: E int32 of device memory. One or two launches. */
int chx_dkd_energy_chain(const int32_t* kinds, int64_t E, const void* energy_in, double mass_eV, int dtype, void* energies,
                         void* kinds_scratch, void* stream);
/* Backward of chx_dkd_track (the reference gets it from torch autograd through utils/bmadx.py): forward-mode dual
 * numbers on device, one seeded evaluation per input. dx[B][N][7] (dtype, may be NULL) = dY . d x_out / d x_in;
 * partials (may be NULL) = chx_dkd_bwd_partials_count() doubles laid out [B][ceil(N/256)][P + 1]: per workgroup
 * sums of dY . d x_out / d theta_k (k < P: params, k = P: energy) — the caller sums axis 1 (deterministic; no
 * atomics). The gradient of energy_out with respect to energy is the identity. */
int64_t chx_dkd_bwd_partials_count(int kind, int64_t B, int64_t N);
int chx_dkd_track_bwd(int kind, const void* x_in, const void* params, const void* energy, const void* dY,
                      double mass_eV, double n_charges, int32_t num_steps, int32_t fringe_at, int64_t B, int64_t Bx,
                      int64_t Bp, int64_t Be, int64_t N, int dtype, void* dx, double* partials, void* stream);

/* The reference's singularity-free special functions (utils/autograd.py:4-74, the torch.autograd.Function pairs at
 * :77-700) element-wise over n values, with their partial derivatives: out[i] = f(a[i] (, b[i])), da[i] = df/da,
 * db[i] = df/db (da / db may be NULL; b is ignored by the one-argument kinds). Evaluated in double, stored as dtype.
 *   LOG1PDIV log(1+x)/x            SI1MDIV (1 - si sqrt x)/x           SICOS1MDIV (1 - si sqrt x cos sqrt x)/x
 *   SIPSICOS3MDIV (3 - 4 si + si cos)/(2x)                             SICOSKUDDELMUDDEL15MDIV (utils/autograd.py:305-308)
 *   COSSQRTMCOSDIVDIFF (cos sqrt b - cos sqrt a)/(a - b)               SIMSIDIVDIFF (si sqrt a - si sqrt b)/(b - a)
 *   SI2MSI2DIVDIFF (si^2 sqrt b - si^2 sqrt a)/(a - b)                 SQRTA2MINUSBDIVA (sqrt(a^2 + b) - a)/b
 * with si x = sin x / x, negative arguments continued through the hyperbolic forms, and the limits at x = 0, a = b,
 * b = 0 the reference substitutes. */
enum chx_special_kind {
    CHX_SP_LOG1PDIV = 0, CHX_SP_SI1MDIV = 1, CHX_SP_SICOS1MDIV = 2, CHX_SP_SIPSICOS3MDIV = 3,
    CHX_SP_SICOSKUDDELMUDDEL15MDIV = 4, CHX_SP_COSSQRTMCOSDIVDIFF = 5, CHX_SP_SIMSIDIVDIFF = 6,
    CHX_SP_SI2MSI2DIVDIFF = 7, CHX_SP_SQRTA2MINUSBDIVA = 8
};
int chx_special(int kind, const void* a, const void* b, int64_t n, int dtype, void* out, void* da, void* db,
                void* stream);

/* Second-order tracking (element.py:195-228): x_out_i = sum_jk T_ijk x_j x_k with the MAD-convention
 * tensors of track_methods.py:80-296 (base_ttensor), the first-order map filled into T[:, 6, :] and the
 * element's rotations / misalignments / fringes folded in (drift.py:68-84, quadrupole.py:113-146,
 * dipole.py:397-428, sextupole.py:91-116). params[Bp][P] (dtype):
 *   CHX_T_DRIFT [length]   CHX_T_QUADRUPOLE [length, k1, tilt, mx, my]   CHX_T_SEXTUPOLE [length, k2, tilt, mx, my]
 *   CHX_T_DIPOLE = the CHX_DIPOLE vector [length, angle, k1, e1, e2, tilt, fint, fint_exit, gap]
 *   CHX_T_GENERAL [length, k1, k2, hx]: the bare tensor of track_methods.base_ttensor (track_methods.py:80-281) for any
 *   combination of the three strengths, without a first-order block and without dressing (custom elements)
 * T_out[B][7][7][7] (dtype). */
enum chx_t_kind { CHX_T_DRIFT = 0, CHX_T_QUADRUPOLE = 1, CHX_T_DIPOLE = 2, CHX_T_SEXTUPOLE = 3, CHX_T_GENERAL = 4 };
int chx_t_num_params(int kind);
int chx_build_ttensor(int kind, const void* params, const void* energy, double mass_eV, int64_t B, int64_t Bp,
                      int64_t Be, int dtype, void* T_out, void* stream);
int chx_apply_second_order(const void* x_in, const void* T, void* x_out, int64_t B, int64_t Bx, int64_t BT,
                           int64_t N, int dtype, void* stream);
/* A run of E elements tracked with their second-order maps on one beam x_in[N][7] (a lattice with tracking_method =
 * "second_order": element.py:195-228 once per element): HOST arrays T_maps[E] (device pointers to each element's [7][7][7] map)
 * and lengths[E] (device pointers to the elements' length scalars, only read when s_in / s_out are given: s_out = (((s_in +
 * l_0) + l_1) + ...) in dtype). x_tmp[N][7] is scratch (may be NULL for E = 1). Two launches per 224 elements, the particles in
 * registers across the run (x_tmp holds the folded coefficients, 256 per element); a beam too small for that scratch: E
 * launches. Either way one call and the values of E separate chx_apply_second_order calls (an exact zero may differ in sign). */
int chx_second_order_chain(const void* const* T_maps, const void* const* lengths, int64_t E, const void* x_in, int64_t N, int dtype,
                           void* x_out, void* x_tmp, const void* s_in, void* s_out, void* stream);
/* The same run with first-order maps in between: linear[E] (HOST array, may be NULL) marks T_maps[e] as a [7][7] map — a merged
 * run of linear elements between second-order ones (segment.py:545-574 tracks it with its composed map) — applied with the
 * arithmetic of chx_apply_affine7; lengths[e] is then the run's summed length. One pass over the beam for the whole stretch. */
int chx_second_order_chain_mixed(const void* const* T_maps, const int32_t* linear, const void* const* lengths, int64_t E,
                                 const void* x_in, int64_t N, int dtype, void* x_out, void* x_tmp, const void* s_in, void* s_out,
                                 void* stream);
/* Derivatives of the two calls above (reference: torch autograd through track_methods.py:80-296 and the einsum).
 * chx_build_ttensor_vjp: dparams[B][P], denergy[B] (dtype) = dT[B][343] . dT/dtheta (dual numbers, one workgroup
 * per (row, input)); rows are NOT reduced when params / energy are broadcast (Bp or Be = 1) — the caller sums.
 * chx_apply_second_order_bwd: dx[B][N][7] (dtype, may be NULL) and dU_partials (may be NULL) =
 * chx_second_order_bwd_partials_count(B) doubles [B][256][196]: partial sums of dY_i x_j x_k over particles for
 * j <= k in the order (i, (0,0),(0,1)..(0,6),(1,1)..(6,6)); dT_ijk = dT_ikj = sum over axis 1. */
int chx_build_ttensor_vjp(int kind, const void* params, const void* energy, double mass_eV, const void* dT, int64_t B,
                          int64_t Bp, int64_t Be, int dtype, void* dparams, void* denergy, void* stream);
int64_t chx_second_order_bwd_partials_count(int64_t B);
int chx_apply_second_order_bwd(const void* x_in, const void* T, const void* dY, void* dx, double* dU_partials,
                               int64_t B, int64_t Bx, int64_t BT, int64_t N, int dtype, void* stream);

/* ---- Aperture (aperture.py:90-135): survival_out[B][N] = survival_in * inside, with
 * limits[Bl][2] = (x_max, y_max); rectangular uses strict inequalities, elliptical x^2/x_max^2 + y^2/y_max^2 <= 1,
 * both evaluated in `dtype`. survival_in may be NULL (all ones). */
enum chx_aperture_shape { CHX_APERTURE_RECTANGULAR = 0, CHX_APERTURE_ELLIPTICAL = 1 };
int chx_aperture_mask(const void* x_in, const void* survival_in, const void* limits, int shape, int64_t B, int64_t Bx,
                      int64_t Bs, int64_t Bl, int64_t N, int dtype, void* survival_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CHX_H */

// chx_sc_kick.hip — one C call for a whole SpaceChargeKick.track (space_charge_kick.py:477-586) on a power-of-two grid:
// moments -> grid geometry -> [side stream: Green-function table + spectrum] -> sorted / direct deposit -> pruned FFT
// convolution -> field gradient -> gather + kick. Every stage is one of the public entry points of chx.h; this file only
// carves the caller's workspace and orders the launches, so that the host pays ONE foreign-function call per kick instead
// of ~20 (at the reference's default 32^3 grid a kick is ~0.12 ms of GPU work behind ~0.24 ms of Python otherwise).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>
#include <cstring>

#include "chx.h"
#include "chx_sc_tiles.h"


// Flags of the events that order the kick's two streams. Both streams belong to one device, the events are created, recorded, waited for
// and destroyed inside one call and never inspected by the host, and what crosses between the streams is device memory written by
// kernels that have ENDED (their own end-of-kernel release made it visible to the device; the waiting stream's next kernel acquires
// at its start): the system-scope fence an event record carries by default — a cache write-back and invalidation in front of the
// kernel that follows it — buys nothing here and cost 2.4 % of a C4 track (1.86-1.88 -> 1.82-1.84 ms, benchmarks/_c4_event_ab.sh).
// CHX_TUNE_EVENT_FLAGS: 0 = the default system-scope fence, 1 = hipEventReleaseToDevice, 2 (default) = hipEventDisableSystemFence.
static unsigned sc_event_flags() {
    static const unsigned flags = [] {
        const char* e = getenv("CHX_TUNE_EVENT_FLAGS");
        const int v = e ? atoi(e) : 2;
        return (unsigned)hipEventDisableTiming | (v == 1 ? (unsigned)hipEventReleaseToDevice : v == 2 ? (unsigned)hipEventDisableSystemFence : 0u);
    }();
    return flags;
}
#include "chx_sc_geom_dev.h"

namespace {

constexpr double kEpsilon0 = 8.8541878188e-12;  // scipy.constants.epsilon_0 (CODATA 2022), space_charge_kick.py:14
constexpr int64_t kSortedMinParticles = 65536;  // below this the direct deposit is cheaper than the five sorted launches

size_t align256(size_t n) { return (n + 255) & ~(size_t)255; }

struct Layout {
    size_t mom_ws, mom, geo, pot, rho, dep_ws, table, green_ws, ghat, conv_ws, phi, total;  // table: end of dep_ws
};

chx_cic_args deposit_args(int64_t B, int64_t Bx, int64_t Bq, int64_t Bs, int64_t N, const int32_t* bins, int dtype) {
    chx_cic_args a;
    std::memset(&a, 0, sizeof(a));
    a.ndim = 3;
    a.cols[0] = 0; a.cols[1] = 2; a.cols[2] = 4;
    for (int d = 0; d < 3; ++d) a.bins[d] = bins[d];
    a.grid_strides[0] = (int64_t)bins[1] * bins[2];
    a.grid_strides[1] = bins[2];
    a.grid_strides[2] = 1;
    a.grid_batch_stride = (int64_t)bins[0] * bins[1] * bins[2];
    a.B = B; a.Bx = Bx; a.Bq = Bq; a.Bs = Bs; a.Be = B; a.Bsc = B; a.Bsh = 1; a.N = N;
    a.dtype = dtype;
    a.abs_charge = 0;
    return a;
}

Layout layout(int64_t B, int64_t N, const int32_t* bins, int dtype) {
    const size_t esz = dtype == CHX_F32 ? 4 : 8;
    const size_t ncell = (size_t)bins[0] * bins[1] * bins[2];
    const size_t npts = (size_t)(bins[0] + 1) * (bins[1] + 1) * (bins[2] + 1);
    Layout L;
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t at = off; off += align256(bytes); return at; };
    L.mom_ws = take(chx_sc_beam_geometry_workspace_bytes(B, N));
    L.mom = take(0);
    L.geo = take((size_t)B * 17 * esz);          // half 3, cell 3, gamma 1, dt 1, scale 3, extent 6
    L.pot = take((size_t)B * sizeof(double));
    L.rho = take((size_t)B * ncell * esz);
    chx_cic_args a = deposit_args(B, B, B, B, N, bins, dtype);
    // the workspace query only reads shapes; it needs non-null x / extent pointers to pass validation
    a.x = &a; a.extent = &a;
    L.dep_ws = take(N >= kSortedMinParticles ? chx_cic_sorted_workspace_bytes(&a) : 0);
    L.table = take(0);
    L.green_ws = take(chx_sc_green_fast_workspace_bytes(B, bins, dtype));   // corner table + compact Green function
    L.ghat = take((size_t)B * npts * esz);
    L.conv_ws = take(chx_sc_convolve_workspace_bytes(B, bins, dtype));
    L.phi = take(chx_sc_phi_halo_elements(B, bins) * esz);
    L.total = off;
    return L;
}

}  // namespace

extern "C" size_t chx_sc_kick_workspace_bytes(int64_t B, int64_t N, const int32_t* bins, int dtype) {
    if (B < 1 || N < 1 || !bins || !chx_sc_pruned_supported(bins, dtype)) return 0;
    return layout(B, N, bins, dtype).total;
}

extern "C" int chx_sc_kick(const void* x_in, const void* charge, const void* survival, const void* energy,
                           const void* length, const void* grid_extent, double mass_eV, int64_t B, int64_t Bx, int64_t Bq,
                           int64_t Bs, int64_t Bext, int64_t N, const int32_t* bins, int dtype, void* x_out,
                           void* workspace, size_t workspace_bytes, void* stream, void* side_stream, const void* post_map,
                           int64_t BR) {
    if (!x_in || !charge || !survival || !energy || !length || !grid_extent || !x_out || !workspace)
        return CHX_ERR_INVALID_ARG;
    if (B < 1 || N < 1 || !bins || !chx_sc_pruned_supported(bins, dtype)) return CHX_ERR_INVALID_ARG;
    const Layout L = layout(B, N, bins, dtype);
    if (workspace_bytes < L.total) return CHX_ERR_WORKSPACE;
    const size_t esz = dtype == CHX_F32 ? 4 : 8;
    char* ws = (char*)workspace;
    hipStream_t main = (hipStream_t)stream;
    hipStream_t side = side_stream ? (hipStream_t)side_stream : main;

    char* geo = ws + L.geo;
    void* half = geo;
    void* cell = geo + (size_t)B * 3 * esz;
    void* gamma = geo + (size_t)B * 6 * esz;
    void* dt = geo + (size_t)B * 7 * esz;
    void* scale = geo + (size_t)B * 8 * esz;
    void* extent = geo + (size_t)B * 11 * esz;
    double* pot_scale = (double*)(ws + L.pot);
    void* rho = ws + L.rho;
    void* ghat = ws + L.ghat;
    void* phi = ws + L.phi;

    // beam sizes -> grid geometry (space_charge_kick.py:531-550); the unnormalised inverse FFT's 1 / (8 g^3) goes into
    // the potential factor
    const double n_padded = 8.0 * bins[0] * bins[1] * bins[2];
    const double pot_factor = 1.0 / (4.0 * M_PI * kEpsilon0) / n_padded;
    int st = chx_sc_beam_geometry(x_in, survival, grid_extent, energy, length, mass_eV, pot_factor, B, Bx, Bs, Bext, B, B, N,
                                  bins, dtype, half, cell, gamma, dt, scale, extent, pot_scale, ws + L.mom_ws,
                                  L.mom - L.mom_ws, main);
    if (st != CHX_OK) return st;

    // Green-function chain on the side stream while the main stream deposits the charge
    hipEvent_t fork = nullptr, join = nullptr;
    const bool forked = side != main;
    if (forked) {
        if (hipEventCreateWithFlags(&fork, sc_event_flags()) != hipSuccess ||
            hipEventCreateWithFlags(&join, sc_event_flags()) != hipSuccess)
            return CHX_ERR_LAUNCH;
        (void)hipEventRecord(fork, main);
        (void)hipStreamWaitEvent(side, fork, 0);
    }
    st = chx_sc_green_spectrum_fast(cell, gamma, B, bins, dtype, ghat, ws + L.green_ws, L.ghat - L.green_ws, side);
    if (forked) (void)hipEventRecord(join, side);

    // the sorted deposit stores every cell of rho itself; the direct one adds into a zeroed grid
    const bool sorted = N >= kSortedMinParticles;
    if (st == CHX_OK && !sorted &&
        hipMemsetAsync(rho, 0, (size_t)B * bins[0] * bins[1] * bins[2] * esz, main) != hipSuccess)
        st = CHX_ERR_LAUNCH;
    if (st == CHX_OK) {
        chx_cic_args a = deposit_args(B, Bx, Bq, Bs, N, bins, dtype);
        a.x = x_in; a.charge = charge; a.survival = survival; a.extent = extent; a.scale = scale; a.shift = nullptr;
        a.grid = rho;
        st = sorted ? chx_cic_deposit_sorted_overwrite(&a, ws + L.dep_ws, L.table - L.dep_ws, main)
                    : chx_cic_deposit(&a, main);
    }
    // potential inside a halo; the field (central differences, space_charge_kick.py:324-385) is formed per particle in the
    // gather: the 33.5 MB force grid of the four-kernel form is neither written nor read. The main stream joins the side
    // stream where the Green spectrum is first read (in front of the z pass), not in front of the whole convolution.
    if (st == CHX_OK)
        st = chx_sc_convolve_halo_after(rho, ghat, pot_scale, B, bins, dtype, phi, ws + L.conv_ws, L.phi - L.conv_ws, main,
                                        forked ? (void*)join : nullptr);
    else if (forked)
        (void)hipStreamWaitEvent(main, join, 0);     // an error path still rejoins the side stream
    if (forked) {
        (void)hipEventDestroy(fork);
        (void)hipEventDestroy(join);
    }
    if (st != CHX_OK) return st;
    return chx_sc_gather_kick_phi(x_in, phi, half, cell, gamma, energy, dt, mass_eV, B, Bx, B, N, bins, dtype, post_map, BR,
                                  x_out, main);
}

// ---- a kick of a CHAIN of kicks on the tile-ordered beam (chx_sc_tiles.h) -------------------------------------------------
// flags bit 0 (CHX_SC_FIRST): x_in is in the caller's particle order, charge / survival are the caller's arrays; the kick sorts
//   the rows by deposit tile into `state` and leaves x_out in tile order. Without it x_in must be the x_out (possibly tracked
//   through linear maps) of the previous kick of the same chain with the same `state`.
// flags bit 1 (CHX_SC_LAST): x_out is written in the caller's particle order (the stored permutation is undone).
extern "C" size_t chx_sc_kick_sorted_workspace_bytes(int64_t N, const int32_t* bins, int dtype) {
    return chx_sc_kick_workspace_bytes(1, N, bins, dtype);
}

namespace {

// pointers into the workspace / state of one kick of a chain
struct SortedKick {
    Layout L;
    ScTileLayout T;
    char *ws, *st;
    void *half, *cell, *gamma, *dt, *scale, *extent, *rho, *ghat, *phi;
    double* pot_scale;
};

// does this kick of a chain form its geometry and run its bookkeeping inside the kernels that need them (chx_sc_geom_dev.h)?
// CHX_SC_RIDERS=0 (benchmarks / bisecting) keeps the two one-workgroup launches.
bool sorted_kick_rides(int flags, int dtype, const int32_t* bins) {
    static const bool enabled = [] { const char* e = getenv("CHX_SC_RIDERS"); return !(e && e[0] == '0'); }();
    return enabled && !(flags & 1) && (flags >> 8) > 0 && dtype == CHX_F32 && chx_sc_convolve_carries_schedule(bins);
}

// The bookkeeping step behind the deposit alone (sc_tile_schedule_block as one extra workgroup of the convolution's first pass) does
// not depend on how the kick's geometry was formed: a kick that is not the first of its chain rides it also when its geometry came
// from exchanged moments (a particle-sharded beam) or from the partial sums — one launch less on the main stream.
bool sorted_kick_rides_schedule(int flags, int dtype, const int32_t* bins) {
    static const bool enabled = [] { const char* e = getenv("CHX_SC_RIDERS"); return !(e && e[0] == '0'); }();
    return enabled && !(flags & 1) && dtype == CHX_F32 && chx_sc_convolve_carries_schedule(bins);
}

int sorted_kick_prepare(int64_t N, const int32_t* bins, int dtype, void* workspace, size_t workspace_bytes, void* state,
                        size_t state_bytes, SortedKick& k) {
    if (!workspace || !state) return CHX_ERR_INVALID_ARG;
    if (N < 1 || !bins || !chx_sc_pruned_supported(bins, dtype)) return CHX_ERR_INVALID_ARG;
    const size_t need_state = chx_sc_tile_state_bytes(N, bins, dtype);
    if (need_state == 0) return CHX_ERR_INVALID_ARG;
    if (state_bytes < need_state) return CHX_ERR_WORKSPACE;
    k.L = layout(1, N, bins, dtype);
    if (workspace_bytes < k.L.total) return CHX_ERR_WORKSPACE;
    k.T = sc_tile_layout(N, bins, dtype);
    const size_t esz = dtype == CHX_F32 ? 4 : 8;
    k.ws = (char*)workspace;
    k.st = (char*)state;
    // the kick's geometry lives in the chain's state (copy 0: the main stream's), not in the workspace: with riders
    // (chx_sc_geom_dev.h) it is written by the first kernels of the kick and read until its last one
    char* geo = k.st + k.T.geo[0];
    k.half = geo;
    k.cell = geo + (size_t)3 * esz;
    k.gamma = geo + (size_t)6 * esz;
    k.dt = geo + (size_t)7 * esz;
    k.scale = geo + (size_t)8 * esz;
    k.extent = geo + (size_t)11 * esz;
    k.pot_scale = (double*)(geo + kScGeoPotOffset);
    k.rho = k.ws + k.L.rho;
    k.ghat = k.ws + k.L.ghat;
    k.phi = k.ws + k.L.phi;
    return CHX_OK;
}

}  // namespace

// The kick in two halves, so that a beam whose particles are spread over several GPUs (cheetah_amd.sharding) can put its two
// exchanges in between WITHOUT leaving the tile-ordered chain:
//   begin : grid geometry (from `beam_moments` when given — the 29 doubles of the WHOLE beam (moment_rows = 0), or the
//           moment_rows x 29 doubles of its shards as all-gathered, merged here — else from this process's own particles) -> [side stream: Green spectrum] -> [first kick: tile sort] -> tile deposit into *rho_out (compact
//           [gx][gy][gz] array of `dtype` inside the workspace: the caller may sum it over the ranks in place)
//   finish: pruned FFT convolution (the main stream joins the side stream in front of the pass that reads the spectrum) ->
//           gather + kick (+ post_map) on the ordered rows.
// The same workspace and state must be handed to both halves; nothing else may use the workspace in between.
extern "C" int chx_sc_kick_sorted_begin(const void* x_in, const void* charge, const void* survival, const void* energy,
                                        const void* length, const void* grid_extent, double mass_eV, int64_t N, const int32_t* bins,
                                        int dtype, void* workspace, size_t workspace_bytes, void* state, size_t state_bytes, int flags,
                                        const double* beam_moments, int32_t moment_rows, void** rho_out, void* stream,
                                        void* side_stream) {
    if (!x_in || !energy || !length || !grid_extent) return CHX_ERR_INVALID_ARG;
    const bool first = flags & 1, last = flags & 2;
    if (first && (!charge || !survival)) return CHX_ERR_INVALID_ARG;
    SortedKick k;
    int rc = sorted_kick_prepare(N, bins, dtype, workspace, workspace_bytes, state, state_bytes, k);
    if (rc != CHX_OK) return rc;
    hipStream_t main = (hipStream_t)stream;
    hipStream_t side = side_stream ? (hipStream_t)side_stream : main;

    const double n_padded = 8.0 * bins[0] * bins[1] * bins[2];
    const double pot_factor = 1.0 / (4.0 * M_PI * kEpsilon0) / n_padded;
    // beam sizes -> grid geometry. From the whole beam's moments when the caller has them (particle-sharded beam). Else, first
    // kick: the two launches of chx_sc_beam_geometry on the caller's arrays; later kicks: the gather pass of the previous kick left
    // the partial sums of the rows it wrote (= this kick's x_in) in the state: one launch.
    // Riders (flags bits 8..: the kick's index in its chain, > 0; fp32; lines the first FFT pass can carry the bookkeeping on): no
    // geometry launch at all — the workgroups of the deposit (main stream) and of the Green function's corner table (side stream)
    // form it from the sums the previous gather pass added up, each launch for its own stream (chx_sc_geom_dev.h), and the
    // bookkeeping step behind the deposit rides in the convolution's first pass: 12 launches per kick + the run's map instead of
    // 15, none of them a one-workgroup kernel on the critical path.
    if (beam_moments && (flags >> 8)) return CHX_ERR_INVALID_ARG;    // (an indexed kick takes its geometry from the chain's own sums)
    const bool ride = sorted_kick_rides(flags, dtype, bins);
    ScGeoSums riders[2];
    if (ride) {
        for (int c = 0; c < 2; ++c) {
            ScGeoSums& r = riders[c];
            r.sums = (const double*)(k.st + k.T.sums[((flags >> 8) - 1) & 1]);     // what the previous kick's gather pass added up
            r.grid_extent = grid_extent;
            r.energy = energy;
            r.length = length;
            r.mass = mass_eV;
            r.pot_factor = pot_factor;
            r.gx = bins[0]; r.gy = bins[1]; r.gz = bins[2];
            r.geo_out = k.st + k.T.geo[c];
        }
    } else if (beam_moments)
        rc = chx_sc_geometry_tiles(beam_moments, grid_extent, energy, length, mass_eV, pot_factor, 1, 1, 1, 1, 1, bins, dtype, k.half,
                                   k.cell, k.gamma, k.dt, k.scale, k.extent, k.pot_scale, first ? nullptr : k.st + k.T.hdr, moment_rows, main);
    else if (first)
        rc = chx_sc_beam_geometry_tiles(x_in, survival, grid_extent, energy, length, mass_eV, pot_factor, 1, 1, 1, 1, 1, 1, N, bins,
                                        dtype, k.half, k.cell, k.gamma, k.dt, k.scale, k.extent, k.pot_scale, k.ws + k.L.mom_ws,
                                        k.L.mom - k.L.mom_ws, k.st + k.T.hdr, 1, main);
    else
        rc = chx_sc_geometry_from_partials((const double*)(k.st + k.T.sigma), k.T.sigma_blocks, grid_extent, energy, length, mass_eV,
                                           pot_factor, bins, dtype, k.half, k.cell, k.gamma, k.dt, k.scale, k.extent, k.pot_scale,
                                           k.st + k.T.hdr, main);
    if (rc != CHX_OK) return rc;

    const bool forked = side != main;
    if (forked) {
        hipEvent_t fork = nullptr;
        if (hipEventCreateWithFlags(&fork, sc_event_flags()) != hipSuccess) return CHX_ERR_LAUNCH;
        (void)hipEventRecord(fork, main);
        (void)hipStreamWaitEvent(side, fork, 0);
        (void)hipEventDestroy(fork);
    }
    rc = chx_sc_green_spectrum_chain(k.cell, k.gamma, bins, dtype, k.ghat, k.ws + k.L.green_ws, k.L.ghat - k.L.green_ws,
                                     ride ? &riders[forked ? 1 : 0] : nullptr, side);

    // first kick of the chain: order the rows by deposit tile (into the state's row buffer); every kick: deposit from the ordered
    // rows (the merge pass decides on the device whether this kick's gather re-orders them for the kicks that follow)
    const void* rows = first ? nullptr : x_in;             // nullptr = the state's row buffer
    if (rc == CHX_OK && first)
        rc = chx_sc_tile_sort(x_in, charge, survival, k.extent, k.scale, N, bins, dtype, state, state_bytes, main);
    void* acc = nullptr;
    if (rc == CHX_OK) {
        acc = k.st + k.T.cross;
        // (riders on ONE stream: the corner table's rider has published copy 0 in front of the deposit; the deposit reads it in place)
        rc = chx_sc_tile_deposit_chain(rows, k.extent, k.scale, N, bins, dtype, state, state_bytes, last ? 0 : 1,
                                       ride && forked ? &riders[0] : nullptr, !(ride || sorted_kick_rides_schedule(flags, dtype, bins)), main);
    }
    if (rc != CHX_OK && forked) {                           // an error path still rejoins the side stream
        hipEvent_t join = nullptr;
        if (hipEventCreateWithFlags(&join, sc_event_flags()) == hipSuccess) {
            (void)hipEventRecord(join, side);
            (void)hipStreamWaitEvent(main, join, 0);
            (void)hipEventDestroy(join);
        }
    }
    if (rho_out) *rho_out = acc;
    return rc;
}

extern "C" int chx_sc_kick_sorted_finish(const void* x_in, const void* energy, double mass_eV, int64_t N, const int32_t* bins,
                                         int dtype, void* x_out, void* workspace, size_t workspace_bytes, void* state,
                                         size_t state_bytes, int flags, void* stream, void* side_stream, const void* post_map) {
    if (!x_in || !energy || !x_out) return CHX_ERR_INVALID_ARG;
    const bool first = flags & 1, last = flags & 2;
    SortedKick k;
    int rc = sorted_kick_prepare(N, bins, dtype, workspace, workspace_bytes, state, state_bytes, k);
    if (rc != CHX_OK) return rc;
    hipStream_t main = (hipStream_t)stream;
    hipStream_t side = side_stream ? (hipStream_t)side_stream : main;
    hipEvent_t join = nullptr;
    const bool forked = side != main;
    if (forked) {
        if (hipEventCreateWithFlags(&join, sc_event_flags()) != hipSuccess) return CHX_ERR_LAUNCH;
        (void)hipEventRecord(join, side);                 // behind the Green spectrum `begin` put on the side stream
    }
    // the charge sits in the chain's accumulation grid (state); the convolution's first pass leaves it zeroed for the next kick
    ScScheduleArgs sched;
    const bool ride = sorted_kick_rides(flags, dtype, bins) || sorted_kick_rides_schedule(flags, dtype, bins);
    if (ride) rc = chx_sc_schedule_args(N, bins, dtype, state, state_bytes, last ? 0 : 1, &sched);
    if (rc == CHX_OK)
        rc = chx_sc_convolve_halo_chain(k.st + k.T.cross, k.ghat, k.pot_scale, bins, dtype, k.phi, k.ws + k.L.conv_ws,
                                        k.L.phi - k.L.conv_ws, main, forked ? (void*)join : nullptr, ride ? &sched : nullptr);
    if (forked) {
        if (rc != CHX_OK) (void)hipStreamWaitEvent(main, join, 0);
        (void)hipEventDestroy(join);
    }
    if (rc != CHX_OK) return rc;
    const void* rows = first ? nullptr : x_in;
    // (the gather adds the beam-size sums of the rows it writes into set index & 1 for the next kick's riders)
    return chx_sc_tile_gather_kick_chain(rows, k.phi, k.half, k.cell, k.gamma, energy, k.dt, mass_eV, N, bins, dtype, post_map, state,
                                         state_bytes, last ? 1 : 0, x_out, (flags >> 8) & 1, main);
}

extern "C" int chx_sc_kick_sorted(const void* x_in, const void* charge, const void* survival, const void* energy, const void* length,
                                  const void* grid_extent, double mass_eV, int64_t N, const int32_t* bins, int dtype, void* x_out,
                                  void* workspace, size_t workspace_bytes, void* state, size_t state_bytes, int flags, void* stream,
                                  void* side_stream, const void* post_map) {
    if (!x_out) return CHX_ERR_INVALID_ARG;
    int rc = chx_sc_kick_sorted_begin(x_in, charge, survival, energy, length, grid_extent, mass_eV, N, bins, dtype, workspace,
                                      workspace_bytes, state, state_bytes, flags, nullptr, 0, nullptr, stream, side_stream);
    if (rc != CHX_OK) return rc;
    return chx_sc_kick_sorted_finish(x_in, energy, mass_eV, N, bins, dtype, x_out, workspace, workspace_bytes, state, state_bytes,
                                     flags, stream, side_stream, post_map);
}

// what this process's rows of a running chain contribute to the beam moments the NEXT kick's grid is built from: the sums the
// last gather pass accumulated (chx_sc_tile_gather_kick), as a chx_moments row (x, y, tau entries; chx_sc_partials_moments)
extern "C" int chx_sc_tile_beam_moments(const void* state, size_t state_bytes, int64_t N, const int32_t* bins, int dtype,
                                        double* moments_out, void* stream) {
    if (!state || !moments_out || N < 1 || !bins) return CHX_ERR_INVALID_ARG;
    const size_t need_state = chx_sc_tile_state_bytes(N, bins, dtype);
    if (need_state == 0) return CHX_ERR_INVALID_ARG;
    if (state_bytes < need_state) return CHX_ERR_WORKSPACE;
    const ScTileLayout T = sc_tile_layout(N, bins, dtype);
    return chx_sc_partials_moments((const double*)((const char*)state + T.sigma), T.sigma_blocks, moments_out, stream);
}

// chx_sc_tiles.h — the tile-ordered beam of a chain of SpaceChargeKicks (chx_sc_kick_sorted; space_charge_kick.py:477-586).
//
// Inside one Segment.track the particles barely move against the space-charge grid between two kicks (the grid follows the
// beam: its extent is a multiple of the beam sigmas), so the counting sort by 8^3 deposit tile that a kick needs is done ONCE:
// the first kick of a chain writes the particle rows in tile order and every later kick works on them in place —
//   * deposit: one workgroup per tile accumulates its own slot range in an LDS block of (tile + 1)^3 cells (no records, no
//     duplication of particles over neighbouring tiles) and adds the non-zero cells of the block to the chain's accumulation
//     grid with one float atomic each; the first FFT pass of the convolution reads that grid and writes the zeros back;
//   * gather: the per-particle kernel of the untiled path on the ordered rows — the lanes of a wave sit in one tile, so the 12
//     line requests per particle for the 32 potential values around its cell hit the CU's vector cache (63 -> 33 us at 1e6
//     particles on 128^3); it also accumulates the beam sizes the next kick's grid needs over the rows it writes;
//   * a particle that has left the tile of its slot ("misfiled": 1 % of the beam per kick in a smooth channel, 10-20 % where
//     the beam goes through a focus) is still handled exactly, by the same code path: a corner that still falls into the
//     tile's block goes there, any other takes a global float atomic into the same accumulation grid; the gather reads the
//     potential from global memory anyway. When more than 1/16 of the beam
//     is misfiled the gather pass of that very kick writes its rows in the new tile order (the deposit pass counted the new
//     tile populations on the way): a device-side decision (a one-workgroup kernel behind the deposit), no host
//     synchronisation and no extra launch. The header keeps the running share of misfiled particles for the host (see
//     ScTileHeader): a beam that reshuffles between kicks is better served by the kick-by-kick path.
// The permutation back to the caller's particle order is carried along and applied by the last kick of the chain.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include "chx.h"

// device-resident control block at the start of the state buffer
struct ScTileHeader {
    int parity;        // which copy of tile_start / perm / ws / cs is current
    int scatter_now;   // this kick's gather writes its rows in a NEW tile order (decided by the merge pass of the same kick);
                       // the geometry kernel of the next kick then flips `parity` and clears the flag
    int ncross;        // misfiled particles found by the last deposit pass
    int misfiled_permille;   // sum over the chain's deposit passes of 1000 * misfiled / N ...
    int last_ncross;   // (= ncross)
    int n_sorts;       // how many times the chain (re)ordered its rows so far
    int n_deposits;    // ... and the number of those passes: a host that reads the header after a track (asynchronously) learns
                       // whether the beam keeps its tile order between kicks; where it does not (mean share above ~1/4) the
                       // kick-by-kick path with the generic sorted deposit is the faster one (cheetah_amd Segment does that)
    int pad;
};
static_assert(sizeof(ScTileHeader) == 32, "the host reads the header as eight ints");

constexpr int kScSortWG = 256;       // workgroups of the count / scatter passes
constexpr int kScSortThreads = 1024;
constexpr int kScTileCap = 8192;     // particles of one tile deposited through LDS by its workgroup; the rest take the slow way
constexpr int kScGatherRows = 256;   // rows per workgroup of the gather pass (four waves of 64 rows, each on its own)
constexpr int kScMisSlots = 64;     // (one counter would serialise a thousand atomics on one address: +4 us on the deposit)
constexpr int kScTdim = 8;           // the tile edge the kernels are written for (grids whose tile rule gives larger tiles — more
                                     // than 8192 tiles of 8^3 — keep the untiled path)

struct ScTileGeom {
    int tdim[3];    // tile edge in cells (a power of two)
    int tshift[3];  // log2(tdim)
    int ntile[3];
    int nt;
};

// same rule as the generic sorted deposit (chx_cic.hip tile_geom, 3-D): 8^3 tiles, edges doubled until nt <= 8192
static __host__ __device__ inline ScTileGeom sc_tile_geom(const int* bins) {
    ScTileGeom g;
    for (int d = 0; d < 3; ++d) g.tdim[d] = 8;
    for (;;) {
        g.nt = 1;
        int widest = 0;
        for (int d = 0; d < 3; ++d) {
            g.ntile[d] = (bins[d] + g.tdim[d] - 1) / g.tdim[d];
            g.nt *= g.ntile[d];
            if (g.ntile[d] > g.ntile[widest]) widest = d;
        }
        if (g.nt <= 8192 || g.tdim[widest] >= 64) break;
        g.tdim[widest] *= 2;
    }
    for (int d = 0; d < 3; ++d) {
        g.tshift[d] = 0;
        while ((1 << g.tshift[d]) < g.tdim[d]) ++g.tshift[d];
    }
    return g;
}

// byte offsets of the pieces of the state buffer (B = 1); [2] = one copy per parity
struct ScTileLayout {
    size_t hdr, newcount, mis, sums[2], cross, cursor, geo[2], tile_start[2], counts, totals, perm[2], ws[2], cs[2], home, rows_tmp, sigma, total;
    int64_t sigma_blocks;   // workgroups of the gather pass = partial sums per moment handed to the next kick's geometry kernel
    size_t zero_bytes;   // header + newcount + the crossers' grid: cleared by the first kick of a chain
};

static inline ScTileLayout sc_tile_layout(int64_t N, const int32_t* bins, int dtype) {
    const ScTileGeom g = sc_tile_geom(bins);
    const size_t esz = dtype == CHX_F32 ? 4 : 8;
    ScTileLayout L;
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t at = off; off += (bytes + 255) & ~(size_t)255; return at; };
    L.hdr = take(sizeof(ScTileHeader));
    L.newcount = take((size_t)g.nt * sizeof(int));
    L.mis = take((size_t)kScMisSlots * sizeof(int));              // misfiled particles of the last deposit, spread over a few counters
    // beam-size sums of the rows a gather pass wrote, [8][256] doubles (chx_sc_geom_dev.h): the pass of kick i adds into set i & 1
    // and clears the other one
    L.sums[0] = take((size_t)8 * 256 * sizeof(double));
    L.sums[1] = take((size_t)8 * 256 * sizeof(double));
    L.cross = take((size_t)bins[0] * bins[1] * bins[2] * esz);   // the accumulation grid of the deposits; all zero between two kicks
    L.zero_bytes = off;
    L.cursor = take((size_t)g.nt * sizeof(int));
    // the kick's grid geometry (half 3 | cell 3 | gamma | dt | scale 3 | extent 6 of the beam dtype, the potential factor as a
    // double at byte 192): copy 0 for the kernels of the main stream, copy 1 for the Green-function chain of the side stream
    L.geo[0] = take(256);
    L.geo[1] = take(256);
    // the two copies of a double-buffered array are contiguous: copy p starts p * (element count) elements behind copy 0
    L.tile_start[0] = take((size_t)2 * (g.nt + 1) * sizeof(int));
    L.tile_start[1] = L.tile_start[0] + (size_t)(g.nt + 1) * sizeof(int);
    L.counts = take((size_t)kScSortWG * g.nt * sizeof(int));
    L.totals = take((size_t)g.nt * sizeof(int));   // sort: tile totals; afterwards: misfiled particles per tile (deposit -> merge)
    L.perm[0] = take((size_t)2 * N * sizeof(int));
    L.perm[1] = L.perm[0] + (size_t)N * sizeof(int);
    L.ws[0] = take((size_t)2 * N * esz);
    L.ws[1] = L.ws[0] + (size_t)N * esz;
    L.cs[0] = take((size_t)2 * N * esz);
    L.cs[1] = L.cs[0] + (size_t)N * esz;
    L.home = take((size_t)N * sizeof(uint16_t));
    L.rows_tmp = take((size_t)N * 7 * esz);
    L.sigma_blocks = (N + kScGatherRows - 1) / kScGatherRows;
    L.sigma = take((size_t)8 * L.sigma_blocks * sizeof(double));
    L.total = off;
    return L;
}

// what the bookkeeping step behind a deposit pass works on (sc_tile_schedule_block)
struct ScScheduleArgs {
    ScTileGeom g;
    int64_t N;
    ScTileHeader* hdr;
    const int* newcount;
    const int* mis;
    int* cursor;
    int* tile_start2;
    int allow_reorder;
};

#ifdef __HIPCC__
// The bookkeeping behind a deposit pass, ONE workgroup of 256 threads (a kernel of its own, or one extra workgroup of a launch
// that follows the deposit on its stream — the first FFT pass of the convolution): rolls the header over from the previous kick
// (its gather wrote the rows in a new tile order: those arrays are in force now), decides whether THIS kick's gather re-orders
// the rows (more than 1/16 of the beam misfiled, and a later kick to profit from it) and, if so, turns the new tile populations
// into the slot cursors and the next tile starts. The counters (newcount[], mis[]) are put back to zero by the gather pass of the
// same kick, which runs behind this step and in front of the next deposit. (A ticket at the end of the deposit kernel instead needs
// a device-scope release per workgroup: 4096 L2 write-backs took that kernel from 30 to 500 us.)
__device__ __forceinline__ void sc_tile_schedule_block(const ScScheduleArgs& a, int* part /* LDS: 257 ints */) {
    const int par0 = a.hdr->parity, scat0 = a.hdr->scatter_now;      // (read by every thread before thread 0 rewrites them)
    const int par = par0 ^ (scat0 & 1);                              // the copy of the arrays in force for this kick
    if (threadIdx.x < 64) {                          // the deposit kernel's counters (zero before a chain and after every gather)
        int v = threadIdx.x < kScMisSlots ? a.mis[threadIdx.x] : 0;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        if (threadIdx.x == 0) part[256] = v;
    }
    __syncthreads();
    const int n = part[256];
    const bool reorder = a.allow_reorder && (int64_t)n * 16 > a.N;
    if (threadIdx.x == 0) {
        a.hdr->parity = par;
        a.hdr->scatter_now = reorder ? 1 : 0;
        a.hdr->ncross = n;
        a.hdr->last_ncross = n;
        a.hdr->misfiled_permille += (int)((int64_t)n * 1000 / a.N);   // over the chain so far: what the host's guard reads
        a.hdr->n_deposits += 1;
        if (reorder) a.hdr->n_sorts += 1;
    }
    if (!reorder) return;                            // the usual case: this step is one round trip long
    // new tile populations -> slot cursors and the next tile starts (exclusive scan over the tiles)
    const int per = (a.g.nt + 255) / 256;
    const int lo = threadIdx.x * per, hi = (lo + per < a.g.nt) ? lo + per : a.g.nt;
    int sum = 0;
    for (int k = lo; k < hi; ++k) sum += a.newcount[k];
    int incl = sum;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int v = __shfl_up(incl, off, 64);
        if (lane >= off) incl += v;
    }
    if (lane == 63) part[wv] = incl;
    __syncthreads();
    int before = 0;
    for (int k = 0; k < wv; ++k) before += part[k];
    const int total = part[0] + part[1] + part[2] + part[3];
    int run = before + incl - sum;
    int* __restrict__ ts_next = a.tile_start2 + (int64_t)(par ^ 1) * (a.g.nt + 1);
    for (int k = lo; k < hi; ++k) {
        a.cursor[k] = run;
        ts_next[k] = run;
        run += a.newcount[k];
    }
    if (threadIdx.x == 255) ts_next[a.g.nt] = total;
}
#endif

// ---- the chain's forms of three public entry points (chx_sc_kick.hip orders them; not part of include/chx.h) -------------------
struct ScGeoSums;
// chx_sc_tile_deposit_acc; `rider`: every workgroup forms the kick's geometry from the gather's sums (extent / scale are then
// ignored); `schedule` false: the caller runs the bookkeeping step elsewhere (chx_sc_convolve_halo_chain)
int chx_sc_tile_deposit_chain(const void* rows, const void* extent, const void* scale, int64_t N, const int32_t* bins, int dtype,
                              void* state, size_t state_bytes, int allow_reorder, const ScGeoSums* rider, bool schedule, void* stream);
// chx_sc_green_spectrum_fast (B = 1); `rider`: the corner-table launch forms cell / gamma itself (those two are then ignored)
int chx_sc_green_spectrum_chain(const void* cell, const void* gamma, const int32_t* bins, int dtype, void* Ghat, void* workspace,
                                size_t workspace_bytes, const ScGeoSums* rider, void* stream);
// chx_sc_convolve_halo_consume (B = 1); `schedule`: an extra workgroup of the first pass runs sc_tile_schedule_block
int chx_sc_convolve_halo_chain(void* rho, const void* Ghat, const double* scale, const int32_t* bins, int dtype, void* phi_halo,
                               void* workspace, size_t workspace_bytes, void* stream, void* ghat_ready_event,
                               const ScScheduleArgs* schedule);
// the bookkeeping step's arguments for a chain's state buffer
int chx_sc_schedule_args(int64_t N, const int32_t* bins, int dtype, void* state, size_t state_bytes, int allow_reorder, ScScheduleArgs* out);
// chx_sc_tile_gather_kick; sums_set 0 / 1: the pass also adds the beam-size sums of the rows it writes into that set of the state and
// clears the other one (-1: neither)
int chx_sc_tile_gather_kick_chain(const void* rows, const void* phi_halo, const void* half, const void* cell, const void* gamma,
                                  const void* energy, const void* dt, double mass_eV, int64_t N, const int32_t* bins, int dtype,
                                  const void* post_map, void* state, size_t state_bytes, int unpermute, void* x_out, int sums_set,
                                  void* stream);
// can the first pass of the convolution carry the bookkeeping step on this grid?
int chx_sc_convolve_carries_schedule(const int32_t* bins);

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
constexpr int kScChunk = 256;        // slots per workgroup of the gather pass (four waves of 64 rows, each on its own)
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
    size_t hdr, newcount, mis, cross, cursor, tile_start[2], counts, totals, perm[2], ws[2], cs[2], home, rows_tmp, sigma, total;
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
    L.cross = take((size_t)bins[0] * bins[1] * bins[2] * esz);   // the accumulation grid of the deposits; all zero between two kicks
    L.zero_bytes = off;
    L.cursor = take((size_t)g.nt * sizeof(int));
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
    L.sigma_blocks = (N + 255) / 256;
    L.sigma = take((size_t)8 * L.sigma_blocks * sizeof(double));
    L.total = off;
    return L;
}

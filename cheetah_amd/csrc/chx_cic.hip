// chx_cic.hip — cloud-in-cell deposition (1-3 D) and the Screen histogram.
//
// Replaces cheetah/utils/cloud_in_cell.py:8-451 (2^d scatter_add_ passes + ~30 N-sized
// temporaries) and the torch.histogramdd call of cheetah/accelerator/screen.py:292-311 with one
// pass per image: positions are picked straight out of the 7-vector rows (no stack / contiguous
// copies), the index arithmetic is done in the working dtype in the reference's exact operation
// order (this file is compiled with -ffp-contract=off, IEEE division) so that cell indices are
// bit-identical, and the 2^d weighted adds go out as hardware float atomics
// (global_atomic_add_f32 / _f64).
#include <cstdlib>

#include "chx_common.h"
#include "chx_cic_dev.h"

namespace {

template <typename T>
__global__ __launch_bounds__(CHX_BLOCK) void cic_deposit_kernel(CicDev a, const T* __restrict__ x,
                                                               const T* __restrict__ q,
                                                               const T* __restrict__ s,
                                                               const T* __restrict__ extent,
                                                               const T* __restrict__ scale,
                                                               const T* __restrict__ shift,
                                                               T* __restrict__ grid,
                                                               const T* __restrict__ Rmaps, int64_t BR) {
    __shared__ long long keys[kCombSlots];
    __shared__ double vals[kCombSlots];
    const int64_t b = blockIdx.y;
    CombTable<T> table;
    table.init(keys, vals, grid + b * a.gbatch);
    auto add = [&](int64_t off, T v) { table.add(off, v); };
    const T* Rb = Rmaps ? Rmaps + (BR == 1 ? 0 : b) * 49 : nullptr;
    for (int64_t n = (int64_t)blockIdx.x * CHX_BLOCK + threadIdx.x; n < a.N;
         n += (int64_t)gridDim.x * CHX_BLOCK) {
        const CicPoint<T> pt = cic_locate<T>(a, x, extent, scale, shift, b, n, Rb);
        if (!pt.inside) continue;  // masked_charges == 0 (cloud_in_cell.py:150-156)
        const T c = cic_charge<T>(a, q, s, b, n);
        T wf[3][2];
        int64_t off[3][2];
        bool ok[3][2];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
#pragma unroll
            for (int o = 0; o < 2; ++o) {
                if (d < a.ndim) {
                    const long long id = pt.i[d] + o;
                    ok[d][o] = (id >= 0) && (id < a.bins[d]);
                    const long long ic = id < 0 ? 0 : (id > a.bins[d] - 1 ? a.bins[d] - 1 : id);
                    off[d][o] = ic * a.gstride[d];
                    wf[d][o] = o ? pt.f[d] : ((T)1.0 - pt.f[d]);
                } else {
                    ok[d][o] = (o == 0);
                    off[d][o] = 0;
                    wf[d][o] = (T)1;
                }
            }
        }
        if (a.ndim == 1) {
#pragma unroll
            for (int ox = 0; ox < 2; ++ox)
                if (ok[0][ox]) add(off[0][ox], c * wf[0][ox]);
        } else if (a.ndim == 2) {
            // src = masked_charges * wx * wy  (cloud_in_cell.py:216-239), y outer / x inner order
#pragma unroll
            for (int oy = 0; oy < 2; ++oy)
#pragma unroll
                for (int ox = 0; ox < 2; ++ox)
                    if (ok[0][ox] && ok[1][oy])
                        add(off[0][ox] + off[1][oy], c * wf[0][ox] * wf[1][oy]);
        } else {
            // weight = wx * wy * wz ; src = masked_charges * weight (cloud_in_cell.py:368-382)
#pragma unroll
            for (int ox = 0; ox < 2; ++ox)
#pragma unroll
                for (int oy = 0; oy < 2; ++oy)
#pragma unroll
                    for (int oz = 0; oz < 2; ++oz)
                        if (ok[0][ox] && ok[1][oy] && ok[2][oz])
                            add(off[0][ox] + off[1][oy] + off[2][oz], c * (wf[0][ox] * wf[1][oy] * wf[2][oz]));
        }
    }
    table.flush();
}

template <typename T>
__global__ __launch_bounds__(CHX_BLOCK) void cic_indices_kernel(CicDev a, const T* __restrict__ x,
                                                               const T* __restrict__ extent,
                                                               const T* __restrict__ scale,
                                                               const T* __restrict__ shift,
                                                               int32_t* __restrict__ idx,
                                                               T* __restrict__ frac) {
    const int64_t b = blockIdx.y;
    for (int64_t n = (int64_t)blockIdx.x * CHX_BLOCK + threadIdx.x; n < a.N;
         n += (int64_t)gridDim.x * CHX_BLOCK) {
        const CicPoint<T> pt = cic_locate<T>(a, x, extent, scale, shift, b, n);
        for (int d = 0; d < a.ndim; ++d) {
            long long i = pt.i[d];
            i = i > 2147483647LL ? 2147483647LL : (i < -2147483647LL ? -2147483647LL : i);
            idx[(b * a.N + n) * a.ndim + d] = (int32_t)i;
            frac[(b * a.N + n) * a.ndim + d] = pt.f[d];
        }
    }
}

// backward: dweight = sum_c dgrid[c] * w_c ; dpos_d = c * sum_c dgrid[c] * dw_c/df_d / bw_d
template <typename T>
__global__ __launch_bounds__(CHX_BLOCK) void cic_bwd_kernel(CicDev a, const T* __restrict__ x,
                                                           const T* __restrict__ q,
                                                           const T* __restrict__ s,
                                                           const T* __restrict__ extent,
                                                           const T* __restrict__ scale,
                                                           const T* __restrict__ shift,
                                                           const T* __restrict__ dgrid,
                                                           T* __restrict__ dweight,
                                                           T* __restrict__ dpos) {
    const int64_t b = blockIdx.y;
    for (int64_t n = (int64_t)blockIdx.x * CHX_BLOCK + threadIdx.x; n < a.N;
         n += (int64_t)gridDim.x * CHX_BLOCK) {
        const CicPoint<T> pt = cic_locate<T>(a, x, extent, scale, shift, b, n);
        double dw = 0.0, dp[3] = {0.0, 0.0, 0.0};
        if (pt.inside) {
            const double c = (double)cic_charge<T>(a, q, s, b, n);
            const T* g = dgrid + b * a.gbatch;
            const int nc = 1 << a.ndim;
            for (int corner = 0; corner < nc; ++corner) {
                int o[3] = {corner & 1, (corner >> 1) & 1, (corner >> 2) & 1};
                bool valid = true;
                int64_t off = 0;
                double wf[3] = {1.0, 1.0, 1.0}, sg[3] = {0.0, 0.0, 0.0};
                for (int d = 0; d < a.ndim; ++d) {
                    const long long id = pt.i[d] + o[d];
                    valid = valid && (id >= 0) && (id < a.bins[d]);
                    const long long ic = id < 0 ? 0 : (id > a.bins[d] - 1 ? a.bins[d] - 1 : id);
                    off += ic * a.gstride[d];
                    wf[d] = o[d] ? (double)pt.f[d] : 1.0 - (double)pt.f[d];
                    sg[d] = o[d] ? 1.0 : -1.0;
                }
                if (!valid) continue;
                const double gv = (double)g[off];
                dw += gv * wf[0] * wf[1] * wf[2];
                for (int d = 0; d < a.ndim; ++d) {
                    double prod = sg[d];
                    for (int e = 0; e < a.ndim; ++e)
                        if (e != d) prod *= wf[e];
                    dp[d] += c * gv * prod / (double)pt.bw[d];
                }
            }
        }
        if (dweight) dweight[b * a.N + n] = (T)dw;
        if (dpos)
            for (int d = 0; d < a.ndim; ++d) dpos[(b * a.N + n) * a.ndim + d] = (T)dp[d];
    }
}

// ---- Screen histogram (screen.py:305-311): hist_bin (chx_cic_dev.h) is ATen histogramdd's bin search on explicit edges ----
struct HistDev {
    int64_t B, Bx, Bq, Bs, Bsh, N;
    int nx, ny;
};

template <typename T, bool INDICES>
__global__ __launch_bounds__(CHX_BLOCK) void hist2d_kernel(HistDev a, const T* __restrict__ x,
                                                          const T* __restrict__ q,
                                                          const T* __restrict__ s,
                                                          const T* __restrict__ shift,
                                                          const T* __restrict__ ex,
                                                          const T* __restrict__ ey,
                                                          T* __restrict__ image,
                                                          int32_t* __restrict__ ij) {
    __shared__ long long keys[INDICES ? 1 : kCombSlots];
    __shared__ double vals[INDICES ? 1 : kCombSlots];
    const int64_t b = blockIdx.y;
    CombTable<T> table;
    if (!INDICES) table.init(keys, vals, image + b * (int64_t)a.ny * a.nx);
    for (int64_t n = (int64_t)blockIdx.x * CHX_BLOCK + threadIdx.x; n < a.N;
         n += (int64_t)gridDim.x * CHX_BLOCK) {
        const int64_t xrow = (a.Bx == 1 ? 0 : b) * a.N + n;
        T vx = x[xrow * 7 + 0], vy = x[xrow * 7 + 2];
        if (shift) {
            vx = vx - shift[(a.Bsh == 1 ? 0 : b) * 2 + 0];
            vy = vy - shift[(a.Bsh == 1 ? 0 : b) * 2 + 1];
        }
        int jx = hist_bin<T>(ex, a.nx, vx);
        int jy = hist_bin<T>(ey, a.ny, vy);
        if (jx < 0 || jy < 0) { jx = -1; jy = -1; }
        if (INDICES) {
            ij[(b * a.N + n) * 2 + 0] = jx;
            ij[(b * a.N + n) * 2 + 1] = jy;
        } else if (jx >= 0) {
            T c = q ? fabs(q[(a.Bq == 1 ? 0 : b) * a.N + n]) : (T)1;
            if (s) c = c * s[(a.Bs == 1 ? 0 : b) * a.N + n];
            table.add((int64_t)jy * a.nx + jx, c);
        }
    }
    if (!INDICES) table.flush();
}

int cic_prepare(const chx_cic_args* p, CicDev& a) {
    if (!p || !p->x || !p->extent) return CHX_ERR_INVALID_ARG;
    if (p->ndim < 1 || p->ndim > 3 || p->B < 1 || p->N < 1 || p->B > 65535) return CHX_ERR_INVALID_ARG;
    if (p->dtype != CHX_F32 && p->dtype != CHX_F64) return CHX_ERR_DTYPE;
    if (!chx_bcast_ok(p->Bx, p->B) || !chx_bcast_ok(p->Be, p->B)) return CHX_ERR_INVALID_ARG;
    if (p->charge && !chx_bcast_ok(p->Bq, p->B)) return CHX_ERR_INVALID_ARG;
    if (p->survival && !chx_bcast_ok(p->Bs, p->B)) return CHX_ERR_INVALID_ARG;
    if (p->scale && !chx_bcast_ok(p->Bsc, p->B)) return CHX_ERR_INVALID_ARG;
    if (p->shift && !chx_bcast_ok(p->Bsh, p->B)) return CHX_ERR_INVALID_ARG;
    a.ndim = p->ndim;
    int64_t total = 1;
    for (int d = 0; d < 3; ++d) {
        a.cols[d] = d < p->ndim ? p->cols[d] : 0;
        a.bins[d] = d < p->ndim ? p->bins[d] : 1;
        if (d < p->ndim && (p->cols[d] < 0 || p->cols[d] > 6 || p->bins[d] < 1)) return CHX_ERR_INVALID_ARG;
        total *= a.bins[d];
    }
    bool custom = false;
    for (int d = 0; d < p->ndim; ++d) custom = custom || p->grid_strides[d] != 0;
    if (custom) {
        for (int d = 0; d < 3; ++d) a.gstride[d] = d < p->ndim ? p->grid_strides[d] : 0;
        a.gbatch = p->grid_batch_stride;
    } else {
        int64_t st = 1;
        for (int d = p->ndim - 1; d >= 0; --d) { a.gstride[d] = st; st *= a.bins[d]; }
        for (int d = p->ndim; d < 3; ++d) a.gstride[d] = 0;
        a.gbatch = p->grid_batch_stride ? p->grid_batch_stride : total;
    }
    a.B = p->B; a.Bx = p->Bx; a.Bq = p->Bq; a.Bs = p->Bs; a.Be = p->Be; a.Bsc = p->Bsc; a.Bsh = p->Bsh;
    a.N = p->N;
    a.abs_charge = p->abs_charge;
    return CHX_OK;
}

// grid of the kernels that combine in LDS first: ~1024 workgroups in total, each strides over many particles so that
// its table absorbs as much as possible before the flush
inline dim3 combine_grid(int64_t N, int64_t B) {
    int64_t g = (N + CHX_BLOCK - 1) / CHX_BLOCK;
    int64_t cap = 1024 / B;
    if (cap < 1) cap = 1;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return dim3((unsigned)g, (unsigned)B);
}

inline dim3 particle_grid(int64_t N, int64_t B) {
    int64_t g = (N + CHX_BLOCK - 1) / CHX_BLOCK;
    int64_t cap = 8192 / B;  // grid-stride beyond ~8k workgroups in total
    if (cap < 1) cap = 1;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return dim3((unsigned)g, (unsigned)B);
}


// =============================================================================================
// Sorted, LDS-privatised deposit (2-D / 3-D, large N) — "owner computes".
// Measured on MI355X (benchmarks/atomic_scope.hip): global float atomics saturate at ~21 G/s chip-wide
// whatever their scope or the grid size (L2 atomic units), i.e. 1e6 particles x 8 corners cost ~390 us.
// LDS atomics are orders of magnitude more plentiful, so the grid is cut into small tiles (16x16 pixels /
// 8^3 cells, doubled until <= 8192 tiles) and every tile is OWNED by one workgroup:
//   pass 1  count     every particle is assigned to each tile its 2^d-corner footprint touches (1 tile for
//                     most particles, up to 2^d at tile corners); per-workgroup LDS histogram -> counts[tile][wg]
//   pass 2  scan      exclusive prefix over (tile-major, wg-minor): deterministic slot ranges, no atomics
//                     (per-tile scan over the workgroups in its own kernel; the scan of the tile totals is redone
//                     in LDS by every workgroup of pass 3, which saves a launch)
//   pass 3  scatter   records {i_d, f_d, charge} into their tile's slot range (LDS cursors)
//   pass 4  accumulate one workgroup per tile: ds_add of the corners that fall into the OWNED cells, then a
//                     plain, coalesced `grid += tile` (no halo exchange); the overflow of hot tiles (beams focused
//                     into a few cells) is load-balanced over chunks of the record array by pass 4b
// The index / weight arithmetic is the one of cic_locate / cic_deposit_kernel above (bit-identical addends).
// A record lives in the slot range of ONE tile, so its base cell is stored relative to that tile's origin: 7 bits per
// axis (i - origin + 2 in [0, 66] for tile edges up to 64) in one word. fp32: 16 bytes in 2-D (one dwordx4 per record),
// 20 bytes in 3-D (dwordx4 + dword at element alignment) instead of 20 / 28 with full indices — the scatter pass issues
// one line request per store instruction and lane, so the record's instruction count is its price.
template <typename T, int ND>
struct alignas((ND == 2 && sizeof(T) == 4) ? 16 : sizeof(T)) CicRec {
    T f[ND];
    T c;
    uint32_t cell;
};
constexpr int kRecBits = 7, kRecBias = 2;

template <typename T, int ND>
__device__ __forceinline__ void rec_store(CicRec<T, ND>* __restrict__ dst, const CicRec<T, ND>& r) {
    if constexpr (ND == 3 && sizeof(T) == 4) {
        struct alignas(4) Quad { float v[4]; };
        *reinterpret_cast<Quad*>(dst) = Quad{{r.f[0], r.f[1], r.f[2], r.c}};
        dst->cell = r.cell;
    } else {
        *dst = r;
    }
}

template <typename T, int ND>
__device__ __forceinline__ CicRec<T, ND> rec_load(const CicRec<T, ND>* __restrict__ src) {
    if constexpr (ND == 3 && sizeof(T) == 4) {
        struct alignas(4) Quad { float v[4]; };
        const Quad q = *reinterpret_cast<const Quad*>(src);
        CicRec<T, ND> r;
        r.f[0] = q.v[0]; r.f[1] = q.v[1]; r.f[2] = q.v[2]; r.c = q.v[3];
        r.cell = src->cell;
        return r;
    } else {
        return *src;
    }
}

struct TileGeom {
    int tdim[3];   // tile edge in cells per axis (a power of two)
    int tshift[3]; // log2(tdim)
    int ntile[3];  // number of tiles per axis
    int nt;        // total tiles
};

// Small tiles balance the load (the hottest 8^3 brick of a Gaussian beam on a +-3 sigma 128^3 grid holds
// 0.3 % of the particles); edges are doubled until the per-workgroup histogram fits 32 KiB of LDS.
__host__ __device__ inline TileGeom tile_geom(int ndim, const int* bins) {
    TileGeom g;
    for (int d = 0; d < 3; ++d) g.tdim[d] = d < ndim ? (ndim == 2 ? 16 : 8) : 1;
    for (;;) {
        g.nt = 1;
        int widest = 0;
        for (int d = 0; d < 3; ++d) {
            g.ntile[d] = d < ndim ? (bins[d] + g.tdim[d] - 1) / g.tdim[d] : 1;
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

// Per-axis constants of a batch row for the sort passes: wave-uniform, read once per workgroup (cic_locate re-reads the
// extent and re-divides for every particle).
template <typename T, int ND>
struct SortAxes {
    T l[ND], r[ND], bw[ND], sc[ND], sh[ND];
    bool has_sc, has_sh;
};

template <typename T, int ND>
__device__ __forceinline__ SortAxes<T, ND> sort_axes(const CicDev& a, const T* __restrict__ extent, const T* __restrict__ scale,
                                                     const T* __restrict__ shift, int64_t b) {
    SortAxes<T, ND> ax;
    const T* ext = extent + (a.Be == 1 ? 0 : b) * ND * 2;
    ax.has_sc = scale != nullptr;
    ax.has_sh = shift != nullptr;
#pragma unroll
    for (int d = 0; d < ND; ++d) {
        ax.l[d] = ext[d * 2];
        ax.r[d] = ext[d * 2 + 1];
        ax.bw[d] = (ax.r[d] - ax.l[d]) / (T)a.bins[d];
        ax.sc[d] = ax.has_sc ? scale[(a.Bsc == 1 ? 0 : b) * ND + d] : (T)1;
        ax.sh[d] = ax.has_sh ? shift[(a.Bsh == 1 ? 0 : b) * ND + d] : (T)0;
    }
    return ax;
}

// The arithmetic of cic_locate (cloud_in_cell.py:150-172, bit for bit) in 32-bit integers: inside the extent floor(p) lies
// in [-1, bins], so the conversion is exact, and p - (T)(long long)floor(p) == p - floor(p) for every integral floor(p);
// particles outside the extent are skipped by the caller. Returns the in-extent mask.
template <typename T, int ND>
__device__ __forceinline__ bool sort_locate(const CicDev& a, const SortAxes<T, ND>& ax, const T (&raw)[ND], int (&i)[ND],
                                            T (&f)[ND]) {
    bool inside = true;
#pragma unroll
    for (int d = 0; d < ND; ++d) {
        T v = raw[d];
        if (ax.has_sc) v = v * ax.sc[d];
        if (ax.has_sh) v = v - ax.sh[d];
        inside = inside && (v >= ax.l[d]) && (v <= ax.r[d]);
        const T pb = (v - ax.l[d]) / ax.bw[d] - (T)0.5;
        const T fl = floor(pb);
        const T flc = fl < (T)-2 ? (T)-2 : (fl > (T)(a.bins[d] + 1) ? (T)(a.bins[d] + 1) : fl);
        i[d] = (int)flc;
        f[d] = pb - fl;
    }
    return inside;
}

// tiles touched per axis by the corners i, i + 1 (cells outside the grid are never deposited): [t0, t1], t1 in {t0, t0 + 1}
template <int ND>
__device__ __forceinline__ void sort_tile_range(const CicDev& a, const TileGeom& g, const int (&i)[ND], int (&t0)[3], int (&t1)[3]) {
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        if (d < ND) {
            int lo = i[d], hi = i[d] + 1;
            lo = lo < 0 ? 0 : lo;
            hi = hi > a.bins[d] - 1 ? a.bins[d] - 1 : hi;
            if (lo > hi) lo = hi;
            t0[d] = lo >> g.tshift[d];
            t1[d] = hi >> g.tshift[d];
        } else {
            t0[d] = t1[d] = 0;
        }
    }
}

constexpr int kAccTileCap = 8192;   // records of a tile handled by its owner workgroup (a +-3 sigma Gaussian stays below)
constexpr int kSortWG = 256;       // workgroups of the count / scatter passes (per batch row)
constexpr int kSortThreads = 1024;  // threads of those workgroups (latency-bound loops: many waves)
constexpr int kSortBatch = 4;       // particles per lane fetched together

// exclusive prefix sum of v[0..n) in LDS by one workgroup of kSortThreads lanes; returns the total
__device__ __forceinline__ int block_exclusive_scan(int* v, int n) {
    __shared__ int wave_sum[kSortThreads / 64];
    const int per = (n + kSortThreads - 1) / kSortThreads;
    const int lo = threadIdx.x * per, hi = (lo + per < n) ? lo + per : n;
    int sum = 0;
    for (int i = lo; i < hi; ++i) sum += v[i];
    int incl = sum;  // inclusive scan across the wave
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(incl, d, 64);
        if ((threadIdx.x & 63) >= d) incl += o;
    }
    if ((threadIdx.x & 63) == 63) wave_sum[threadIdx.x >> 6] = incl;
    __syncthreads();
    int base = 0, total = 0;
#pragma unroll
    for (int w = 0; w < kSortThreads / 64; ++w) {
        const int ws = wave_sum[w];
        if (w < (int)(threadIdx.x >> 6)) base += ws;
        total += ws;
    }
    int run = base + incl - sum;
    for (int i = lo; i < hi; ++i) { const int t = v[i]; v[i] = run; run += t; }
    __syncthreads();
    return total;
}

// pass 1 (SCATTER = false) and pass 3 (SCATTER = true) share the particle loop
template <typename T, int ND, bool SCATTER>
__global__ __launch_bounds__(kSortThreads) void cic_sort_kernel(CicDev a, TileGeom g, const T* __restrict__ x,
                                                               const T* __restrict__ q, const T* __restrict__ s,
                                                               const T* __restrict__ extent,
                                                               const T* __restrict__ scale,
                                                               const T* __restrict__ shift,
                                                               int* __restrict__ counts /*[B][kSortWG][nt]*/,
                                                               const int* __restrict__ totals /*[B][nt]*/,
                                                               int* __restrict__ tile_start /*[B][nt+1]*/,
                                                               CicRec<T, ND>* __restrict__ recs, int64_t rec_cap) {
    extern __shared__ int hist[];  // [nt] — pass 1: per-tile counters; pass 3: absolute write cursors
    const int64_t b = blockIdx.y;
    const int wg = blockIdx.x;
    int* cnt = counts + (b * kSortWG + wg) * (int64_t)g.nt;
    // The workgroup's particles are fetched FIRST — kSortBatch per lane, all loads in flight at once — so that their
    // latency overlaps the cursor set-up below (zero fill, or the scan of the 4096 tile totals), and one memory round trip
    // serves the whole chunk at the benchmark size (1e6 particles / 256 workgroups / 1024 lanes = 3.8 per lane).
    const int64_t per = (a.N + kSortWG - 1) / kSortWG;
    const int64_t n0 = (int64_t)wg * per, n1 = (n0 + per < a.N) ? n0 + per : a.N;
    const T* __restrict__ xb = x + (a.Bx == 1 ? 0 : b) * a.N * 7;
    const T* __restrict__ qb = q ? q + (a.Bq == 1 ? 0 : b) * a.N : nullptr;
    const T* __restrict__ sb = s ? s + (a.Bs == 1 ? 0 : b) * a.N : nullptr;
    T raw[kSortBatch][ND], craw[kSortBatch];
    auto fetch = [&](int64_t base) {
#pragma unroll
        for (int u = 0; u < kSortBatch; ++u) {
            const int64_t n = base + (int64_t)u * kSortThreads;
            const int64_t src = n < n1 ? n : (n1 > n0 ? n1 - 1 : 0);
#pragma unroll
            for (int d = 0; d < ND; ++d) raw[u][d] = xb[src * 7 + a.cols[d]];
            if (SCATTER) {
                T c = qb ? qb[src] : (T)1;
                if (a.abs_charge) c = fabs(c);
                if (sb) c = c * sb[src];
                craw[u] = c;
            }
        }
    };
    if (n1 > n0) fetch(n0 + threadIdx.x);
    if (!SCATTER) {
        for (int t = threadIdx.x; t < g.nt; t += kSortThreads) hist[t] = 0;
    } else {
        // pass 2b folded in: every workgroup scans the tile totals itself (nt <= 16k ints in LDS), workgroup 0
        // publishes the tile starts for pass 4; cursor = tile start + this workgroup's offset inside the tile
        __shared__ int fullest;
        if (threadIdx.x == 0) fullest = 0;
        __syncthreads();
        int mine = 0;
        for (int t = threadIdx.x; t < g.nt; t += kSortThreads) {
            const int c = totals[b * g.nt + t];
            hist[t] = c;
            mine = c > mine ? c : mine;
        }
        if (wg == 0 && mine > kAccTileCap) atomicMax(&fullest, mine);
        __syncthreads();
        const int sum = block_exclusive_scan(hist, g.nt);
        if (wg == 0) {
            for (int t = threadIdx.x; t < g.nt; t += kSortThreads) tile_start[b * (g.nt + 1) + t] = hist[t];
            if (threadIdx.x == 0) {
                tile_start[b * (g.nt + 1) + g.nt] = sum;
                // one flag per batch row behind the starts: does any tile overflow into pass 4b?
                tile_start[(int64_t)gridDim.y * (g.nt + 1) + b] = fullest > kAccTileCap ? 1 : 0;
            }
        }
        for (int t = threadIdx.x; t < g.nt; t += kSortThreads) hist[t] += cnt[t];
    }
    __syncthreads();
    const SortAxes<T, ND> ax = sort_axes<T, ND>(a, extent, scale, shift, b);
    CicRec<T, ND>* __restrict__ rb = recs + b * rec_cap;
    for (int64_t base = n0 + threadIdx.x; base < n1; base += (int64_t)kSortBatch * kSortThreads) {
        if (base != n0 + threadIdx.x) fetch(base);
#pragma unroll
        for (int u = 0; u < kSortBatch; ++u) {
            const int64_t n = base + (int64_t)u * kSortThreads;
            if (n >= n1) break;
            int pi[ND];
            T pf[ND];
            if (!sort_locate<T, ND>(a, ax, raw[u], pi, pf)) continue;
            int t0[3], t1[3];
            sort_tile_range<ND>(a, g, pi, t0, t1);
            CicRec<T, ND> r;
            if (SCATTER) {
#pragma unroll
                for (int d = 0; d < ND; ++d) r.f[d] = pf[d];
                r.c = craw[u];
            }
            for (int tx = t0[0]; tx <= t1[0]; ++tx)
                for (int ty = t0[1]; ty <= t1[1]; ++ty)
                    for (int tz = t0[2]; tz <= t1[2]; ++tz) {
                        const int tile = (tx * g.ntile[1] + ty) * g.ntile[2] + tz;
                        const int pos = atomicAdd(&hist[tile], 1);
                        if (SCATTER) {
                            const int tt[3] = {tx, ty, tz};
                            uint32_t cell = 0;
#pragma unroll
                            for (int d = 0; d < ND; ++d)
                                cell |= (uint32_t)(pi[d] - (tt[d] << g.tshift[d]) + kRecBias) << (kRecBits * d);
                            r.cell = cell;
                            if (pos < rec_cap) rec_store<T, ND>(rb + pos, r);
                        }
                    }
        }
    }
    if (!SCATTER) {
        __syncthreads();
        for (int t = threadIdx.x; t < g.nt; t += kSortThreads) cnt[t] = hist[t];
    }
}

// pass 2a: per tile, exclusive prefix over the workgroups (in place) and the tile total. counts is wg-major
// ([wg][tile]) so every access below is coalesced across the 64 tiles of a workgroup; the 256 workgroup counts of
// a tile are cut into 16 chunks of 16 (one lane each, 16 loads in flight) that meet through LDS.
constexpr int kScanChunk = 16;
constexpr int kScanTiles = 16;   // tiles per workgroup: 256-thread workgroups, nt / 16 of them (256 at 128^3) instead of
                                 // nt / 64 with 1024 threads — they find free slots next to the Green-function chain
static_assert(kSortWG == kScanChunk * 16, "scan decomposition");
__global__ __launch_bounds__(16 * kScanTiles) void cic_scan_tiles_kernel(int* __restrict__ counts, int* __restrict__ totals,
                                                                        int nt) {
    __shared__ int part[16][kScanTiles];
    const int64_t b = blockIdx.y;
    const int lane = threadIdx.x % kScanTiles, c = threadIdx.x / kScanTiles;
    const int t = blockIdx.x * kScanTiles + lane;
    int* cb = counts + b * kSortWG * (int64_t)nt;
    int v[kScanChunk];
    int run = 0;
    if (t < nt) {
#pragma unroll
        for (int i = 0; i < kScanChunk; ++i) v[i] = cb[(int64_t)(c * kScanChunk + i) * nt + t];
#pragma unroll
        for (int i = 0; i < kScanChunk; ++i) { const int x = v[i]; v[i] = run; run += x; }
    }
    part[c][lane] = run;
    __syncthreads();
    int off = 0;
    for (int cc = 0; cc < c; ++cc) off += part[cc][lane];
    if (t < nt) {
#pragma unroll
        for (int i = 0; i < kScanChunk; ++i) cb[(int64_t)(c * kScanChunk + i) * nt + t] = v[i] + off;
        if (c == 15) totals[b * nt + t] = off + run;
    }
}

// pass 4a: one workgroup per (tile, batch row); LDS tile = exactly the owned cells. Takes the first kAccTileCap records
// of its tile; what a hot tile holds beyond that is spread over many workgroups by pass 4b.
constexpr int kAccThreads = 1024;  // a hot tile is latency-bound on its record stream: many waves per tile
constexpr int kAccLightTile = 1024;  // average records per tile below which pass 4a uses 256-thread workgroups

// OVERWRITE: the tile's cells are stored (zeros included, empty tiles too) instead of added to the grid — every grid cell
// belongs to exactly one tile, so the caller needs no zero fill and the flush has no dependent load.
template <typename T, int ND, int THREADS, bool OVERWRITE>
__global__ __launch_bounds__(THREADS) void cic_accumulate_kernel(CicDev a, TileGeom g,
                                                                  const int* __restrict__ tile_start,
                                                                  const CicRec<T, ND>* __restrict__ recs,
                                                                  int64_t rec_cap, T* __restrict__ grid) {
    // The LDS tile is fp64 for both dtypes: measured on MI355X (benchmarks/lds_atomic_rate.hip) ds_add_f64
    // sustains 5.7 G lane-atomics/s per CU, ds_add_f32 only 0.78 G/s — and the sums are more accurate.
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* tile = reinterpret_cast<double*>(smem);
    const int64_t b = blockIdx.y;
    // Workgroups are dispatched in blockIdx order: the tiles are visited from the middle of the grid outwards, so that the
    // full tiles of a centred beam (thousands of records, a dozen dependent memory round trips) start in the first wave of
    // workgroups and the nearly empty ones at the grid's edge fill in behind them.
    int t;
    {
        int rem = blockIdx.x, tc[3];
        for (int d = 2; d >= 0; --d) {
            const int n = d < ND ? g.ntile[d] : 1;
            const int k = rem % n;
            rem /= n;
            const int c = (n - 1) / 2;
            tc[d] = (k & 1) ? c + (k + 1) / 2 : c - k / 2;
        }
        t = (tc[0] * (ND > 1 ? g.ntile[1] : 1) + (ND > 1 ? tc[1] : 0)) * (ND > 2 ? g.ntile[2] : 1) + (ND > 2 ? tc[2] : 0);
    }
    const int beg = tile_start[b * (g.nt + 1) + t];
    int end = tile_start[b * (g.nt + 1) + t + 1];
    if (end > rec_cap) end = (int)rec_cap;
    if (end > beg + kAccTileCap) end = beg + kAccTileCap;
    if (!OVERWRITE && beg >= end) return;  // empty tile: nothing to add
    int ld[3], org[3];       // owned extents and cell origin
    {
        int rem = t;
        for (int d = 2; d >= 0; --d) {
            int tc = 0;
            if (d < ND) { tc = rem % g.ntile[d]; rem /= g.ntile[d]; }
            ld[d] = d < ND ? g.tdim[d] : 1;
            org[d] = tc * ld[d];
        }
    }
    const int lcells = ld[0] * ld[1] * ld[2];
    for (int i = threadIdx.x; i < lcells; i += THREADS) tile[i] = 0.0;
    __syncthreads();
    const CicRec<T, ND>* rb = recs + b * rec_cap;
    auto deposit = [&](const CicRec<T, ND>& rec) {
        T wf[3][2];
        int li[3][2];
        bool ok[3][2];
#pragma unroll
        for (int d = 0; d < 3; ++d)
#pragma unroll
            for (int o = 0; o < 2; ++o) {
                if (d < ND) {
                    const int l = (int)((rec.cell >> (kRecBits * d)) & ((1u << kRecBits) - 1)) - kRecBias + o;
                    const int id = l + org[d];
                    // valid grid cell AND owned by this tile
                    ok[d][o] = (id >= 0) && (id < a.bins[d]) && (l >= 0) && (l < ld[d]);
                    li[d][o] = l;
                    wf[d][o] = o ? rec.f[d] : ((T)1.0 - rec.f[d]);
                } else {
                    ok[d][o] = (o == 0); li[d][o] = 0; wf[d][o] = (T)1;
                }
            }
        if (ND == 2) {
#pragma unroll
            for (int oy = 0; oy < 2; ++oy)
#pragma unroll
                for (int ox = 0; ox < 2; ++ox)
                    if (ok[0][ox] && ok[1][oy])
                        unsafeAtomicAdd(&tile[li[0][ox] * ld[1] + li[1][oy]], (double)(rec.c * wf[0][ox] * wf[1][oy]));
        } else {
#pragma unroll
            for (int ox = 0; ox < 2; ++ox)
#pragma unroll
                for (int oy = 0; oy < 2; ++oy)
#pragma unroll
                    for (int oz = 0; oz < 2; ++oz)
                        if (ok[0][ox] && ok[1][oy] && ok[2][oz])
                            unsafeAtomicAdd(&tile[(li[0][ox] * ld[1] + li[1][oy]) * ld[2] + li[2][oz]],
                                            (double)(rec.c * (wf[0][ox] * wf[1][oy] * wf[2][oz])));
        }
    };
    int r = beg + threadIdx.x;
    // four record loads in flight per lane before the first ds_add depends on them
    for (; r + 3 * THREADS < end; r += 4 * THREADS) {
        const CicRec<T, ND> r0 = rec_load<T, ND>(rb + r), r1 = rec_load<T, ND>(rb + r + THREADS),
                            r2 = rec_load<T, ND>(rb + r + 2 * THREADS), r3 = rec_load<T, ND>(rb + r + 3 * THREADS);
        deposit(r0); deposit(r1); deposit(r2); deposit(r3);
    }
    for (; r < end; r += THREADS) deposit(rec_load<T, ND>(rb + r));
    __syncthreads();
    // flush the owned cells: exclusive owner -> plain read-modify-write, last axis fastest (coalesced)
    T* gb = grid + b * a.gbatch;
    for (int i = threadIdx.x; i < lcells; i += THREADS) {
        const double v = tile[i];
        if (!OVERWRITE && v == 0.0) continue;
        int l[3];
        int rem = i;
        l[2] = rem & (ld[2] - 1); rem >>= g.tshift[2];   // tile edges are powers of two
        l[1] = rem & (ld[1] - 1); rem >>= g.tshift[1];
        l[0] = rem;
        bool in_grid = true;
        int64_t off = 0;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            if (d < ND) {
                const int cell = org[d] + l[d];
                in_grid = in_grid && cell < a.bins[d];
                off += (int64_t)cell * a.gstride[d];
            }
        }
        if (in_grid) gb[off] = OVERWRITE ? (T)v : (T)((double)gb[off] + v);
    }
}

// pass 4b: hot tiles. The sorted record array is cut into chunks of kAccChunk records, one workgroup per (chunk, batch
// row); it handles, for every tile overlapping its chunk, the records BEYOND the first kAccTileCap of that tile (those
// belong to pass 4a) and adds its LDS tile to the grid with atomics. For a +-3 sigma beam no tile is that full and every
// workgroup leaves after a binary search; a beam focused into one tile (1e6 particles inside 16 x 16 pixels) is spread
// over 500 workgroups instead of serialising in one (measured: 3.0 ms -> 0.1 ms for a Screen reading).
constexpr int kAccChunk = 2048;    // records per workgroup of pass 4b

template <typename T, int ND>
__global__ __launch_bounds__(kAccThreads) void cic_accumulate_hot_kernel(CicDev a, TileGeom g,
                                                                  const int* __restrict__ tile_start,
                                                                  const CicRec<T, ND>* __restrict__ recs,
                                                                  int64_t rec_cap, T* __restrict__ grid) {
    // The LDS tile is fp64 for both dtypes: measured on MI355X (benchmarks/lds_atomic_rate.hip) ds_add_f64
    // sustains 5.7 G lane-atomics/s per CU, ds_add_f32 only 0.78 G/s — and the sums are more accurate.
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* tile = reinterpret_cast<double*>(smem);
    const int64_t b = blockIdx.y;
    if (!tile_start[(int64_t)gridDim.y * (g.nt + 1) + b]) return;  // no tile of this row is hot (flag from pass 3)
    const int* ts = tile_start + b * (g.nt + 1);
    int total = ts[g.nt];
    if (total > rec_cap) total = (int)rec_cap;
    const CicRec<T, ND>* rb = recs + b * rec_cap;
    T* gb = grid + b * a.gbatch;
    for (int r0 = blockIdx.x * kAccChunk; r0 < total; r0 += gridDim.x * kAccChunk) {
    const int r1 = (r0 + kAccChunk < total) ? r0 + kAccChunk : total;
    // last tile whose start is <= r0 (empty tiles share their start with the next non-empty one, so this is the tile
    // that holds record r0)
    int lo = 0, hi = g.nt - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (ts[mid] <= r0) lo = mid; else hi = mid - 1;
    }
    {   // quick reject (the normal case): 64 lanes look at the next 64 tiles at once
        const int t = lo + (threadIdx.x & 63);
        bool hot_here = false;
        if (t < g.nt) {
            const int tb = ts[t], te = ts[t + 1];
            hot_here = tb < r1 && te > r0 && te - tb > kAccTileCap;
        }
        const int tl = lo + 64 < g.nt ? lo + 64 : g.nt;
        if (!__any(hot_here) && ts[tl] >= r1) continue;
    }
    for (int t = lo; t < g.nt && ts[t] < r1; ++t) {
        const int tbeg = ts[t], tend = ts[t + 1];
        const int hot = tbeg + kAccTileCap;  // first record of this tile that pass 4a leaves behind
        const int beg = hot > r0 ? hot : r0, end = tend < r1 ? tend : r1;
        if (beg >= end) continue;
        int ld[3], org[3];       // owned extents and cell origin
        {
            int rem = t;
            for (int d = 2; d >= 0; --d) {
                int tc = 0;
                if (d < ND) { tc = rem % g.ntile[d]; rem /= g.ntile[d]; }
                ld[d] = d < ND ? g.tdim[d] : 1;
                org[d] = tc * ld[d];
            }
        }
        const int lcells = ld[0] * ld[1] * ld[2];
        for (int i = threadIdx.x; i < lcells; i += kAccThreads) tile[i] = 0.0;
        __syncthreads();
        auto deposit = [&](const CicRec<T, ND>& rec) {
            T wf[3][2];
            int li[3][2];
            bool ok[3][2];
#pragma unroll
            for (int d = 0; d < 3; ++d)
#pragma unroll
                for (int o = 0; o < 2; ++o) {
                    if (d < ND) {
                        const int l = (int)((rec.cell >> (kRecBits * d)) & ((1u << kRecBits) - 1)) - kRecBias + o;
                        const int id = l + org[d];
                        // valid grid cell AND owned by this tile
                        ok[d][o] = (id >= 0) && (id < a.bins[d]) && (l >= 0) && (l < ld[d]);
                        li[d][o] = l;
                        wf[d][o] = o ? rec.f[d] : ((T)1.0 - rec.f[d]);
                    } else {
                        ok[d][o] = (o == 0); li[d][o] = 0; wf[d][o] = (T)1;
                    }
                }
            if (ND == 2) {
#pragma unroll
                for (int oy = 0; oy < 2; ++oy)
#pragma unroll
                    for (int ox = 0; ox < 2; ++ox)
                        if (ok[0][ox] && ok[1][oy])
                            unsafeAtomicAdd(&tile[li[0][ox] * ld[1] + li[1][oy]], (double)(rec.c * wf[0][ox] * wf[1][oy]));
            } else {
#pragma unroll
                for (int ox = 0; ox < 2; ++ox)
#pragma unroll
                    for (int oy = 0; oy < 2; ++oy)
#pragma unroll
                        for (int oz = 0; oz < 2; ++oz)
                            if (ok[0][ox] && ok[1][oy] && ok[2][oz])
                                unsafeAtomicAdd(&tile[(li[0][ox] * ld[1] + li[1][oy]) * ld[2] + li[2][oz]],
                                                (double)(rec.c * (wf[0][ox] * wf[1][oy] * wf[2][oz])));
            }
        };
        int r = beg + threadIdx.x;
        // two record loads in flight per lane before the first ds_add depends on them
        for (; r + kAccThreads < end; r += 2 * kAccThreads) {
            const CicRec<T, ND> ra = rec_load<T, ND>(rb + r), rc = rec_load<T, ND>(rb + r + kAccThreads);
            deposit(ra); deposit(rc);
        }
        for (; r < end; r += kAccThreads) deposit(rec_load<T, ND>(rb + r));
        __syncthreads();
        // add the tile to the grid (pass 4a may be writing the same cells: atomics)
        for (int i = threadIdx.x; i < lcells; i += kAccThreads) {
            const double v = tile[i];
            if (v == 0.0) continue;
            int l[3];
            int rem = i;
            l[2] = rem % ld[2]; rem /= ld[2];
            l[1] = rem % ld[1]; rem /= ld[1];
            l[0] = rem;
            bool in_grid = true;
            int64_t off = 0;
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                if (d < ND) {
                    const int cell = org[d] + l[d];
                    in_grid = in_grid && cell < a.bins[d];
                    off += (int64_t)cell * a.gstride[d];
                }
            }
            if (!in_grid) continue;
            unsafeAtomicAdd(gb + off, (T)v);
        }
        __syncthreads();
    }
    }
}

// record capacity per batch row: every particle can touch up to 2^ND tiles (only at tile corners); the
// expected multiplicity is prod(1 + 1/tdim) ~ 1.13 (2-D, 16 px) .. 1.42 (3-D, 8 cells)
template <int ND>
int64_t rec_capacity(int64_t N) { return N * (1 << ND); }

template <typename T, int ND>
size_t sorted_ws_bytes(const CicDev& a, const TileGeom& g) {
    size_t bytes = (size_t)a.B * g.nt * kSortWG * sizeof(int);          // counts
    bytes += (size_t)a.B * (2 * g.nt + 2) * sizeof(int);                // tile totals, tile starts, hot flag
    bytes = (bytes + 255) & ~(size_t)255;
    bytes += (size_t)a.B * (size_t)rec_capacity<ND>(a.N) * sizeof(CicRec<T, ND>);  // sorted records
    return bytes;
}

template <typename T, int ND>
int launch_sorted(const CicDev& a, const chx_cic_args* p, void* workspace, size_t workspace_bytes, hipStream_t s,
                  bool overwrite) {
    const TileGeom g = tile_geom(a.ndim, a.bins);
    if (workspace_bytes < sorted_ws_bytes<T, ND>(a, g)) return CHX_ERR_WORKSPACE;
    if ((size_t)g.nt * sizeof(int) > 60 * 1024) return CHX_ERR_INVALID_ARG;
    if (rec_capacity<ND>(a.N) > 0x7fffffffLL) return CHX_ERR_INVALID_ARG;
    int* counts = (int*)workspace;
    int* totals = counts + (size_t)a.B * g.nt * kSortWG;
    int* starts = totals + (size_t)a.B * g.nt;
    size_t off = ((size_t)a.B * g.nt * kSortWG + (size_t)a.B * (2 * g.nt + 2)) * sizeof(int);
    off = (off + 255) & ~(size_t)255;
    CicRec<T, ND>* recs = (CicRec<T, ND>*)((char*)workspace + off);
    const int64_t cap = rec_capacity<ND>(a.N);
    const dim3 sgrid(kSortWG, (unsigned)a.B);
    const size_t hist_bytes = (size_t)g.nt * sizeof(int);
    hipLaunchKernelGGL((cic_sort_kernel<T, ND, false>), sgrid, dim3(kSortThreads), hist_bytes, s, a, g, (const T*)p->x,
                       (const T*)p->charge, (const T*)p->survival, (const T*)p->extent, (const T*)p->scale,
                       (const T*)p->shift, counts, (const int*)totals, starts, recs, cap);
    CHX_CHECK_LAUNCH();
    hipLaunchKernelGGL(cic_scan_tiles_kernel, dim3((g.nt + kScanTiles - 1) / kScanTiles, (unsigned)a.B), dim3(16 * kScanTiles), 0, s,
                       counts, totals, g.nt);
    CHX_CHECK_LAUNCH();
    hipLaunchKernelGGL((cic_sort_kernel<T, ND, true>), sgrid, dim3(kSortThreads), hist_bytes, s, a, g, (const T*)p->x,
                       (const T*)p->charge, (const T*)p->survival, (const T*)p->extent, (const T*)p->scale,
                       (const T*)p->shift, counts, (const int*)totals, starts, recs, cap);
    CHX_CHECK_LAUNCH();
    size_t tile_bytes = sizeof(double);
    for (int d = 0; d < ND; ++d) tile_bytes *= (size_t)g.tdim[d];
    // Lightly filled tiles (a diffuse beam: ~250 records per 8^3 tile at N = 1e6 on 128^3) run as 256-thread workgroups,
    // eight per CU instead of two, so the per-workgroup zero / flush latency overlaps; fuller tiles keep 1024 threads.
    const dim3 agrid((unsigned)g.nt, (unsigned)a.B);
    const bool light = a.N / g.nt < kAccLightTile;
#define CHX_ACC_LAUNCH(THREADS, OVER)                                                                                     \
    hipLaunchKernelGGL((cic_accumulate_kernel<T, ND, THREADS, OVER>), agrid, dim3(THREADS), tile_bytes, s, a, g,          \
                       (const int*)starts, (const CicRec<T, ND>*)recs, cap, (T*)p->grid)
    if (light) { if (overwrite) CHX_ACC_LAUNCH(256, true); else CHX_ACC_LAUNCH(256, false); }
    else       { if (overwrite) CHX_ACC_LAUNCH(kAccThreads, true); else CHX_ACC_LAUNCH(kAccThreads, false); }
#undef CHX_ACC_LAUNCH
    CHX_CHECK_LAUNCH();
    unsigned nchunks = (unsigned)((cap + kAccChunk - 1) / kAccChunk);
    if (nchunks > 256) nchunks = 256;  // chunk-strided inside the kernel; it returns at once unless a tile is hot
    hipLaunchKernelGGL((cic_accumulate_hot_kernel<T, ND>), dim3(nchunks, (unsigned)a.B), dim3(kAccThreads), tile_bytes, s,
                       a, g, (const int*)starts, (const CicRec<T, ND>*)recs, cap, (T*)p->grid);
    CHX_CHECK_LAUNCH();
    return CHX_OK;
}

}  // namespace

static int deposit_direct(const chx_cic_args* p, const void* R, int64_t BR, void* stream) {
    CicDev a;
    int st = cic_prepare(p, a);
    if (st != CHX_OK) return st;
    if (!p->grid) return CHX_ERR_INVALID_ARG;
    if (R && !chx_bcast_ok(BR, a.B)) return CHX_ERR_INVALID_ARG;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid = combine_grid(a.N, a.B);
    if (p->dtype == CHX_F32)
        hipLaunchKernelGGL(cic_deposit_kernel<float>, grid, dim3(CHX_BLOCK), 0, s, a, (const float*)p->x,
                           (const float*)p->charge, (const float*)p->survival, (const float*)p->extent,
                           (const float*)p->scale, (const float*)p->shift, (float*)p->grid, (const float*)R, BR);
    else
        hipLaunchKernelGGL(cic_deposit_kernel<double>, grid, dim3(CHX_BLOCK), 0, s, a, (const double*)p->x,
                           (const double*)p->charge, (const double*)p->survival, (const double*)p->extent,
                           (const double*)p->scale, (const double*)p->shift, (double*)p->grid, (const double*)R, BR);
    CHX_CHECK_LAUNCH();
    return CHX_OK;
}

extern "C" int chx_cic_deposit(const chx_cic_args* p, void* stream) { return deposit_direct(p, nullptr, 1, stream); }

extern "C" int chx_cic_deposit_mapped(const chx_cic_args* p, const void* R, int64_t BR, void* stream) {
    if (!R) return CHX_ERR_INVALID_ARG;
    return deposit_direct(p, R, BR, stream);
}

extern "C" int chx_cic_indices(const chx_cic_args* p, int32_t* idx_out, void* frac_out, void* stream) {
    CicDev a;
    int st = cic_prepare(p, a);
    if (st != CHX_OK) return st;
    if (!idx_out || !frac_out) return CHX_ERR_INVALID_ARG;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid = particle_grid(a.N, a.B);
    if (p->dtype == CHX_F32)
        hipLaunchKernelGGL(cic_indices_kernel<float>, grid, dim3(CHX_BLOCK), 0, s, a, (const float*)p->x,
                           (const float*)p->extent, (const float*)p->scale, (const float*)p->shift,
                           idx_out, (float*)frac_out);
    else
        hipLaunchKernelGGL(cic_indices_kernel<double>, grid, dim3(CHX_BLOCK), 0, s, a, (const double*)p->x,
                           (const double*)p->extent, (const double*)p->scale, (const double*)p->shift,
                           idx_out, (double*)frac_out);
    CHX_CHECK_LAUNCH();
    return CHX_OK;
}

extern "C" int chx_cic_deposit_bwd(const chx_cic_args* p, const void* dgrid, void* dweight, void* dpos,
                                   void* stream) {
    CicDev a;
    int st = cic_prepare(p, a);
    if (st != CHX_OK) return st;
    if (!dgrid || (!dweight && !dpos)) return CHX_ERR_INVALID_ARG;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid = particle_grid(a.N, a.B);
    if (p->dtype == CHX_F32)
        hipLaunchKernelGGL(cic_bwd_kernel<float>, grid, dim3(CHX_BLOCK), 0, s, a, (const float*)p->x,
                           (const float*)p->charge, (const float*)p->survival, (const float*)p->extent,
                           (const float*)p->scale, (const float*)p->shift, (const float*)dgrid,
                           (float*)dweight, (float*)dpos);
    else
        hipLaunchKernelGGL(cic_bwd_kernel<double>, grid, dim3(CHX_BLOCK), 0, s, a, (const double*)p->x,
                           (const double*)p->charge, (const double*)p->survival, (const double*)p->extent,
                           (const double*)p->scale, (const double*)p->shift, (const double*)dgrid,
                           (double*)dweight, (double*)dpos);
    CHX_CHECK_LAUNCH();
    return CHX_OK;
}

static int hist_prepare(const chx_hist2d_args* p, HistDev& a) {
    if (!p || !p->x || !p->edges_x || !p->edges_y) return CHX_ERR_INVALID_ARG;
    if (p->B < 1 || p->N < 1 || p->nx < 1 || p->ny < 1 || p->B > 65535) return CHX_ERR_INVALID_ARG;
    if (p->dtype != CHX_F32 && p->dtype != CHX_F64) return CHX_ERR_DTYPE;
    if (!chx_bcast_ok(p->Bx, p->B)) return CHX_ERR_INVALID_ARG;
    if (p->charge && !chx_bcast_ok(p->Bq, p->B)) return CHX_ERR_INVALID_ARG;
    if (p->survival && !chx_bcast_ok(p->Bs, p->B)) return CHX_ERR_INVALID_ARG;
    if (p->shift && !chx_bcast_ok(p->Bsh, p->B)) return CHX_ERR_INVALID_ARG;
    a.B = p->B; a.Bx = p->Bx; a.Bq = p->Bq; a.Bs = p->Bs; a.Bsh = p->Bsh; a.N = p->N;
    a.nx = p->nx; a.ny = p->ny;
    return CHX_OK;
}

extern "C" int chx_hist2d(const chx_hist2d_args* p, void* stream) {
    HistDev a;
    int st = hist_prepare(p, a);
    if (st != CHX_OK) return st;
    if (!p->image) return CHX_ERR_INVALID_ARG;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid = combine_grid(a.N, a.B);
    if (p->dtype == CHX_F32)
        hipLaunchKernelGGL((hist2d_kernel<float, false>), grid, dim3(CHX_BLOCK), 0, s, a, (const float*)p->x,
                           (const float*)p->charge, (const float*)p->survival, (const float*)p->shift,
                           (const float*)p->edges_x, (const float*)p->edges_y, (float*)p->image,
                           (int32_t*)nullptr);
    else
        hipLaunchKernelGGL((hist2d_kernel<double, false>), grid, dim3(CHX_BLOCK), 0, s, a, (const double*)p->x,
                           (const double*)p->charge, (const double*)p->survival, (const double*)p->shift,
                           (const double*)p->edges_x, (const double*)p->edges_y, (double*)p->image,
                           (int32_t*)nullptr);
    CHX_CHECK_LAUNCH();
    return CHX_OK;
}

extern "C" int chx_hist2d_indices(const chx_hist2d_args* p, int32_t* ij_out, void* stream) {
    HistDev a;
    int st = hist_prepare(p, a);
    if (st != CHX_OK) return st;
    if (!ij_out) return CHX_ERR_INVALID_ARG;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid = particle_grid(a.N, a.B);
    if (p->dtype == CHX_F32)
        hipLaunchKernelGGL((hist2d_kernel<float, true>), grid, dim3(CHX_BLOCK), 0, s, a, (const float*)p->x,
                           (const float*)nullptr, (const float*)nullptr, (const float*)p->shift,
                           (const float*)p->edges_x, (const float*)p->edges_y, (float*)nullptr, ij_out);
    else
        hipLaunchKernelGGL((hist2d_kernel<double, true>), grid, dim3(CHX_BLOCK), 0, s, a, (const double*)p->x,
                           (const double*)nullptr, (const double*)nullptr, (const double*)p->shift,
                           (const double*)p->edges_x, (const double*)p->edges_y, (double*)nullptr, ij_out);
    CHX_CHECK_LAUNCH();
    return CHX_OK;
}

extern "C" size_t chx_cic_sorted_workspace_bytes(const chx_cic_args* p) {
    CicDev a;
    if (cic_prepare(p, a) != CHX_OK || a.ndim < 2) return 0;
    const TileGeom g = tile_geom(a.ndim, a.bins);
    if (p->dtype == CHX_F32) return a.ndim == 2 ? sorted_ws_bytes<float, 2>(a, g) : sorted_ws_bytes<float, 3>(a, g);
    return a.ndim == 2 ? sorted_ws_bytes<double, 2>(a, g) : sorted_ws_bytes<double, 3>(a, g);
}

static int deposit_sorted(const chx_cic_args* p, void* workspace, size_t workspace_bytes, void* stream, bool overwrite) {
    CicDev a;
    int st = cic_prepare(p, a);
    if (st != CHX_OK) return st;
    if (!p->grid || a.ndim < 2 || a.N > 0x7fffffffLL) return CHX_ERR_INVALID_ARG;
    if (!workspace) return CHX_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    if (p->dtype == CHX_F32)
        return a.ndim == 2 ? launch_sorted<float, 2>(a, p, workspace, workspace_bytes, s, overwrite)
                           : launch_sorted<float, 3>(a, p, workspace, workspace_bytes, s, overwrite);
    return a.ndim == 2 ? launch_sorted<double, 2>(a, p, workspace, workspace_bytes, s, overwrite)
                       : launch_sorted<double, 3>(a, p, workspace, workspace_bytes, s, overwrite);
}

extern "C" int chx_cic_deposit_sorted(const chx_cic_args* p, void* workspace, size_t workspace_bytes, void* stream) {
    return deposit_sorted(p, workspace, workspace_bytes, stream, false);
}

extern "C" int chx_cic_deposit_sorted_overwrite(const chx_cic_args* p, void* workspace, size_t workspace_bytes,
                                                void* stream) {
    return deposit_sorted(p, workspace, workspace_bytes, stream, true);
}

// =====================================================================================================================
// Tile-ordered beam of a chain of SpaceChargeKicks (chx_sc_tiles.h): counting sort of the particle ROWS by deposit tile
// (first kick of the chain), deposit straight from the ordered rows, and the bookkeeping that lets the gather pass re-order the
// rows when too many particles have left their tile. The index / weight arithmetic is sort_locate's
// (cloud_in_cell.py:150-172, 262-311): identical addends to chx_cic_deposit.
#include "chx_sc_tiles.h"
#include "chx_sc_geom_dev.h"

namespace {

static_assert(kScSortThreads == kSortThreads && kScSortWG == kSortWG, "block_exclusive_scan / scan decomposition are shared");

__device__ __forceinline__ int sc_home_tile(const CicDev& a, const ScTileGeom& g, const int (&pi)[3]) {
    int t = 0;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const int c = pi[d] < 0 ? 0 : (pi[d] > a.bins[d] - 1 ? a.bins[d] - 1 : pi[d]);
        t = t * g.ntile[d] + (c >> g.tshift[d]);
    }
    return t;
}

// first kick, pass 1: per-workgroup histogram of home tiles (one tile per particle)
template <typename T>
__global__ __launch_bounds__(kScSortThreads) void sc_tile_count_kernel(CicDev a, ScTileGeom g, const T* __restrict__ x,
                                                                      const T* __restrict__ extent, const T* __restrict__ scale,
                                                                      int* __restrict__ counts) {
    extern __shared__ int hist[];
    const int wg = blockIdx.x;
    const int64_t per = (a.N + kScSortWG - 1) / kScSortWG;
    const int64_t n0 = (int64_t)wg * per, n1 = (n0 + per < a.N) ? n0 + per : a.N;
    for (int t = threadIdx.x; t < g.nt; t += kScSortThreads) hist[t] = 0;
    __syncthreads();
    const SortAxes<T, 3> ax = sort_axes<T, 3>(a, extent, scale, nullptr, 0);
    for (int64_t n = n0 + threadIdx.x; n < n1; n += kScSortThreads) {
        const T raw[3] = {x[n * 7 + a.cols[0]], x[n * 7 + a.cols[1]], x[n * 7 + a.cols[2]]};
        int pi[3];
        T pf[3];
        sort_locate<T, 3>(a, ax, raw, pi, pf);
        atomicAdd(&hist[sc_home_tile(a, g, pi)], 1);
    }
    __syncthreads();
    for (int t = threadIdx.x; t < g.nt; t += kScSortThreads) counts[(int64_t)wg * g.nt + t] = hist[t];
}

// first kick, pass 3 (pass 2 is cic_scan_tiles_kernel): rows, weights, charges and the identity permutation into their tile's
// slot range
template <typename T>
__global__ __launch_bounds__(kScSortThreads) void sc_tile_scatter_kernel(
    CicDev a, ScTileGeom g, const T* __restrict__ x, const T* __restrict__ q_in, const T* __restrict__ w_in,
    const T* __restrict__ extent, const T* __restrict__ scale, const int* __restrict__ counts, const int* __restrict__ totals,
    int* __restrict__ tile_start, T* __restrict__ rows_out, T* __restrict__ ws_out, T* __restrict__ cs_out, int* __restrict__ perm_out) {
    extern __shared__ int hist[];
    const int wg = blockIdx.x;
    const int64_t per = (a.N + kScSortWG - 1) / kScSortWG;
    const int64_t n0 = (int64_t)wg * per, n1 = (n0 + per < a.N) ? n0 + per : a.N;
    for (int t = threadIdx.x; t < g.nt; t += kScSortThreads) hist[t] = totals[t];
    __syncthreads();
    const int sum = block_exclusive_scan(hist, g.nt);
    if (wg == 0) {
        for (int t = threadIdx.x; t < g.nt; t += kScSortThreads) tile_start[t] = hist[t];
        if (threadIdx.x == 0) tile_start[g.nt] = sum;
    }
    for (int t = threadIdx.x; t < g.nt; t += kScSortThreads) hist[t] += counts[(int64_t)wg * g.nt + t];
    __syncthreads();
    const SortAxes<T, 3> ax = sort_axes<T, 3>(a, extent, scale, nullptr, 0);
    for (int64_t n = n0 + threadIdx.x; n < n1; n += kScSortThreads) {
        T row[7];
#pragma unroll
        for (int j = 0; j < 7; ++j) row[j] = x[n * 7 + j];
        const T raw[3] = {row[a.cols[0]], row[a.cols[1]], row[a.cols[2]]};
        int pi[3];
        T pf[3];
        sort_locate<T, 3>(a, ax, raw, pi, pf);
        const int pos = atomicAdd(&hist[sc_home_tile(a, g, pi)], 1);
#pragma unroll
        for (int j = 0; j < 7; ++j) rows_out[(int64_t)pos * 7 + j] = row[j];
        const T w = w_in ? w_in[n] : (T)1;
        T c = q_in ? q_in[n] : (T)1;
        if (w_in) c = c * w;                       // charges = q * survival (space_charge_kick.py:556-563)
        ws_out[pos] = w;
        cs_out[pos] = c;
        perm_out[pos] = (int)n;
    }
}

// deposit: one workgroup per tile over its slot range. LDS block = the tile's cells plus the +1 layer (fp64, ds_add: see
// cic_accumulate_kernel), flushed with one float atomic per NON-ZERO cell into the chain's accumulation grid `acc` (state; all
// zero between two kicks: whoever consumes it — the first FFT pass of the convolution, or chx_sc_tile_deposit's collect pass —
// writes the zeros back). A tile without slots (3/4 of the tiles of a 3-sigma grid) returns at once: nothing to zero-fill,
// nothing to hand to neighbours. (Round 3 stored the owned cells, handed the +1 layer over through per-tile face buffers and
// merged them in a second kernel: 46 + 15 us at 1e6 particles on 128^3; the zero stores of the empty tiles alone were 16 us.)
// A misfiled particle (and what a tile holds beyond kScTileCap) adds the corners that still fall into this tile's block there
// and the others to `acc` with global atomics right here — 4096 workgroups' worth of parallelism for them; a list + a pass of
// its own behind this kernel cost 7 us at 1 % misfiled and 59 us at 17 % (C4's last kick). On the way every particle's CURRENT
// home tile is recorded (home[], newcount[] — movers to the 26 neighbours are counted in LDS first: their global atomics would
// pile up on a few lines). Slots are read eight at a time per thread: the pass is bound by the load -> ds_add chain of its
// fullest tiles.
// Measured and dropped: an LDS queue for the misfiled particles with a dense pass (one lane per corner) behind the loop, with
// and without the global atomics moved behind the last barrier — no faster at 1 % misfiled, 1.6 x slower at 25 %.
constexpr int kScDepUnroll = 8;

template <typename T, int TH>
__global__ __launch_bounds__(TH) void sc_tile_deposit_kernel(CicDev a, ScTileGeom g, ScTileHeader* __restrict__ hdr,
                                                             const int* __restrict__ tile_start2 /*[2][nt+1]*/,
                                                             const T* __restrict__ src, const T* __restrict__ cs2 /*[2][N]*/,
                                                             const T* __restrict__ extent, const T* __restrict__ scale,
                                                             T* __restrict__ cross /* = acc */, uint16_t* __restrict__ home,
                                                             int* __restrict__ newcount, int* __restrict__ mis_slots,
                                                             int parts_shift, ScGeoSums rider, int diag) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* blk = reinterpret_cast<double*>(smem);
    __shared__ int nbr[27];
    __shared__ int nstay, nmis;
    __shared__ T geo_s[kScGeoValues];                    // rider: the geometry this workgroup formed (chx_sc_geom_dev.h)
    __shared__ double pot_s[1];
    // 2^parts_shift workgroups share a tile (dense tiles — few tiles, many particles: the reference's default 32^3 grid with 1e6
    // particles has 64 tiles of 15 000 slots on average, 60 000 in the occupied ones): each takes a contiguous share of the
    // tile's slot range; everything a workgroup leaves behind is a sum (charge, counters), so the shares simply add up
    const int t = (int)(blockIdx.x >> parts_shift), part = (int)(blockIdx.x & ((1u << parts_shift) - 1u));
    // the slot range of both parities is fetched next to the parity itself: one round trip instead of two in front of everything
    // this workgroup does (a workgroup of an empty tile is two round trips long otherwise)
    const int b0 = tile_start2[t], e0 = tile_start2[t + 1], b1 = tile_start2[g.nt + 1 + t], e1 = tile_start2[g.nt + 2 + t];
    // (a gather pass that re-ordered the rows leaves scatter_now set: its copy of the arrays is in force from this kick on; the flag
    // itself is taken back by the bookkeeping step behind this pass, or by the geometry kernel in front of it where one runs)
    const int par = (hdr->parity ^ hdr->scatter_now) & 1;
    const T* __restrict__ cs = cs2 + (int64_t)par * a.N;
    const int TX = g.tdim[0], TY = g.tdim[1], TZ = g.tdim[2];
    const int BY = TZ + 1, BX = (TY + 1) * BY, ncell = (TX + 1) * BX;
    int tc[3], org[3];
    {
        int rem = t;
        tc[2] = rem % g.ntile[2]; rem /= g.ntile[2];
        tc[1] = rem % g.ntile[1]; rem /= g.ntile[1];
        tc[0] = rem;
#pragma unroll
        for (int d = 0; d < 3; ++d) org[d] = tc[d] << g.tshift[d];
    }
    int beg = par ? b1 : b0, end = par ? e1 : e0;
    if (parts_shift) {
        const int share = (end - beg + (1 << parts_shift) - 1) >> parts_shift;
        beg += part * share;
        end = (beg + share < end) ? beg + share : end;
    }
    const int lim = (end - beg > kScTileCap) ? beg + kScTileCap : end;
    // rider: this kick's grid geometry from the sums the previous gather pass left — every workgroup with particles, redundantly
    // (the (TX + 1)^3 block is zeroed behind it: its first (TH / 16 + 1) * 8 doubles serve the reduction); workgroup 0, with or
    // without particles, also leaves it behind for the kernels that follow on this stream
    if (rider.sums && (blockIdx.x == 0 || end > beg)) sc_geo_from_sums<T, TH>(rider, blk, geo_s, pot_s, blockIdx.x == 0);
    if (end <= beg) return;                             // no slots (3/4 of the tiles of a 3-sigma grid)
    if (threadIdx.x < 27) nbr[threadIdx.x] = 0;
    if (threadIdx.x == 0) { nstay = 0; nmis = 0; }
    for (int i = threadIdx.x; i < ncell; i += TH) blk[i] = 0.0;
    __syncthreads();
    const SortAxes<T, 3> ax = rider.sums ? sort_axes<T, 3>(a, geo_s + 11, geo_s + 8, nullptr, 0) : sort_axes<T, 3>(a, extent, scale, nullptr, 0);
    int stay = 0, mis = 0;
    for (int r0 = beg; r0 < end; r0 += TH * kScDepUnroll) {
        T raw[kScDepUnroll][3], cq[kScDepUnroll];
#pragma unroll
        for (int u = 0; u < kScDepUnroll; ++u) {
            const int r = r0 + u * TH + (int)threadIdx.x;
            const int rr = r < end ? r : beg;
#pragma unroll
            for (int d = 0; d < 3; ++d) raw[u][d] = src[(int64_t)rr * 7 + a.cols[d]];
            cq[u] = cs[rr];
        }
#pragma unroll
        for (int u = 0; u < kScDepUnroll; ++u) {
            const int r = r0 + u * TH + (int)threadIdx.x;
            if (r >= end) continue;
            int pi[3];
            T pf[3];
            const bool inside = sort_locate<T, 3>(a, ax, raw[u], pi, pf);
            int hc[3];
#pragma unroll
            for (int d = 0; d < 3; ++d)
                hc[d] = (pi[d] < 0 ? 0 : (pi[d] > a.bins[d] - 1 ? a.bins[d] - 1 : pi[d])) >> g.tshift[d];   // = sc_home_tile
            const int h = (hc[0] * g.ntile[1] + hc[1]) * g.ntile[2] + hc[2];
            if (!(diag & 4)) home[r] = (uint16_t)h;
            if (h == t) ++stay;
            else if (!(diag & 4)) {
                const int dx = hc[0] - tc[0], dy = hc[1] - tc[1], dz = hc[2] - tc[2];
                if (dx >= -1 && dx <= 1 && dy >= -1 && dy <= 1 && dz >= -1 && dz <= 1) atomicAdd(&nbr[(dx + 1) * 9 + (dy + 1) * 3 + dz + 1], 1);
                else atomicAdd(&newcount[h], 1);
            }
            const int l[3] = {pi[0] - org[0], pi[1] - org[1], pi[2] - org[2]};
            if (!(l[0] >= 0 && l[0] < TX && l[1] >= 0 && l[1] < TY && l[2] >= 0 && l[2] < TZ)) ++mis;
            if (!inside || (diag & 1)) continue;        // outside the extent: no charge (cloud_in_cell.py:289-311)
            // ONE code path for filed and misfiled particles (a branch for the latter cost every wave that held a single one of
            // them the whole detour: +9 us on the kernel at 1 % misfiled). Per corner: in the grid? (the reference's in-range
            // mask) -> in this tile's block? -> LDS, else the `cross` grid with a global atomic (the arithmetic of
            // cic_deposit_kernel); the overflow of a hot tile (r >= lim) goes to `cross` altogether.
            const T c = cq[u];
            const T wgt[3][2] = {{(T)1.0 - pf[0], pf[0]}, {(T)1.0 - pf[1], pf[1]}, {(T)1.0 - pf[2], pf[2]}};
            bool ok[3][2], inb[3][2];
#pragma unroll
            for (int d = 0; d < 3; ++d)
#pragma unroll
                for (int o = 0; o < 2; ++o) {
                    ok[d][o] = pi[d] + o >= 0 && pi[d] + o < a.bins[d];
                    inb[d][o] = l[d] + o >= 0 && l[d] + o <= TX;         // tiles are cubes (sc_tile_prepare)
                }
            const bool lds_ok = r < lim;
#pragma unroll
            for (int ox = 0; ox < 2; ++ox)
#pragma unroll
                for (int oy = 0; oy < 2; ++oy)
#pragma unroll
                    for (int oz = 0; oz < 2; ++oz) {
                        if (!(ok[0][ox] && ok[1][oy] && ok[2][oz])) continue;
                        const T v = c * (wgt[0][ox] * wgt[1][oy] * wgt[2][oz]);
                        if (lds_ok && inb[0][ox] && inb[1][oy] && inb[2][oz])
                            unsafeAtomicAdd(&blk[(l[0] + ox) * BX + (l[1] + oy) * BY + (l[2] + oz)], (double)v);
                        else
                            unsafeAtomicAdd(cross + (int64_t)(pi[0] + ox) * a.gstride[0] + (int64_t)(pi[1] + oy) * a.gstride[1] +
                                                (int64_t)(pi[2] + oz) * a.gstride[2], v);
                    }
        }
    }
    if (stay) atomicAdd(&nstay, stay);
    if (mis) atomicAdd(&nmis, mis);
    __syncthreads();
    if (threadIdx.x < 27 && nbr[threadIdx.x]) {        // a counted neighbour holds the particle's clamped cell: it exists
        const int k = threadIdx.x;
        const int nb = ((tc[0] + k / 9 - 1) * g.ntile[1] + (tc[1] + (k / 3) % 3 - 1)) * g.ntile[2] + (tc[2] + k % 3 - 1);
        atomicAdd(&newcount[nb], nbr[k]);
    }
    if (threadIdx.x == 32 && nstay) atomicAdd(&newcount[t], nstay);
    for (int i = threadIdx.x; i < ncell; i += TH) {
        const T v = (T)blk[i];
        if (v == (T)0 || (diag & 2)) continue;          // (cells of the +1 layer beyond the grid edge never receive anything)
        const int lz = i % BY, ly = (i / BY) % (TY + 1), lx = i / BX;
        unsafeAtomicAdd(cross + (int64_t)(org[0] + lx) * a.gstride[0] + (int64_t)(org[1] + ly) * a.gstride[1] +
                            (int64_t)(org[2] + lz) * a.gstride[2], v);
    }
    if (threadIdx.x == 64 && nmis) atomicAdd(&mis_slots[t & (kScMisSlots - 1)], nmis);   // the beam's misfiled particles (-> schedule kernel)
}

// deposit, the bookkeeping behind the pass (ONE workgroup): sc_tile_schedule_block (chx_sc_tiles.h) as a kernel of its own. Inside a
// chain of kicks it rides in the first FFT pass of the convolution instead (chx_sc_convolve_halo_chain).
__global__ __launch_bounds__(256) void sc_tile_schedule_kernel(ScScheduleArgs a) {
    __shared__ int part[257];
    sc_tile_schedule_block(a, part);
}

// chx_sc_tile_deposit as an entry point of its own: move the accumulated charge into the caller's grid and leave zeros behind
// (inside chx_sc_kick_sorted the first FFT pass of the convolution reads `acc` itself and writes the zeros back)
template <typename T>
__global__ __launch_bounds__(256) void sc_tile_collect_kernel(T* __restrict__ acc, T* __restrict__ grid, int64_t n,
                                                             int* __restrict__ newcount, int nt, int* __restrict__ mis) {
    // (the counters the gather pass of a chain kick would put back to zero)
    for (int k = (int)blockIdx.x * 256 + threadIdx.x; k < nt; k += (int)gridDim.x * 256) newcount[k] = 0;
    if (blockIdx.x == 0 && threadIdx.x < kScMisSlots) mis[threadIdx.x] = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const T v = acc[i];
        grid[i] = v;
        if (v != (T)0) acc[i] = (T)0;
    }
}

int sc_tile_prepare(int64_t N, const int32_t* bins, int dtype, CicDev& a, ScTileGeom& g) {
    if (N < 1 || N > 0x7fffffffLL / 8 || !bins) return CHX_ERR_INVALID_ARG;
    if (dtype != CHX_F32 && dtype != CHX_F64) return CHX_ERR_DTYPE;
    g = sc_tile_geom(bins);
    for (int d = 0; d < 3; ++d)
        if (bins[d] < 8 || (bins[d] & (g.tdim[d] - 1)) || g.tdim[d] != kScTdim) return CHX_ERR_INVALID_ARG;
    if ((size_t)g.nt * sizeof(int) > 60 * 1024 || g.nt > 65535) return CHX_ERR_INVALID_ARG;
    a.ndim = 3;
    a.cols[0] = 0; a.cols[1] = 2; a.cols[2] = 4;
    for (int d = 0; d < 3; ++d) a.bins[d] = bins[d];
    a.gstride[0] = (int64_t)bins[1] * bins[2];
    a.gstride[1] = bins[2];
    a.gstride[2] = 1;
    a.gbatch = (int64_t)bins[0] * bins[1] * bins[2];
    a.B = a.Bx = a.Bq = a.Bs = a.Be = a.Bsc = a.Bsh = 1;
    a.N = N;
    a.abs_charge = 0;
    return CHX_OK;
}

template <typename T>
int sc_tile_sort_launch(const CicDev& a, const ScTileGeom& g, const ScTileLayout& L, char* st, const void* x_in, const void* charge,
                        const void* survival, const void* extent, const void* scale, hipStream_t s) {
    int* counts = (int*)(st + L.counts);
    int* totals = (int*)(st + L.totals);
    const size_t hist_bytes = (size_t)g.nt * sizeof(int);
    if (hipMemsetAsync(st + L.hdr, 0, L.zero_bytes, s) != hipSuccess) return CHX_ERR_LAUNCH;     // header, newcount
    hipLaunchKernelGGL(sc_tile_count_kernel<T>, dim3(kScSortWG), dim3(kScSortThreads), hist_bytes, s, a, g, (const T*)x_in,
                       (const T*)extent, (const T*)scale, counts);
    CHX_CHECK_LAUNCH();
    hipLaunchKernelGGL(cic_scan_tiles_kernel, dim3((g.nt + kScanTiles - 1) / kScanTiles, 1), dim3(16 * kScanTiles), 0, s, counts, totals,
                       g.nt);
    CHX_CHECK_LAUNCH();
    hipLaunchKernelGGL(sc_tile_scatter_kernel<T>, dim3(kScSortWG), dim3(kScSortThreads), hist_bytes, s, a, g, (const T*)x_in,
                       (const T*)charge, (const T*)survival, (const T*)extent, (const T*)scale, counts, totals,
                       (int*)(st + L.tile_start[0]), (T*)(st + L.rows_tmp), (T*)(st + L.ws[0]), (T*)(st + L.cs[0]), (int*)(st + L.perm[0]));
    CHX_CHECK_LAUNCH();
    return CHX_OK;
}

ScScheduleArgs sc_schedule_args(const ScTileGeom& g, const ScTileLayout& L, char* st, int64_t N, int allow_reorder) {
    ScScheduleArgs sa;
    sa.g = g;
    sa.N = N;
    sa.hdr = (ScTileHeader*)(st + L.hdr);
    sa.newcount = (const int*)(st + L.newcount);
    sa.mis = (const int*)(st + L.mis);
    sa.cursor = (int*)(st + L.cursor);
    sa.tile_start2 = (int*)(st + L.tile_start[0]);
    sa.allow_reorder = allow_reorder;
    return sa;
}

template <typename T>
int sc_tile_deposit_launch(const CicDev& a, const ScTileGeom& g, const ScTileLayout& L, char* st, const void* rows, const void* extent,
                           const void* scale, void* grid /*nullptr: leave the charge in the state's accumulation grid*/,
                           int allow_reorder, hipStream_t s, const ScGeoSums* rider = nullptr, bool schedule = true) {
    ScTileHeader* hdr = (ScTileHeader*)(st + L.hdr);
    const size_t blk_bytes = (size_t)(g.tdim[0] + 1) * (g.tdim[1] + 1) * (g.tdim[2] + 1) * sizeof(double);
    // threads per tile: CHX_TUNE_DEPOSIT_THREADS (benchmarks only) picks 256 / 512 / 1024
    static const int th = [] { const char* e = getenv("CHX_TUNE_DEPOSIT_THREADS"); const int v = e ? atoi(e) : 0; return (v == 256 || v == 512 || v == 1024) ? v : 256; }();
    auto kern = th == 1024 ? sc_tile_deposit_kernel<T, 1024> : th == 512 ? sc_tile_deposit_kernel<T, 512> : sc_tile_deposit_kernel<T, 256>;
    // workgroups per tile: an occupied tile of a 3-sigma grid holds ~4 N / nt slots; shares of at most ~2048 slots
    // (benchmarks/sc_chain_density.py: one workgroup per tile costs 841 us per kick at 1e6 particles on 32^3)
    static const int share = [] { const char* e = getenv("CHX_TUNE_DEPOSIT_SHARE"); const int v = e ? atoi(e) : 0; return v >= 64 ? v : 2048; }();
    // CHX_TUNE_DEPOSIT_DIAG (timing experiments, WRONG results): 1 no LDS atomics, 2 no flush, 4 no home / neighbour counters
    static const int diag = [] { const char* e = getenv("CHX_TUNE_DEPOSIT_DIAG"); return e ? atoi(e) : 0; }();
    int parts_shift = 0;
    while (parts_shift < 6 && ((a.N * 4 / g.nt) >> parts_shift) > share) ++parts_shift;
    ScGeoSums rd = ScGeoSums();
    if (rider) rd = *rider;
    // (the rider's reduction borrows the tile block's LDS: (th / 16 + 1) * 8 doubles)
    if (rd.sums && blk_bytes < (size_t)(th / 16 + 1) * 8 * sizeof(double)) return CHX_ERR_INVALID_ARG;
    hipLaunchKernelGGL(kern, dim3((unsigned)g.nt << parts_shift), dim3(th), blk_bytes, s, a, g, hdr, (const int*)(st + L.tile_start[0]),
                       (const T*)rows, (const T*)(st + L.cs[0]), (const T*)extent, (const T*)scale, (T*)(st + L.cross),
                       (uint16_t*)(st + L.home), (int*)(st + L.newcount), (int*)(st + L.mis), parts_shift, rd, diag);
    CHX_CHECK_LAUNCH();
    if (schedule) {
        const ScScheduleArgs sa = sc_schedule_args(g, L, st, a.N, allow_reorder);
        hipLaunchKernelGGL(sc_tile_schedule_kernel, dim3(1), dim3(256), 0, s, sa);
        CHX_CHECK_LAUNCH();
    }
    if (grid) {
        const int64_t n = a.gbatch;
        hipLaunchKernelGGL(sc_tile_collect_kernel<T>, dim3((unsigned)chx_grid_for(n, 256 * 8, 4096)), dim3(256), 0, s, (T*)(st + L.cross),
                           (T*)grid, n, (int*)(st + L.newcount), g.nt, (int*)(st + L.mis));
        CHX_CHECK_LAUNCH();
    }
    return CHX_OK;
}

}  // namespace

extern "C" size_t chx_sc_tile_state_bytes(int64_t N, const int32_t* bins, int dtype) {
    CicDev a;
    ScTileGeom g;
    if (sc_tile_prepare(N, bins, dtype, a, g) != CHX_OK) return 0;
    return sc_tile_layout(N, bins, dtype).total;
}

extern "C" int chx_sc_tile_sort(const void* x_in, const void* charge, const void* survival, const void* extent, const void* scale,
                                int64_t N, const int32_t* bins, int dtype, void* state, size_t state_bytes, void* stream) {
    CicDev a;
    ScTileGeom g;
    int st = sc_tile_prepare(N, bins, dtype, a, g);
    if (st != CHX_OK) return st;
    if (!x_in || !extent || !state) return CHX_ERR_INVALID_ARG;
    const ScTileLayout L = sc_tile_layout(N, bins, dtype);
    if (state_bytes < L.total) return CHX_ERR_WORKSPACE;
    return dtype == CHX_F32 ? sc_tile_sort_launch<float>(a, g, L, (char*)state, x_in, charge, survival, extent, scale, (hipStream_t)stream)
                            : sc_tile_sort_launch<double>(a, g, L, (char*)state, x_in, charge, survival, extent, scale, (hipStream_t)stream);
}

extern "C" int chx_sc_tile_deposit(const void* rows, const void* extent, const void* scale, int64_t N, const int32_t* bins, int dtype,
                                   void* state, size_t state_bytes, void* grid, int allow_reorder, void* stream) {
    CicDev a;
    ScTileGeom g;
    int st = sc_tile_prepare(N, bins, dtype, a, g);
    if (st != CHX_OK) return st;
    if (!extent || !state || !grid) return CHX_ERR_INVALID_ARG;
    const ScTileLayout L = sc_tile_layout(N, bins, dtype);
    if (state_bytes < L.total) return CHX_ERR_WORKSPACE;
    if (!rows) rows = (char*)state + L.rows_tmp;       // the rows the sort of the first kick wrote
    return dtype == CHX_F32 ? sc_tile_deposit_launch<float>(a, g, L, (char*)state, rows, extent, scale, grid, allow_reorder, (hipStream_t)stream)
                            : sc_tile_deposit_launch<double>(a, g, L, (char*)state, rows, extent, scale, grid, allow_reorder, (hipStream_t)stream);
}

// the same deposit, the charge left in the chain's own accumulation grid (*acc_out: [gx][gy][gz] of `dtype` inside `state`): the
// consumer must leave zeros behind (chx_sc_convolve_halo_consume does)
extern "C" int chx_sc_tile_deposit_acc(const void* rows, const void* extent, const void* scale, int64_t N, const int32_t* bins, int dtype,
                                       void* state, size_t state_bytes, int allow_reorder, void** acc_out, void* stream) {
    CicDev a;
    ScTileGeom g;
    int st = sc_tile_prepare(N, bins, dtype, a, g);
    if (st != CHX_OK) return st;
    if (!extent || !state) return CHX_ERR_INVALID_ARG;
    const ScTileLayout L = sc_tile_layout(N, bins, dtype);
    if (state_bytes < L.total) return CHX_ERR_WORKSPACE;
    if (!rows) rows = (char*)state + L.rows_tmp;
    if (acc_out) *acc_out = (char*)state + L.cross;
    return dtype == CHX_F32 ? sc_tile_deposit_launch<float>(a, g, L, (char*)state, rows, extent, scale, nullptr, allow_reorder, (hipStream_t)stream)
                            : sc_tile_deposit_launch<double>(a, g, L, (char*)state, rows, extent, scale, nullptr, allow_reorder, (hipStream_t)stream);
}

// chx_sc_tile_deposit_acc inside a chain kick (chx_sc_tiles.h)
int chx_sc_tile_deposit_chain(const void* rows, const void* extent, const void* scale, int64_t N, const int32_t* bins, int dtype,
                              void* state, size_t state_bytes, int allow_reorder, const ScGeoSums* rider, bool schedule, void* stream) {
    CicDev a;
    ScTileGeom g;
    int st = sc_tile_prepare(N, bins, dtype, a, g);
    if (st != CHX_OK) return st;
    if ((!rider && !extent) || !state) return CHX_ERR_INVALID_ARG;
    const ScTileLayout L = sc_tile_layout(N, bins, dtype);
    if (state_bytes < L.total) return CHX_ERR_WORKSPACE;
    if (!rows) rows = (char*)state + L.rows_tmp;
    return dtype == CHX_F32 ? sc_tile_deposit_launch<float>(a, g, L, (char*)state, rows, extent, scale, nullptr, allow_reorder, (hipStream_t)stream, rider, schedule)
                            : sc_tile_deposit_launch<double>(a, g, L, (char*)state, rows, extent, scale, nullptr, allow_reorder, (hipStream_t)stream, rider, schedule);
}

// the bookkeeping step's arguments for a state buffer (chx_sc_convolve_halo_chain's rider)
int chx_sc_schedule_args(int64_t N, const int32_t* bins, int dtype, void* state, size_t state_bytes, int allow_reorder, ScScheduleArgs* out) {
    CicDev a;
    ScTileGeom g;
    int st = sc_tile_prepare(N, bins, dtype, a, g);
    if (st != CHX_OK) return st;
    if (!state || !out) return CHX_ERR_INVALID_ARG;
    const ScTileLayout L = sc_tile_layout(N, bins, dtype);
    if (state_bytes < L.total) return CHX_ERR_WORKSPACE;
    *out = sc_schedule_args(g, L, (char*)state, N, allow_reorder);
    return CHX_OK;
}

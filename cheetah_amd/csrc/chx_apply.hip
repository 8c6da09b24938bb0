// chx_apply.hip — linear (affine 7x7) tracking kernels for gfx950.
//
// Replaces `new_particles = incoming.particles @ tm.mT` (cheetah/accelerator/element.py:182),
// the element-by-element loop of Segment.track (segment.py:571-572) and the per-particle part of
// Cavity.track (cavity.py:112-151,220-226).
//
// Design (HBM-bound, 56 B/particle fp32, 112 B fp64):
//  * the reference's AoS layout particles[B][N][7] is kept (28-byte rows are not 16-B aligned),
//    so a workgroup streams a TILE of rows as contiguous 16-byte vectors (global_load_dwordx4,
//    fully coalesced) into LDS, every lane then picks its own rows at dword stride 7
//    (gcd(7,32)=1 -> conflict-free ds_read_b32), the results go back through the same LDS tile and
//    leave as 16-byte vector stores;
//  * the 7x7 map is wave-uniform for a tile that lies inside one batch row -> read through the
//    scalar cache (s_load), only tiles straddling a batch boundary fall back to per-lane loads;
//  * fp32 rows are evaluated as an fmaf chain in the order j=0..6 (same sequence in the single
//    pass, element-wise and fused kernels, so those agree bit for bit).
#include <cstdlib>
#include <type_traits>

#include "chx_common.h"
#include "chx_cic_dev.h"

namespace {

template <typename T> __device__ __forceinline__ T chx_fma(T a, T b, T c);
template <> __device__ __forceinline__ float chx_fma<float>(float a, float b, float c) { return fmaf(a, b, c); }
template <> __device__ __forceinline__ double chx_fma<double>(double a, double b, double c) { return fma(a, b, c); }

// y = R x with R row-major 7x7 (49 values). R may live in SGPRs (uniform pointer).
template <typename T>
__device__ __forceinline__ void apply7(const T* __restrict__ R, const T (&x)[7], T (&y)[7]) {
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        T acc = R[i * 7] * x[0];
#pragma unroll
        for (int j = 1; j < 7; ++j) acc = chx_fma<T>(R[i * 7 + j], x[j], acc);
        y[i] = acc;
    }
}

// Cavity per-particle epilogue (cavity.py:135-151, 220-226), evaluated in fp64.
// c = [a, b, kbeta0, phi, cosphi, T566, T556, T555]; tau/delta are the INCOMING coordinates.
template <typename T>
__device__ __forceinline__ void cavity_epilogue(const double* __restrict__ c, const T (&x)[7], T (&y)[7]) {
    const double tau = (double)x[4], delta = (double)x[5];
    const double dnew = delta * c[0] + c[1] * (cos(-tau * c[2] + c[3]) - c[4]);
    const double tnew = (double)y[4] + (c[5] * delta * delta + c[6] * tau * delta + c[7] * tau * tau);
    y[5] = (T)dnew;
    y[4] = (T)tnew;
}

// ---- generic multi-map kernel -------------------------------------------------------------
// MODE 0: x_out = R[0] x_in                      (single pass)
// MODE 1: x_out = R[E-1] ... R[0] x_in           (fused run, particle stays in registers)
// MODE 2: single pass + cavity epilogue
// Output rows are flat over B*N. Input is flat too (Bx == B) or shared (Bx == 1, handled by
// indexing with n only). TP = particles per tile, PPT = TP / CHX_BLOCK.
template <typename T, int PPT, int MODE>
__global__ __launch_bounds__(CHX_BLOCK) void apply_tile_kernel(
    const T* x_in, const T* __restrict__ R, T* x_out,
    const double* __restrict__ coeffs, int64_t B, int64_t Bx, int64_t BR, int64_t N, int E,
    int in_vec_ok, int out_vec_ok) {
    constexpr int TP = PPT * CHX_BLOCK;
    __shared__ __attribute__((aligned(16))) T lds[TP * 7];

    // tiles are laid out per batch row so that a tile never straddles two rows:
    // tiles_per_row = ceil(N / TP); blockIdx.x = b * tiles_per_row + t
    const int64_t tiles_per_row = (N + TP - 1) / TP;
    const int64_t b = blockIdx.x / tiles_per_row;
    const int64_t t = blockIdx.x - b * tiles_per_row;
    const int64_t n0 = t * TP;
    const int np = (int)((N - n0 < TP) ? (N - n0) : TP);

    const int64_t in_row = (Bx == 1) ? 0 : b;
    const T* gin = x_in + (in_row * N + n0) * 7;
    T* gout = x_out + (b * N + n0) * 7;
    // vector path needs the tile start 16-B aligned: base aligned (checked on host) and
    // (row*N + n0)*7*sizeof(T) % 16 == 0. n0*7*sizeof(T) is a multiple of 16 by construction.
    const bool in_vec = in_vec_ok && (((in_row * N * 7 * (int64_t)sizeof(T)) & 15) == 0);
    const bool out_vec = out_vec_ok && (((b * N * 7 * (int64_t)sizeof(T)) & 15) == 0);

    // streaming pass: bypass L2 allocation unless the input row is shared by several batch rows (re-read by others)
    const bool nt_in = !(Bx == 1 && B > 1);
    tile_load<T, TP>(gin, lds, np * 7, in_vec, nt_in);
    __syncthreads();

    const int64_t rb = (BR == 1) ? 0 : b;
    if (MODE == 1) {
        // fused run: all PPT rows of the lane stay in registers; the element loop is OUTSIDE the row loop so
        // that every map is fetched into SGPRs once per lane and reused for PPT x 49 FMAs
        if constexpr (std::is_same<T, float>::value && (PPT % 2 == 0)) {
            // fp32: two particles of the lane share one 64-bit register pair, so every step of the fmaf chain is ONE
            // v_pk_fma_f32 for both (the map entry is a scalar operand): 49 packed FMAs per pair and element, the
            // same per-particle operation order as apply7 -> bit-identical results, twice the FMA issue rate
            chx_v2f xp[PPT / 2][7];
#pragma unroll
            for (int q = 0; q < PPT / 2; ++q) {
                const int p0 = threadIdx.x + (2 * q) * CHX_BLOCK, p1 = p0 + CHX_BLOCK;
#pragma unroll
                for (int j = 0; j < 7; ++j) {
                    xp[q][j].x = (p0 < np) ? lds[p0 * 7 + j] : 0.f;
                    xp[q][j].y = (p1 < np) ? lds[p1 * 7 + j] : 0.f;
                }
            }
            for (int e = 0; e < E; ++e) {
                const float* __restrict__ Re = R + ((int64_t)e * BR + rb) * 49;
#pragma unroll
                for (int q = 0; q < PPT / 2; ++q) {
                    chx_v2f y[7];
#pragma unroll
                    for (int i = 0; i < 7; ++i) {
                        chx_v2f acc = xp[q][0] * Re[i * 7];
#pragma unroll
                        for (int j = 1; j < 7; ++j) {
                            const chx_v2f r = {Re[i * 7 + j], Re[i * 7 + j]};
                            acc = __builtin_elementwise_fma(r, xp[q][j], acc);
                        }
                        y[i] = acc;
                    }
#pragma unroll
                    for (int j = 0; j < 7; ++j) xp[q][j] = y[j];
                }
            }
#pragma unroll
            for (int q = 0; q < PPT / 2; ++q) {
                const int p0 = threadIdx.x + (2 * q) * CHX_BLOCK, p1 = p0 + CHX_BLOCK;
#pragma unroll
                for (int j = 0; j < 7; ++j) {
                    if (p0 < np) lds[p0 * 7 + j] = xp[q][j].x;
                    if (p1 < np) lds[p1 * 7 + j] = xp[q][j].y;
                }
            }
            __syncthreads();
            tile_store<T, TP>(gout, lds, np * 7, out_vec, true);
            return;
        }
        T xs[PPT][7];
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int p = threadIdx.x + k * CHX_BLOCK;
#pragma unroll
            for (int j = 0; j < 7; ++j) xs[k][j] = (p < np) ? lds[p * 7 + j] : (T)0;
        }
        for (int e = 0; e < E; ++e) {
            const T* __restrict__ Re = R + ((int64_t)e * BR + rb) * 49;
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                T y[7];
                apply7<T>(Re, xs[k], y);
#pragma unroll
                for (int j = 0; j < 7; ++j) xs[k][j] = y[j];
            }
        }
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int p = threadIdx.x + k * CHX_BLOCK;
            if (p < np) {
#pragma unroll
                for (int j = 0; j < 7; ++j) lds[p * 7 + j] = xs[k][j];
            }
        }
        __syncthreads();
        tile_store<T, TP>(gout, lds, np * 7, out_vec, true);
        return;
    }
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const int p = threadIdx.x + k * CHX_BLOCK;
        if (p < np) {
            T x[7], y[7];
#pragma unroll
            for (int j = 0; j < 7; ++j) x[j] = lds[p * 7 + j];
            if (MODE == 1) {
                for (int e = 0; e < E; ++e) {
                    const T* __restrict__ Re = R + ((int64_t)e * BR + rb) * 49;
                    apply7<T>(Re, x, y);
#pragma unroll
                    for (int j = 0; j < 7; ++j) x[j] = y[j];
                }
            } else {
                const T* __restrict__ Rb = R + rb * 49;
                apply7<T>(Rb, x, y);
                if (MODE == 2) cavity_epilogue<T>(coeffs + b * CHX_CAV_NCOEF, x, y);
            }
#pragma unroll
            for (int j = 0; j < 7; ++j) lds[p * 7 + j] = y[j];
        }
    }
    __syncthreads();
    tile_store<T, TP>(gout, lds, np * 7, out_vec, true);
}

// ---- one map per batch row, wave-private staging (MODE 0 only) ------------------------------------------------------------
// The kernel above stages a tile per WORKGROUP: two barriers per tile. Here every wave owns 64 * PPT consecutive rows —
// a whole number of 16-byte chunks — loads them into its own LDS slice, applies the map and streams them out: no workgroup
// barrier at all, the four waves of a workgroup run independently. Needs 16-byte aligned batch rows on both sides.
template <typename T, int PPT, int THREADS = CHX_BLOCK>
__global__ __launch_bounds__(THREADS) void apply_wave_kernel(const T* __restrict__ x_in, const T* __restrict__ R,
                                                               T* __restrict__ x_out, int64_t B, int64_t Bx, int64_t BR,
                                                               int64_t N) {
    using V = typename chx_vec16<T>::type;
    constexpr int VN = chx_vec16<T>::n;
    constexpr int TP = PPT * THREADS;
    constexpr int WP = PPT * 64, WE = WP * 7, WV = WE / VN;
    __shared__ __attribute__((aligned(16))) T lds[TP * 7];
    const int64_t tiles_per_row = (N + TP - 1) / TP;
    const int64_t b = blockIdx.x / tiles_per_row;
    const int64_t t = blockIdx.x - b * tiles_per_row;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t n0 = t * TP + wave * WP;                 // first row of this wave
    if (n0 >= N) return;
    const int valid = (int)((N - n0 < WP) ? (N - n0) : WP);
    const int vchunks = valid * 7 / VN;
    const int64_t in_row = (Bx == 1) ? 0 : b;
    const T* __restrict__ gin = x_in + (in_row * N + n0) * 7;
    T* __restrict__ gout = x_out + (b * N + n0) * 7;
    T* wl = lds + wave * WE;
    const bool nt_in = !(Bx == 1 && B > 1);
    {
        const V* __restrict__ gv = reinterpret_cast<const V*>(gin);
        V* lv = reinterpret_cast<V*>(wl);
#pragma unroll
        for (int c = 0; c < (WV + 63) / 64; ++c) {
            const int v = c * 64 + lane;
            if (v < vchunks) lv[v] = nt_in ? chx_nt_load(gv + v) : gv[v];
        }
        for (int e = vchunks * VN + lane; e < valid * 7; e += 64) wl[e] = gin[e];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const T* __restrict__ Rb = R + ((BR == 1) ? 0 : b) * 49;
    T y[PPT][7];
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const int p = k * 64 + lane;
        T x[7];
#pragma unroll
        for (int j = 0; j < 7; ++j) x[j] = (p < valid) ? wl[p * 7 + j] : (T)0;
        apply7<T>(Rb, x, y[k]);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const int p = k * 64 + lane;
        if (p < valid) {
#pragma unroll
            for (int j = 0; j < 7; ++j) wl[p * 7 + j] = y[k][j];
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    {
        V* __restrict__ gv = reinterpret_cast<V*>(gout);
        const V* lv = reinterpret_cast<const V*>(wl);
#pragma unroll
        for (int c = 0; c < (WV + 63) / 64; ++c) {
            const int v = c * 64 + lane;
            if (v < vchunks) chx_nt_store(lv[v], gv + v);
        }
        for (int e = vchunks * VN + lane; e < valid * 7; e += 64) gout[e] = wl[e];
    }
}

// A lane's PPT particles. float32 with an even PPT: two particles per 64-bit register pair (particle 2 p in .x, 2 p + 1 in .y) FROM
// LOAD TO STORE, so that a map is 49 v_pk_fma_f32 per pair and nothing else (packed anew per item, the 28 + 28 register moves
// around every map were as many instructions as the map itself).
template <typename T, int PPT, bool PAIRS = (std::is_same<T, float>::value && PPT % 2 == 0)>
struct LaneRows {
    T v[PPT][7];
    __device__ __forceinline__ T get(int k, int j) const { return v[k][j]; }
    __device__ __forceinline__ void set(int k, int j, T val) { v[k][j] = val; }
};
template <int PPT>
struct LaneRows<float, PPT, true> {
    chx_v2f v[PPT / 2][7];
    __device__ __forceinline__ float get(int k, int j) const { return (k & 1) ? v[k >> 1][j].y : v[k >> 1][j].x; }
    __device__ __forceinline__ void set(int k, int j, float val) {
        if (k & 1) v[k >> 1][j].y = val;
        else v[k >> 1][j].x = val;
    }
};

// ---- shared-input kernel (Bx == 1, B > 1): one x tile, many maps ---------------------------
// grid = (tiles over N, batch chunks). Each block keeps its particles in registers and loops
// over its chunk of batch rows; writes dominate (28 B per (batch, particle) in fp32).
template <typename T, int PPT>
__global__ __launch_bounds__(CHX_BLOCK) void apply_shared_kernel(
    const T* __restrict__ x_in, const T* __restrict__ R, T* __restrict__ x_out, int64_t B,
    int64_t N, int64_t rows_per_chunk, int in_vec_ok, int out_vec_ok) {
    constexpr int TP = PPT * CHX_BLOCK;
    // two output tiles: row b is written to one while row b - 1 still leaves from the other -> one barrier per row
    __shared__ __attribute__((aligned(16))) T lds2[2][TP * 7];
    T* lds = lds2[0];
    const int64_t n0 = (int64_t)blockIdx.x * TP;
    const int np = (int)((N - n0 < TP) ? (N - n0) : TP);
    const int64_t b0 = (int64_t)blockIdx.y * rows_per_chunk;
    const int64_t b1 = (b0 + rows_per_chunk < B) ? b0 + rows_per_chunk : B;

    tile_load<T, TP>(x_in + n0 * 7, lds, np * 7, in_vec_ok != 0);
    __syncthreads();
    T x[PPT][7];
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const int p = threadIdx.x + k * CHX_BLOCK;
#pragma unroll
        for (int j = 0; j < 7; ++j) x[k][j] = (p < np) ? lds[p * 7 + j] : (T)0;
    }
    __syncthreads();  // everyone has its rows in registers before tile 0 is overwritten
    for (int64_t b = b0; b < b1; ++b) {
        const T* __restrict__ Rb = R + b * 49;
        T* tile = lds2[(b - b0) & 1];
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int p = threadIdx.x + k * CHX_BLOCK;
            if (p < np) {
                T y[7];
                apply7<T>(Rb, x[k], y);
#pragma unroll
                for (int j = 0; j < 7; ++j) tile[p * 7 + j] = y[j];
            }
        }
        __syncthreads();
        const bool out_vec = out_vec_ok && (((b * N * 7 * (int64_t)sizeof(T)) & 15) == 0);
        tile_store<T, TP>(x_out + (b * N + n0) * 7, tile, np * 7, out_vec, true);
    }
}

// The same with wave-private staging: every wave keeps 64 * PPT consecutive particles, writes its outgoing rows into its own
// LDS slice and streams them out as 16-byte chunks (64 rows of 28 / 56 bytes are a whole number of chunks) — no workgroup
// barrier per batch row, so the four waves of a workgroup drift apart and their stores overlap the others' arithmetic.
// Requires every batch row of the output to start on a 16-byte boundary (N * 7 * sizeof(T) % 16 == 0, aligned base).
template <typename T, int PPT>
__global__ __launch_bounds__(CHX_BLOCK) void apply_shared_wave_kernel(
    const T* __restrict__ x_in, const T* __restrict__ R, T* __restrict__ x_out, int64_t B,
    int64_t N, int64_t rows_per_chunk, int in_vec_ok) {
    using V = typename chx_vec16<T>::type;
    constexpr int VN = chx_vec16<T>::n;
    constexpr int TP = PPT * CHX_BLOCK;
    constexpr int WP = PPT * 64;                 // particles per wave
    constexpr int WE = WP * 7;                   // elements per wave
    constexpr int WV = WE / VN;                  // 16-byte chunks per wave
    __shared__ __attribute__((aligned(16))) T lds[TP * 7];
    const int64_t n0 = (int64_t)blockIdx.x * TP;
    const int np = (int)((N - n0 < TP) ? (N - n0) : TP);
    const int64_t b0 = (int64_t)blockIdx.y * rows_per_chunk;
    const int64_t b1 = (b0 + rows_per_chunk < B) ? b0 + rows_per_chunk : B;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;

    tile_load<T, TP>(x_in + n0 * 7, lds, np * 7, in_vec_ok != 0);
    __syncthreads();
    LaneRows<T, PPT> x;                           // (float32: two particles per register pair, the maps as packed FMAs)
    T* wl = lds + wave * WE;                      // this wave's slice: rows wave * WP + [0, WP)
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const int p = wave * WP + k * 64 + lane;
#pragma unroll
        for (int j = 0; j < 7; ++j) x.set(k, j, (p < np) ? wl[(k * 64 + lane) * 7 + j] : (T)0);
    }
    // from here on a wave touches only its own slice
    const int valid = (np - wave * WP < 0) ? 0 : ((np - wave * WP < WP) ? (np - wave * WP) : WP);   // rows of this wave that exist
    const int vchunks = valid * 7 / VN;           // whole chunks inside the valid rows
    for (int64_t b = b0; b < b1; ++b) {
        const T* __restrict__ Rb = R + b * 49;
        if constexpr (std::is_same<T, float>::value && PPT % 2 == 0) {
            // apply7's fmaf chain for the two particles of a pair in ONE v_pk_fma_f32 per step (same order per particle: same bits)
#pragma unroll
            for (int r = 0; r < 7; ++r) {
                chx_v2f y[PPT / 2];
#pragma unroll
                for (int pr = 0; pr < PPT / 2; ++pr) y[pr] = x.v[pr][0] * Rb[r * 7];
#pragma unroll
                for (int j = 1; j < 7; ++j) {
                    const chx_v2f m = {Rb[r * 7 + j], Rb[r * 7 + j]};
#pragma unroll
                    for (int pr = 0; pr < PPT / 2; ++pr) y[pr] = __builtin_elementwise_fma(m, x.v[pr][j], y[pr]);
                }
#pragma unroll
                for (int pr = 0; pr < PPT / 2; ++pr) {
                    wl[((2 * pr) * 64 + lane) * 7 + r] = y[pr].x;
                    wl[((2 * pr + 1) * 64 + lane) * 7 + r] = y[pr].y;
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                T xi[7], y[7];
#pragma unroll
                for (int j = 0; j < 7; ++j) xi[j] = x.get(k, j);
                apply7<T>(Rb, xi, y);
#pragma unroll
                for (int j = 0; j < 7; ++j) wl[(k * 64 + lane) * 7 + j] = y[j];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        T* __restrict__ gout = x_out + (b * N + n0 + wave * WP) * 7;
        V* __restrict__ gv = reinterpret_cast<V*>(gout);
        const V* lv = reinterpret_cast<const V*>(wl);
#pragma unroll
        for (int c = 0; c < (WV + 63) / 64; ++c) {
            const int v = c * 64 + lane;
            if (v < vchunks) chx_nt_store(lv[v], gv + v);
        }
        if (vchunks < WV) {                       // the last tile of a row: a few elements beyond the last whole chunk
            for (int e = vchunks * VN + lane; e < valid * 7; e += 64) gout[e] = wl[e];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

template <typename T> struct tile_cfg;
template <> struct tile_cfg<float> { static constexpr int PPT = 2; };   // 512 rows, 14 KiB LDS
template <> struct tile_cfg<double> { static constexpr int PPT = 1; };  // 256 rows, 14 KiB LDS

template <typename T, int PPT, int MODE>
int launch_tiles_ppt(const void* x_in, const void* R, void* x_out, const double* coeffs, int64_t B,
                     int64_t Bx, int64_t BR, int64_t N, int E, hipStream_t s) {
    constexpr int TP = PPT * CHX_BLOCK;
    const int64_t tiles = ((N + TP - 1) / TP) * B;
    if (tiles > 0x7fffffffLL) return CHX_ERR_INVALID_ARG;
    hipLaunchKernelGGL((apply_tile_kernel<T, PPT, MODE>), dim3((unsigned)tiles), dim3(CHX_BLOCK), 0, s,
                       (const T*)x_in, (const T*)R, (T*)x_out, coeffs, B, Bx, BR, N, E,
                       (int)chx_aligned16(x_in), (int)chx_aligned16(x_out));
    CHX_CHECK_LAUNCH();
    return CHX_OK;
}

// Tile size: measured on MI355X (benchmarks/apply_variants.hip, fp32): 512-row tiles win while the
// working set is Infinity-Cache resident (5.76 vs 5.71 TB/s at N = 1e6), 256-row tiles win once the
// launch streams from HBM (5.74 vs 5.42 TB/s at N = 1.6e7; a float4 copy of the same bytes: 5.83 TB/s).
template <typename T, int PPT, int THREADS = CHX_BLOCK>
int launch_wave(const void* x_in, const void* R, void* x_out, int64_t B, int64_t Bx, int64_t BR, int64_t N, hipStream_t s) {
    constexpr int TP = PPT * THREADS;
    const int64_t tiles = ((N + TP - 1) / TP) * B;
    if (tiles > 0x7fffffffLL) return CHX_ERR_INVALID_ARG;
    hipLaunchKernelGGL((apply_wave_kernel<T, PPT, THREADS>), dim3((unsigned)tiles), dim3(THREADS), 0, s, (const T*)x_in, (const T*)R,
                       (T*)x_out, B, Bx, BR, N);
    CHX_CHECK_LAUNCH();
    return CHX_OK;
}

template <typename T, int MODE>
int launch_tiles(const void* x_in, const void* R, void* x_out, const double* coeffs, int64_t B,
                 int64_t Bx, int64_t BR, int64_t N, int E, hipStream_t s) {
    constexpr int PPT = tile_cfg<T>::PPT;
    if (MODE == 0 && B * N * 7 * (int64_t)sizeof(T) > (int64_t)96 * 1024 * 1024) {
        // the launch streams from HBM: single-wave workgroups of 64 rows, no barriers, when the batch rows are 16-byte aligned
        // on both sides — measured at N = 1.6e7, fp32: 156.1 us (256-row workgroup tiles) -> 150.0 (four independent waves
        // per workgroup) -> 146.6 (one wave per workgroup): 5.74 -> 6.11 TB/s. While the working set is Infinity-Cache
        // resident (N = 1e6) the workgroup-staged 512-row tiles are faster (9.1 us against 9.3 - 11.0 us for every variant).
        const bool rows_aligned = chx_aligned16(x_in) && chx_aligned16(x_out) &&
                                  ((N * 7 * (int64_t)sizeof(T)) % 16 == 0 || B == 1);
        if (rows_aligned) return launch_wave<T, 1, 64>(x_in, R, x_out, B, Bx, BR, N, s);
        if (PPT > 1) return launch_tiles_ppt<T, 1, MODE>(x_in, R, x_out, coeffs, B, Bx, BR, N, E, s);
    }
    if (MODE == 0 && B * N * 7 * (int64_t)sizeof(T) <= (int64_t)14 * 1024 * 1024 + 700 * 1024) {
        // a small beam (what a rank of a strong-scaling run holds: 1.25e5 - 5e5 particles): a few hundred workgroup tiles leave
        // most CUs with one dependent load -> barrier -> store chain; four independent waves per workgroup, each staging its own
        // 64 rows, overlap them. Measured, 100 launches back to back (benchmarks/strong_leg_trace.py): 3e5 particles 0.459 ->
        // 0.411 ms, 4e5 0.619 -> 0.520; from 6e5 on the workgroup tiles win (0.668 vs 0.690; 1e6: 0.899 vs 1.092). Below ~1.5e5
        // the step sits on the launch floor either way (3.7 - 3.8 us per launch)
        const bool rows_aligned = chx_aligned16(x_in) && chx_aligned16(x_out) &&
                                  ((N * 7 * (int64_t)sizeof(T)) % 16 == 0 || B == 1);
        if (rows_aligned) return launch_wave<T, 1, 256>(x_in, R, x_out, B, Bx, BR, N, s);
    }
    // fused run: VALU-bound; 4 rows per lane amortise each map's scalar loads over 4 x 49 FMAs
    if (MODE == 1 && E >= 4 && N >= 4 * CHX_BLOCK * 64)
        return launch_tiles_ppt<T, 2 * PPT, MODE>(x_in, R, x_out, coeffs, B, Bx, BR, N, E, s);
    return launch_tiles_ppt<T, PPT, MODE>(x_in, R, x_out, coeffs, B, Bx, BR, N, E, s);
}

template <typename T>
int launch_shared(const void* x_in, const void* R, void* x_out, int64_t B, int64_t N, hipStream_t s) {
    constexpr int PPT = tile_cfg<T>::PPT;
    constexpr int TP = PPT * CHX_BLOCK;
    const int64_t tiles = (N + TP - 1) / TP;
    // enough blocks to fill 256 CUs x ~8 blocks, but keep >= 8 rows per chunk for x reuse
    int64_t chunks = (2048 + tiles - 1) / tiles;
    if (chunks < 1) chunks = 1;
    if (chunks > B) chunks = B;
    int64_t rows = (B + chunks - 1) / chunks;
    if (rows < 8 && B >= 8) rows = 8;
    chunks = (B + rows - 1) / rows;
    if (tiles > 0x7fffffffLL || chunks > 65535) return CHX_ERR_INVALID_ARG;
    if (chx_aligned16(x_out) && (N * 7 * (int64_t)sizeof(T)) % 16 == 0) {
        // every batch row of the output starts on a 16-byte boundary: wave-private staging, 16 bytes per lane and row in
        // flight (measured at B = 4096 x N = 1e5, fp32: rows per lane 1 / 2 / 4 / 8 -> 2.06 / 2.03 / 1.99 / 1.98 ms with
        // ~8192 workgroups, against 2.25 ms for the workgroup-staged kernel below)
        constexpr int WPPT = 16 / (int)sizeof(T);
        constexpr int WTP = WPPT * CHX_BLOCK;
        const int64_t wtiles = (N + WTP - 1) / WTP;
        int64_t wchunks = (8192 + wtiles - 1) / wtiles;
        if (wchunks > B) wchunks = B;
        int64_t wrows = (B + wchunks - 1) / wchunks;
        if (wrows < 8 && B >= 8) wrows = 8;
        wchunks = (B + wrows - 1) / wrows;
        if (wtiles > 0x7fffffffLL || wchunks > 65535) return CHX_ERR_INVALID_ARG;
        hipLaunchKernelGGL((apply_shared_wave_kernel<T, WPPT>), dim3((unsigned)wtiles, (unsigned)wchunks), dim3(CHX_BLOCK), 0, s,
                           (const T*)x_in, (const T*)R, (T*)x_out, B, N, wrows, (int)chx_aligned16(x_in));
    } else
        hipLaunchKernelGGL((apply_shared_kernel<T, PPT>), dim3((unsigned)tiles, (unsigned)chunks),
                           dim3(CHX_BLOCK), 0, s, (const T*)x_in, (const T*)R, (T*)x_out, B, N, rows,
                           (int)chx_aligned16(x_in), (int)chx_aligned16(x_out));
    CHX_CHECK_LAUNCH();
    return CHX_OK;
}

int check_common(const void* x_in, const void* R, const void* x_out, int64_t B, int64_t Bx,
                 int64_t BR, int64_t N, int dtype) {
    if (!x_in || !R || !x_out || B < 1 || N < 1) return CHX_ERR_INVALID_ARG;
    if (!chx_bcast_ok(Bx, B) || !chx_bcast_ok(BR, B)) return CHX_ERR_INVALID_ARG;
    if (dtype != CHX_F32 && dtype != CHX_F64) return CHX_ERR_DTYPE;
    const size_t esz = dtype == CHX_F32 ? 4 : 8;
    if ((reinterpret_cast<uintptr_t>(x_in) % esz) || (reinterpret_cast<uintptr_t>(x_out) % esz) ||
        (reinterpret_cast<uintptr_t>(R) % esz))
        return CHX_ERR_MISALIGNED;
    return CHX_OK;
}

}  // namespace

extern "C" int chx_apply_affine7(const void* x_in, const void* R, void* x_out, int64_t B, int64_t Bx,
                                 int64_t BR, int64_t N, int dtype, void* stream) {
    int st = check_common(x_in, R, x_out, B, Bx, BR, N, dtype);
    if (st != CHX_OK) return st;
    hipStream_t s = (hipStream_t)stream;
    if (Bx == 1 && B > 1 && BR == B) {
        return dtype == CHX_F32 ? launch_shared<float>(x_in, R, x_out, B, N, s)
                                : launch_shared<double>(x_in, R, x_out, B, N, s);
    }
    return dtype == CHX_F32 ? launch_tiles<float, 0>(x_in, R, x_out, nullptr, B, Bx, BR, N, 1, s)
                            : launch_tiles<double, 0>(x_in, R, x_out, nullptr, B, Bx, BR, N, 1, s);
}

extern "C" int chx_track_fused(const void* x_in, const void* R, void* x_out, int64_t E, int64_t B,
                               int64_t Bx, int64_t BR, int64_t N, int dtype, void* stream) {
    int st = check_common(x_in, R, x_out, B, Bx, BR, N, dtype);
    if (st != CHX_OK) return st;
    if (E < 1 || E > 0x7fffffff) return CHX_ERR_INVALID_ARG;
    hipStream_t s = (hipStream_t)stream;
    return dtype == CHX_F32 ? launch_tiles<float, 1>(x_in, R, x_out, nullptr, B, Bx, BR, N, (int)E, s)
                            : launch_tiles<double, 1>(x_in, R, x_out, nullptr, B, Bx, BR, N, (int)E, s);
}

extern "C" int chx_track_elementwise(const void* x_in, const void* R, void* x_out, void* scratch,
                                     int64_t E, int64_t B, int64_t Bx, int64_t BR, int64_t N,
                                     int dtype, void* stream) {
    int st = check_common(x_in, R, x_out, B, Bx, BR, N, dtype);
    if (st != CHX_OK) return st;
    if (E < 1) return CHX_ERR_INVALID_ARG;
    const size_t esz = dtype == CHX_F32 ? 4 : 8;
    const char* Rp = (const char*)R;
    // pass 0 goes x_in -> x_out, every later pass updates x_out IN PLACE: a workgroup reads its whole tile into LDS before
    // it writes the same rows back, and no other workgroup touches them. Measured on MI355X at 1e6 particles: 8.9 us per
    // pass in place vs 9.3 us ping-ponging between x_out and scratch (half the footprint in L2 / Infinity Cache).
    // `scratch` is kept in the signature for ABI stability and is not used.
    (void)scratch;
    const void* src = x_in;
    int64_t src_B = Bx;
    for (int64_t e = 0; e < E; ++e) {
        st = chx_apply_affine7(src, Rp + (size_t)e * (size_t)BR * 49 * esz, x_out, B, src_B, BR, N, dtype, stream);
        if (st != CHX_OK) return st;
        src = x_out;
        src_B = B;
    }
    return CHX_OK;
}

extern "C" int chx_cavity_track(const void* x_in, const void* R, const double* coeffs, void* x_out,
                                int64_t B, int64_t Bx, int64_t N, int dtype, void* stream) {
    int st = check_common(x_in, R, x_out, B, Bx, B, N, dtype);
    if (st != CHX_OK) return st;
    if (!coeffs) return CHX_ERR_INVALID_ARG;
    hipStream_t s = (hipStream_t)stream;
    return dtype == CHX_F32 ? launch_tiles<float, 2>(x_in, R, x_out, coeffs, B, Bx, B, N, 1, s)
                            : launch_tiles<double, 2>(x_in, R, x_out, coeffs, B, Bx, B, N, 1, s);
}

// Cavity.track for one beam and a cavity whose settings are device scalars, in ONE call: chx_cavity_prepare_scalars (map,
// coefficients, outgoing energy: one thread) + the particle pass of chx_cavity_track. workspace: 49 dtype values + 8 doubles.
extern "C" size_t chx_cavity_track_scalars_workspace_bytes(void) { return 49 * sizeof(double) + 8 + CHX_CAV_NCOEF * sizeof(double); }

extern "C" int chx_cavity_track_scalars(const void* x_in, const void* const* param_ptrs, const void* energy, int kind,
                                        double mass_eV, double n_charges, int64_t N, int dtype, void* x_out, void* energy_out,
                                        const void* s_in, void* s_out, void* workspace, size_t workspace_bytes, void* stream) {
    if (!x_in || !x_out || !workspace || workspace_bytes < chx_cavity_track_scalars_workspace_bytes() || N < 1)
        return CHX_ERR_INVALID_ARG;
    if ((reinterpret_cast<uintptr_t>(workspace) & 7) != 0) return CHX_ERR_MISALIGNED;
    void* R = workspace;
    double* coeffs = reinterpret_cast<double*>((char*)workspace + 49 * sizeof(double) + 8);
    int st = chx_cavity_prepare_scalars(param_ptrs, energy, kind, mass_eV, n_charges, dtype, R, coeffs, energy_out, s_in, s_out,
                                        stream);
    if (st != CHX_OK) return st;
    return chx_cavity_track(x_in, R, coeffs, x_out, 1, 1, N, dtype, stream);
}

// ---- a stretch of lattice, every particle through all its items in registers (see lattice_prepare_kernel, chx_build.hip) ----
// items: [n_items][4] int64 = {type 0 linear run / 1 active cavity, ...}; Rs[n_items][49] (T in double-sized slots);
// coeffs[n_items][8]. Per item the arithmetic of apply_tile_kernel MODE 0 / MODE 2 on coordinates rounded to T — what storing
// the beam behind every element and loading it again gives: bit-identical to element-by-element tracking.
namespace {
// An item of type 2 is an active beam position monitor (bpm.py:77-87): it reads the weighted means of x and y of the beam AT
// that point of the lattice and lets the beam pass. Every wave leaves its sums of w, w x and w y there (fp64, slot by slot);
// lattice_bpm_finalize_kernel forms reading = (T)(sum / W) - misalignment for all monitors of the stretch.
// An item of type 3 is an active aperture (aperture.py:90-135): survival *= inside(x, y) in T, the arithmetic of
// aperture_kernel (chx_aperture.hip) — the weights of the monitors behind it are the reduced ones, and the pass writes the
// outgoing survival probabilities.
// An item of type 4 is an active Screen (screen.py:187-239, 241-344): the rows, charges and survival probabilities of the beam AT
// that point are recorded (the screen's copy of the beam, unshifted) and — flag bit 0 — every particle adds |q| w to its four
// pixels of the image, which the preparation launch zeroed: the arithmetic of cic_deposit_kernel on (x - misalignment) through the
// same per-workgroup combining table, the extent derived from the pixel size (screen_extent_axis).
struct ApplyScreens {
    chx_lattice_screen s[CHX_LATTICE_MAX_SCREENS];
    const void* charge;      // [N] or NULL (= 1)
};

// SCREENS: 0 = no screen items, 1 = screens that record only (no image buffer in this call: none of the combining table's 32 KiB
// of LDS, which would leave four workgroups per CU), 2 = screens with images, 3 = record + the one-pass sums of the recorded
// beam's moments (chx_lattice_screen.mom_partials: a beam property of the screen's beam then needs no pass over its rows —
// d sigma_x(screen) / d k1, BASELINE config C5: one launch less per step, 32 MB less to read)
// STAGED (small beams, stretches of at most kApplyStagedItems items): the workgroup copies the item table, this row's maps and the
// cavities' coefficient rows into LDS next to its particle tile — one round of loads — and walks the items out of LDS. A beam of 1e5
// particles is one or two workgroups per CU: nothing hides the scalar loads in front of every item (its type, then its 49 map entries
// in up to four pieces), and a 100-element lattice with 25 monitors is 51 items: 26 us of a pass whose arithmetic is a fifth of that.
constexpr int kApplyStagedItems = 96;
constexpr int kMomSlots = 64;       // sets of moment sums a screen's particle pass adds into (chx_lattice_screen.mom_partials)
constexpr int kApplyMapStride = 52;

template <typename T, int PPT, int SCREENS, bool STAGED>
__global__ __launch_bounds__(CHX_BLOCK) void lattice_apply_kernel(const T* x_in, T* x_out, const int64_t* __restrict__ items_g, int n_items,
                                                                 const double* __restrict__ Rs, const double* __restrict__ coeffs_g,
                                                                 int64_t N, int in_vec_ok, int out_vec_ok,
                                                                 const T* __restrict__ survival, double* __restrict__ bpm_ws, int diag,
                                                                 const int64_t* __restrict__ ptrs, T* __restrict__ survival_out,
                                                                 int shared_in /*x_in is ONE beam of N particles for all rows*/,
                                                                 int64_t Bm /*rows of lattice settings: 1, or gridDim.y*/,
                                                                 int shared_sv /*survival is ONE row of N weights for all rows*/,
                                                                 ApplyScreens scr) {
    constexpr int TP = PPT * CHX_BLOCK;
    __shared__ __attribute__((aligned(16))) T lds[TP * 7];
    // blockIdx.y = beam (a vectorised ParticleBeam of gridDim.y beams of N particles under ONE lattice setting and energy: the
    // same maps for all of them); tiles do not straddle beams, so a wave's sums at a monitor belong to one beam
    const int64_t beam = blockIdx.y;
    const int64_t t0 = (int64_t)blockIdx.x * TP;
    const int64_t n0 = beam * N + t0;                      // first flat particle of this tile
    const int np = (int)((N - t0 < TP) ? (N - t0) : TP);
    const bool row_vec = ((beam * N * 7 * (int64_t)sizeof(T)) & 15) == 0;
    // (a beam shared by the rows of a scan of lattice settings is re-read by every row: no streaming hint then)
    tile_load<T, TP>(x_in + (shared_in ? t0 : n0) * 7, lds, np * 7, in_vec_ok != 0 && (shared_in || row_vec), !shared_in);
    extern __shared__ __attribute__((aligned(16))) unsigned char stage_raw[];
    T* maps_s = reinterpret_cast<T*>(stage_raw);                                              // [n_items][kApplyMapStride]
    double* coeffs_s = reinterpret_cast<double*>(maps_s + (STAGED ? n_items * kApplyMapStride : 0));   // [n_items][CHX_CAV_NCOEF]
    int64_t* items_s = reinterpret_cast<int64_t*>(coeffs_s + (STAGED ? n_items * CHX_CAV_NCOEF : 0));  // [n_items][4]
    const int64_t* items = STAGED ? items_s : items_g;
    if constexpr (STAGED) {
        const int64_t mrow0 = (Bm == 1 ? 0 : beam);
        for (int w = threadIdx.x; w < n_items * 49; w += CHX_BLOCK) {
            const int it = w / 49, q = w - it * 49;
            maps_s[it * kApplyMapStride + q] = reinterpret_cast<const T*>(Rs + ((int64_t)it * Bm + mrow0) * 49)[q];
        }
        for (int w = threadIdx.x; w < n_items * CHX_CAV_NCOEF; w += CHX_BLOCK) {
            const int it = w / CHX_CAV_NCOEF, q = w - it * CHX_CAV_NCOEF;
            coeffs_s[w] = coeffs_g[((int64_t)it * Bm + mrow0) * CHX_CAV_NCOEF + q];
        }
        for (int w = threadIdx.x; w < n_items * 4; w += CHX_BLOCK) items_s[w] = items_g[w];
    }
    __syncthreads();
    LaneRows<T, PPT> x;
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const int p = threadIdx.x + k * CHX_BLOCK;
#pragma unroll
        for (int j = 0; j < 7; ++j) x.set(k, j, (p < np) ? lds[p * 7 + j] : (T)0);
    }
    // survival probabilities of this lane's particles (0 beyond the beam): the monitors' weights, what the apertures reduce
    T sv[PPT];
    const int64_t nw = (int64_t)gridDim.x * (CHX_BLOCK / 64);            // waves per beam
    const int64_t wslot = beam * nw + (int64_t)blockIdx.x * (CHX_BLOCK / 64) + (threadIdx.x >> 6);
    const int64_t nw_all = nw * gridDim.y;
    if (diag) {
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int p = threadIdx.x + k * CHX_BLOCK;
            sv[k] = (p < np) ? (survival ? survival[(shared_sv ? t0 : n0) + p] : (T)1) : (T)0;
        }
    }
    // SCREENS == 3: the beam's first row rides along (every lane the same values): the common centre of the moment sums at a screen
    T x0[7];
    if constexpr (SCREENS == 3) {
        const T* r0 = x_in + (shared_in ? 0 : beam * N) * 7;
#pragma unroll
        for (int j = 0; j < 7; ++j) x0[j] = r0[j];
    }
    // the wave's sum of weights stands from monitor to monitor until an aperture thins them (a wave sum of doubles is ~100
    // cycles: at 4e8 particle rows the monitors, not HBM, bound the pass)
    bool sw_known = false;
    double sw = 0.0;
    for (int i = 0; i < n_items; ++i) {
        const int type = (int)items[i * 4];
        if (type == 2) {
            double sx = 0.0, sy = 0.0;
            if (!sw_known) {
                sw = 0.0;
#pragma unroll
                for (int k = 0; k < PPT; ++k) sw += (double)sv[k];
                sw = chx_wave_sum_lane63(sw);
                sw_known = true;
            }
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                const double w = (double)sv[k];
                sx = fma(w, (double)x.get(k, 0), sx);
                sy = fma(w, (double)x.get(k, 2), sy);
            }
            sx = chx_wave_sum_lane63(sx);        // (the totals land in lane 63: DPP row broadcasts instead of ds_bpermute)
            sy = chx_wave_sum_lane63(sy);
            if ((threadIdx.x & 63) == 63) {
                double* part = bpm_ws + ((int64_t)items[i * 4 + 3] * nw_all + wslot) * 3;
                part[0] = sw;
                part[1] = sx;
                part[2] = sy;
            }
            continue;
        }
        if constexpr (SCREENS != 0) {
            if (type == 4) {
                const int64_t q = items[i * 4 + 2];
                const chx_lattice_screen& so = scr.s[(int)items[i * 4 + 3]];
                const T* __restrict__ charge = (const T*)scr.charge;
                T* rows = (T*)so.rows;
                if (rows) {
                    __syncthreads();                   // (the tile is free: its rows sit in registers)
#pragma unroll
                    for (int k = 0; k < PPT; ++k) {
                        const int p = threadIdx.x + k * CHX_BLOCK;
                        if (p < np) {
#pragma unroll
                            for (int j = 0; j < 7; ++j) lds[p * 7 + j] = x.get(k, j);
                        }
                    }
                    __syncthreads();
                    tile_store<T, TP>(rows + n0 * 7, lds, np * 7, chx_aligned16(rows) && row_vec, false);
                }
                T* sv_rec = (T*)so.survival;
                T* q_rec = (T*)so.charges;
#pragma unroll
                for (int k = 0; k < PPT; ++k) {
                    const int p = threadIdx.x + k * CHX_BLOCK;
                    if (p < np) {
                        if (sv_rec) sv_rec[n0 + p] = sv[k];
                        if (q_rec) q_rec[n0 + p] = charge ? charge[t0 + p] : (T)1;
                    }
                }
                if constexpr (SCREENS == 3) {
                    double* mp = (double*)so.mom_partials;
                    if (mp) {
                        // sums about the beam's FIRST row as it stands at this screen (carried through the items next to the lane's
                        // own rows: the same value in every workgroup, a particle of the beam — |x - c| stays beam-sized), added
                        // into kMomSlots sets with fp64 atomics (the preparation launch zeroed them)
                        __shared__ double mom_red[16 * 32];
                        __syncthreads();                 // (a second screen of the stretch: the array is free again)
                        double c0[6];
#pragma unroll
                        for (int j = 0; j < 6; ++j) c0[j] = isfinite((double)x0[j]) ? (double)x0[j] : 0.0;
                        // two halves of 16 accumulators (all 32 at once held the pass at three waves per SIMD): {W, W2, s[6], m_0*[6]},
                        // then {m_1*[5], m_2*[4], m_3*[3], m_4*[2], m_55}; mom_red column of value k: k (k < 14), k + 2 (k >= 14)
#pragma unroll
                        for (int half = 0; half < 2; ++half) {
                            __builtin_amdgcn_sched_barrier(0);      // (the halves' live ranges must not overlap: that is their point)
                            double acc[16];
#pragma unroll
                            for (int k2 = 0; k2 < 16; ++k2) acc[k2] = 0.0;
#pragma unroll
                            for (int k = 0; k < PPT; ++k) {
                                const int p = threadIdx.x + k * CHX_BLOCK;
                                if (p >= np) continue;
                                const double w = (double)sv[k];
                                double d[6];
#pragma unroll
                                for (int j = 0; j < 6; ++j) d[j] = (double)x.get(k, j) - c0[j];
                                if (half == 0) {
                                    acc[0] += w;
                                    acc[1] += w * w;
                                    const double wd0 = w * d[0];
#pragma unroll
                                    for (int a = 0; a < 6; ++a) {
                                        acc[2 + a] = fma(w, d[a], acc[2 + a]);
                                        acc[8 + a] = fma(wd0, d[a], acc[8 + a]);
                                    }
                                } else {
                                    int k2 = 0;
#pragma unroll
                                    for (int a = 1; a < 6; ++a) {
                                        const double wd = w * d[a];
#pragma unroll
                                        for (int b2 = a; b2 < 6; ++b2) { acc[k2] = fma(wd, d[b2], acc[k2]); ++k2; }
                                    }
                                }
                            }
                            chx_row16_sum16_folded(acc, mom_red + half * 16, 32);
                        }
                        __syncthreads();
                        const int nslots = (int)((gridDim.x < (unsigned)kMomSlots) ? gridDim.x : (unsigned)kMomSlots);
                        const int64_t stride = (int64_t)nslots * gridDim.y, slot = beam * nslots + (int)(blockIdx.x % (unsigned)nslots);
                        if (threadIdx.x < CHX_MOM_NOUT) {
                            const int col = threadIdx.x < 14 ? threadIdx.x : threadIdx.x + 2;
                            double t = 0.0;
#pragma unroll
                            for (int r2 = 0; r2 < 16; ++r2) t += mom_red[r2 * 32 + col];
                            unsafeAtomicAdd(&mp[(int64_t)threadIdx.x * stride + slot], t);
                        } else if (threadIdx.x < CHX_MOM_NOUT + 6 && blockIdx.x == 0) {
                            // the sums' centre, behind the 29 rows of sets: [29 * sets + j] of this beam's block
                            mp[(int64_t)CHX_MOM_NOUT * stride + beam * 6 + (threadIdx.x - CHX_MOM_NOUT)] = c0[threadIdx.x - CHX_MOM_NOUT];
                        }
                    }
                }
                if constexpr (SCREENS == 2) {
                  __shared__ long long comb_keys[kCombSlots];        // (one table for either kind of image)
                  __shared__ double comb_vals[kCombSlots];
                  if ((items[i * 4 + 1] & 1) && so.image) {
                    const T* mis = (const T*)ptrs[q];
                    const T* ps = (const T*)ptrs[q + 1];
                    const int bins_x = (int)ptrs[q + 4], bins_y = (int)ptrs[q + 5];
                    T lx, rx, ly, ry;
                    screen_extent_axis<T>((int)ptrs[q + 2], ps[0], lx, rx);
                    screen_extent_axis<T>((int)ptrs[q + 3], ps[1], ly, ry);
                    const T mx = mis[0], my = mis[1];
                    CombTable<T> table;
                    table.init(comb_keys, comb_vals, (T*)so.image + beam * ((int64_t)bins_x * bins_y));     // (beam b's image behind beam b - 1's)
#pragma unroll
                    for (int k = 0; k < PPT; ++k) {
                        const int p = threadIdx.x + k * CHX_BLOCK;
                        if (p >= np) continue;
                        const T vx = x.get(k, 0) - mx, vy = x.get(k, 2) - my;      // screen.py:200-212: positions relative to the screen
                        long long ix, iy;
                        T fx, fy, bw;
                        bool inside = cic_axis<T>(vx, lx, rx, bins_x, ix, fx, bw);
                        inside = cic_axis<T>(vy, ly, ry, bins_y, iy, fy, bw) && inside;
                        if (!inside) continue;
                        T c = charge ? charge[t0 + p] : (T)1;
                        c = fabs(c);
                        c = c * sv[k];
#pragma unroll
                        for (int oy = 0; oy < 2; ++oy) {
                            const long long jy = iy + oy;
                            if (jy < 0 || jy >= bins_y) continue;
                            const T wy = oy ? fy : ((T)1.0 - fy);
#pragma unroll
                            for (int ox = 0; ox < 2; ++ox) {
                                const long long jx = ix + ox;
                                if (jx < 0 || jx >= bins_x) continue;
                                const T wx = ox ? fx : ((T)1.0 - fx);
                                table.add((int64_t)jx + (int64_t)jy * bins_x, c * wx * wy);
                            }
                        }
                    }
                    table.flush();
                    __syncthreads();                   // (the table is free for the next screen of the stretch)
                  } else if ((items[i * 4 + 1] & 2) && so.image) {
                    // the 'histogram' image (screen.py:292-311: torch.histogramdd on the edges torch.linspace gave, weight |q| w):
                    // hist2d_kernel's arithmetic from registers — x - misalignment in T, ATen's bin search on the edge arrays the
                    // host formed (their addresses behind the screen's other entries in ptrs) — through the same combining table
                    const T* mis = (const T*)ptrs[q];
                    const int bins_x = (int)ptrs[q + 4], bins_y = (int)ptrs[q + 5];
                    const T* __restrict__ ex = (const T*)ptrs[q + 6];
                    const T* __restrict__ ey = (const T*)ptrs[q + 7];
                    const T mx = mis[0], my = mis[1];
                    CombTable<T> table;
                    table.init(comb_keys, comb_vals, (T*)so.image + beam * ((int64_t)bins_x * bins_y));     // (beam b's image behind beam b - 1's)
#pragma unroll
                    for (int k = 0; k < PPT; ++k) {
                        const int p = threadIdx.x + k * CHX_BLOCK;
                        if (p >= np) continue;
                        const T vx = x.get(k, 0) - mx, vy = x.get(k, 2) - my;
                        const int jx = hist_bin<T>(ex, bins_x, vx), jy = hist_bin<T>(ey, bins_y, vy);
                        if (jx < 0 || jy < 0) continue;
                        T c = charge ? fabs(charge[t0 + p]) : (T)1;
                        c = c * sv[k];
                        table.add((int64_t)jy * bins_x + jx, c);
                    }
                    table.flush();
                    __syncthreads();
                  }
                }
                continue;
            }
        }
        if (type == 3) {
            const int64_t q = items[i * 4 + 2];
            const T x_max = *(const T*)ptrs[q], y_max = *(const T*)ptrs[q + 1];
            const bool elliptical = items[i * 4 + 1] != 0;
            const T x_max2 = x_max * x_max, y_max2 = y_max * y_max;
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                const T px = x.get(k, 0), py = x.get(k, 2);
                bool inside;
                if (elliptical) {
                    const T a = (px * px) / x_max2;
                    const T c = (py * py) / y_max2;
                    inside = (a + c) <= (T)1;
                } else {
                    inside = (px > -x_max) && (px < x_max) && (py > -y_max) && (py < y_max);
                }
                sv[k] = sv[k] * (inside ? (T)1 : (T)0);
            }
            sw_known = false;
            continue;
        }
        const int64_t mrow = (int64_t)i * Bm + (Bm == 1 ? 0 : beam);       // this row's map of the item (vectorised settings)
        const T* __restrict__ R = STAGED ? maps_s + i * kApplyMapStride : reinterpret_cast<const T*>(Rs + mrow * 49);
        const bool cavity = type == 1;
        const double* __restrict__ c = STAGED ? coeffs_s + i * CHX_CAV_NCOEF : coeffs_g + mrow * CHX_CAV_NCOEF;
        if constexpr (SCREENS == 3) {
            T y0[7];
            apply7<T>(R, x0, y0);
            if (cavity) cavity_epilogue<T>(c, x0, y0);
#pragma unroll
            for (int j = 0; j < 7; ++j) x0[j] = y0[j];
        }
        if constexpr (std::is_same<T, float>::value && PPT % 2 == 0) {
            // the lane's particles two to a register pair, kept that way from load to store: every step of apply7's fmaf chain is ONE
            // v_pk_fma_f32 for both (same per-particle order -> same bits; at 4e8 particle rows the maps of a stretch are VALU time,
            // not HBM time)
#pragma unroll
            for (int pr = 0; pr < PPT / 2; ++pr) {
                chx_v2f y[7];
#pragma unroll
                for (int r = 0; r < 7; ++r) {
                    chx_v2f acc = x.v[pr][0] * R[r * 7];
#pragma unroll
                    for (int j = 1; j < 7; ++j) {
                        const chx_v2f m = {R[r * 7 + j], R[r * 7 + j]};
                        acc = __builtin_elementwise_fma(m, x.v[pr][j], acc);
                    }
                    y[r] = acc;
                }
                if (cavity) {
                    T xa[7], xb[7], ya[7], yb[7];
#pragma unroll
                    for (int j = 0; j < 7; ++j) {
                        xa[j] = x.v[pr][j].x; xb[j] = x.v[pr][j].y;
                        ya[j] = y[j].x; yb[j] = y[j].y;
                    }
                    cavity_epilogue<T>(c, xa, ya);
                    cavity_epilogue<T>(c, xb, yb);
                    y[4] = chx_v2f{ya[4], yb[4]};
                    y[5] = chx_v2f{ya[5], yb[5]};
                }
#pragma unroll
                for (int j = 0; j < 7; ++j) x.v[pr][j] = y[j];
            }
        } else {
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                T xi[7], y[7];
#pragma unroll
                for (int j = 0; j < 7; ++j) xi[j] = x.get(k, j);
                apply7<T>(R, xi, y);
                if (cavity) cavity_epilogue<T>(c, xi, y);
#pragma unroll
                for (int j = 0; j < 7; ++j) x.set(k, j, y[j]);
            }
        }
    }
    if (survival_out) {
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int p = threadIdx.x + k * CHX_BLOCK;
            if (p < np) survival_out[n0 + p] = sv[k];
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const int p = threadIdx.x + k * CHX_BLOCK;
        if (p < np) {
#pragma unroll
            for (int j = 0; j < 7; ++j) lds[p * 7 + j] = x.get(k, j);
        }
    }
    __syncthreads();
    tile_store<T, TP>(x_out + n0 * 7, lds, np * 7, out_vec_ok != 0 && row_vec, true);
}

// The same pass for a SCAN: B rows of lattice settings over ONE shared beam (x_in[N][7]; the maps of item i and row b at Rs[i * B + b]),
// no screens. Built like apply_shared_wave_kernel: a workgroup keeps its PPT * 256 particles in registers and walks a chunk of rows;
// every wave owns PPT * 64 consecutive particles, puts its outgoing rows into its own LDS slice and streams them out as 16-byte
// chunks — no workgroup barrier per row, the four waves drift apart and their stores overlap the others' maps and monitor sums.
// (lattice_apply_kernel takes one (tile, row) per workgroup: 8e5 short workgroups for 4096 x 1e5, each a load -> barrier -> maps ->
// barrier -> store chain; 2.69 ms with one monitor where the plain shared-beam apply takes 2.07.) Needs every row of the output to
// start on a 16-byte boundary. Per particle the arithmetic of lattice_apply_kernel, item by item; a monitor's per-wave sums cover
// other particles than there (a wave = PPT * 64 consecutive particles), the finalize kernel adds them up all the same.
// CAV false: the caller vouches that the stretch holds no cavity (chx_lattice_prepare_rows' small_runs) — without the fp64 cosine of the
// cavity epilogue the kernel keeps a map's 49 entries in scalar registers in fewer pieces.
template <typename T, int PPT, bool CAV>
__global__ __launch_bounds__(CHX_BLOCK) void lattice_scan_wave_kernel(const T* __restrict__ x_in, T* __restrict__ x_out,
                                                                     const int64_t* __restrict__ items, int n_items,
                                                                     const double* __restrict__ Rs, const double* __restrict__ coeffs,
                                                                     int64_t N, int64_t B, int64_t rows_per_chunk, int in_vec_ok,
                                                                     const T* __restrict__ survival, double* __restrict__ bpm_ws, int diag,
                                                                     const int64_t* __restrict__ ptrs, T* __restrict__ survival_out,
                                                                     int shared_sv, int transport /*monitors behind a linear prefix are
                                                                     evaluated elsewhere (lattice_scan_bpm_transport_kernel)*/) {
    using V = typename chx_vec16<T>::type;
    constexpr int VN = chx_vec16<T>::n;
    constexpr int TP = PPT * CHX_BLOCK;
    constexpr int WP = PPT * 64;                 // particles per wave
    constexpr int WE = WP * 7;                   // elements per wave
    constexpr int WV = WE / VN;                  // 16-byte chunks per wave
    constexpr bool kPairs = std::is_same<T, float>::value && PPT % 2 == 0;
    __shared__ __attribute__((aligned(16))) T lds[TP * 7];
    const int64_t t0 = (int64_t)blockIdx.x * TP;
    const int np = (int)((N - t0 < TP) ? (N - t0) : TP);
    const int b0 = (int)blockIdx.y * (int)rows_per_chunk;                  // (B <= 65535)
    const int b1 = (b0 + (int)rows_per_chunk < (int)B) ? b0 + (int)rows_per_chunk : (int)B;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;

    tile_load<T, TP>(x_in + t0 * 7, lds, np * 7, in_vec_ok != 0, false);      // (re-read by every chunk of rows: no streaming hint)
    __syncthreads();
    LaneRows<T, PPT> x0;
    T* wl = lds + wave * WE;                      // this wave's slice: particles t0 + wave * WP + [0, WP)
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const int p = wave * WP + k * 64 + lane;
#pragma unroll
        for (int j = 0; j < 7; ++j) x0.set(k, j, (p < np) ? wl[(k * 64 + lane) * 7 + j] : (T)0);
    }
    T sv0[PPT];
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const int p = wave * WP + k * 64 + lane;
        sv0[k] = (p < np) ? ((survival && shared_sv) ? survival[t0 + p] : (T)1) : (T)0;
    }
    // from here on a wave touches only its own slice
    const int valid = (np - wave * WP < 0) ? 0 : ((np - wave * WP < WP) ? (np - wave * WP) : WP);   // particles of this wave that exist
    const int vchunks = valid * 7 / VN;           // whole chunks inside the valid rows
    const int nw = (int)gridDim.x * (CHX_BLOCK / 64);                    // waves per row
    for (int b = b0; b < b1; ++b) {
        LaneRows<T, PPT> x = x0;
        T sv[PPT];
#pragma unroll
        for (int k = 0; k < PPT; ++k) sv[k] = sv0[k];
        if (diag && survival && !shared_sv) {
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                const int p = wave * WP + k * 64 + lane;
                sv[k] = (p < np) ? survival[(int64_t)b * N + t0 + p] : (T)0;
            }
        }

        bool sw_known = false;
        double sw = 0.0;
        bool linear = transport != 0;                // nothing but maps and monitors so far in this row's stretch
        for (int i = 0; i < n_items; ++i) {
            const int type = (int)items[i * 4];
            if (type == 1 || type == 3) linear = false;
            if (type == 2) {                         // active monitor (bpm.py:77-87): the wave's sums of w, w x, w y
                if (linear) continue;
                double sx = 0.0, sy = 0.0;
                if (!sw_known) {
                    sw = 0.0;
#pragma unroll
                    for (int k = 0; k < PPT; ++k) sw += (double)sv[k];
                    sw = chx_wave_sum_lane63(sw);
                    sw_known = true;
                }
#pragma unroll
                for (int k = 0; k < PPT; ++k) {
                    const double w = (double)sv[k];
                    sx = fma(w, (double)x.get(k, 0), sx);
                    sy = fma(w, (double)x.get(k, 2), sy);
                }
                sx = chx_wave_sum_lane63(sx);
                sy = chx_wave_sum_lane63(sy);
                if (lane == 63) {
                    double* part = bpm_ws + (((int64_t)items[i * 4 + 3] * B + b) * nw + (int64_t)blockIdx.x * (CHX_BLOCK / 64) + wave) * 3;
                    part[0] = sw;
                    part[1] = sx;
                    part[2] = sy;
                }
                continue;
            }
            if (type == 3) {                         // active aperture (aperture.py:90-135)
                const int64_t q = items[i * 4 + 2];
                const T x_max = *(const T*)ptrs[q], y_max = *(const T*)ptrs[q + 1];
                const bool elliptical = items[i * 4 + 1] != 0;
                const T x_max2 = x_max * x_max, y_max2 = y_max * y_max;
#pragma unroll
                for (int k = 0; k < PPT; ++k) {
                    const T px = x.get(k, 0), py = x.get(k, 2);
                    bool inside;
                    if (elliptical) {
                        const T a = (px * px) / x_max2;
                        const T c = (py * py) / y_max2;
                        inside = (a + c) <= (T)1;
                    } else {
                        inside = (px > -x_max) && (px < x_max) && (py > -y_max) && (py < y_max);
                    }
                    sv[k] = sv[k] * (inside ? (T)1 : (T)0);
                }
                sw_known = false;
                continue;
            }
            const int64_t mrow = (int64_t)i * B + b;
            const T* __restrict__ R = reinterpret_cast<const T*>(Rs + mrow * 49);
            const bool cavity = CAV && type == 1;
            const double* __restrict__ c = coeffs + mrow * CHX_CAV_NCOEF;
            if constexpr (kPairs) {
                // every step of apply7's fmaf chain is ONE v_pk_fma_f32 for the two particles of a pair (same per-particle order -> same
                // bits); matrix row by matrix row over ALL pairs, so that a map entry is fetched once per item (pair by pair the compiler,
                // short of scalar registers, fetched the 49 entries in four pieces per pair, each waited for on the spot)
                chx_v2f y[PPT / 2][7];
#pragma unroll
                for (int r = 0; r < 7; ++r) {
#pragma unroll
                    for (int pr = 0; pr < PPT / 2; ++pr) y[pr][r] = x.v[pr][0] * R[r * 7];
#pragma unroll
                    for (int j = 1; j < 7; ++j) {
                        const chx_v2f m = {R[r * 7 + j], R[r * 7 + j]};
#pragma unroll
                        for (int pr = 0; pr < PPT / 2; ++pr) y[pr][r] = __builtin_elementwise_fma(m, x.v[pr][j], y[pr][r]);
                    }
                }
                if (cavity) {
#pragma unroll
                    for (int pr = 0; pr < PPT / 2; ++pr) {
                        T xa[7], xb[7], ya[7], yb[7];
#pragma unroll
                        for (int j = 0; j < 7; ++j) {
                            xa[j] = x.v[pr][j].x; xb[j] = x.v[pr][j].y;
                            ya[j] = y[pr][j].x; yb[j] = y[pr][j].y;
                        }
                        cavity_epilogue<T>(c, xa, ya);
                        cavity_epilogue<T>(c, xb, yb);
                        y[pr][4] = chx_v2f{ya[4], yb[4]};
                        y[pr][5] = chx_v2f{ya[5], yb[5]};
                    }
                }
#pragma unroll
                for (int pr = 0; pr < PPT / 2; ++pr)
#pragma unroll
                    for (int j = 0; j < 7; ++j) x.v[pr][j] = y[pr][j];
            } else {
#pragma unroll
                for (int k = 0; k < PPT; ++k) {
                    T xi[7], y[7];
#pragma unroll
                    for (int j = 0; j < 7; ++j) xi[j] = x.get(k, j);
                    apply7<T>(R, xi, y);
                    if (cavity) cavity_epilogue<T>(c, xi, y);
#pragma unroll
                    for (int j = 0; j < 7; ++j) x.set(k, j, y[j]);
                }
            }
        }
        if (survival_out) {
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                const int p = wave * WP + k * 64 + lane;
                if (p < np) survival_out[(int64_t)b * N + t0 + p] = sv[k];
            }
        }
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
#pragma unroll
            for (int j = 0; j < 7; ++j) wl[(k * 64 + lane) * 7 + j] = x.get(k, j);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        T* __restrict__ gout = x_out + ((int64_t)b * N + t0 + wave * WP) * 7;
        V* __restrict__ gv = reinterpret_cast<V*>(gout);
        const V* lv = reinterpret_cast<const V*>(wl);
#pragma unroll
        for (int cidx = 0; cidx < (WV + 63) / 64; ++cidx) {
            const int v = cidx * 64 + lane;
            if (v < vchunks) chx_nt_store(lv[v], gv + v);
        }
        if (vchunks < WV) {                       // the last tile of a row: a few elements beyond the last whole chunk
            for (int e = vchunks * VN + lane; e < valid * 7; e += 64) gout[e] = wl[e];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// ---- monitors behind a LINEAR prefix of a scan: readings by transporting the beam's first moments ------------------------------------
// A monitor reads the weighted means of x and y AT its place (bpm.py:77-87). While everything in front of it in the stretch is linear
// (runs of maps, other monitors: no cavity's cosine, no aperture's mask), the weighted mean of the tracked coordinates IS the
// composed map applied to the weighted mean of the incoming beam (the 7th coordinate carries the affine part). For a scan of B rows
// of settings over ONE shared float32 beam that replaces three fp64 wave sums per monitor, wave and row in the particle pass by
// 8 sums over the shared beam, once, and one thread per row that takes the mean through the row's maps in fp64 — the maps the
// particles see, float32-rounded — and evaluates every such monitor on the way.
// What differs from summing the tracked float32 particles: their per-item rounding errors (0.5 ulp each, random in sign) are not in
// the transported mean — ~1e-10 of the beam size after averaging over 1e5 particles, three orders below the float32 reading's own
// resolution. float64 beams keep the particle sums (their readings are compared at 1e-15).
constexpr int kScanSumBlocks = 64;

// partials[kScanSumBlocks][8]: sum w, sum w x_j (j = 0..6) of the shared beam
template <typename T>
__global__ __launch_bounds__(CHX_BLOCK) void scan_beam_sums_kernel(const T* __restrict__ x, const T* __restrict__ survival, int64_t N,
                                                                  double* __restrict__ partials) {
    __shared__ double red[4 * 8];
    double a[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = 0.0;
    for (int64_t n = (int64_t)blockIdx.x * CHX_BLOCK + threadIdx.x; n < N; n += (int64_t)gridDim.x * CHX_BLOCK) {
        const double w = survival ? (double)survival[n] : 1.0;
        a[0] += w;
#pragma unroll
        for (int j = 0; j < 7; ++j) a[1 + j] = fma(w, (double)x[n * 7 + j], a[1 + j]);
    }
    chx_block_sum<8>(a, red);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) partials[blockIdx.x * 8 + k] = a[k];
    }
}

// one thread per row of settings: the beam's weighted mean through the row's maps item by item (fp64), the monitors evaluated on the
// way while the prefix is linear
template <typename T>
__global__ __launch_bounds__(CHX_BLOCK) void lattice_scan_bpm_transport_kernel(const int64_t* __restrict__ items, int n_items,
                                                                              const int64_t* __restrict__ ptrs, const double* __restrict__ Rs,
                                                                              const double* __restrict__ partials, int nblk, int64_t B,
                                                                              T* __restrict__ readings) {
    __shared__ double mean[8];
    if (threadIdx.x < 8) {
        double t = 0.0;
        for (int k = 0; k < nblk; ++k) t += partials[k * 8 + threadIdx.x];
        mean[threadIdx.x] = t;
    }
    __syncthreads();
    const int64_t b = (int64_t)blockIdx.x * CHX_BLOCK + threadIdx.x;
    if (b >= B) return;
    const double W = mean[0];
    double v[7];                                   // the weighted mean of the beam at the current place of the row's lattice
#pragma unroll
    for (int j = 0; j < 7; ++j) v[j] = mean[1 + j] / W;
    for (int i = 0; i < n_items; ++i) {
        const int type = (int)items[i * 4];
        if (type == 1 || type == 3) return;        // a cavity / an aperture: the monitors behind it take the particle sums
        if (type == 2) {
            const T* mis = (const T*)ptrs[items[i * 4 + 2]];
            T* r = readings + ((int64_t)items[i * 4 + 3] * B + b) * 2;
            r[0] = (T)v[0] - mis[0];
            r[1] = (T)v[2] - mis[1];
            continue;
        }
        if (type != 0) continue;
        const T* __restrict__ R = reinterpret_cast<const T*>(Rs + ((int64_t)i * B + b) * 49);
        double y[7];
#pragma unroll
        for (int r = 0; r < 7; ++r) {
            double acc = (double)R[r * 7] * v[0];
#pragma unroll
            for (int j = 1; j < 7; ++j) acc = fma((double)R[r * 7 + j], v[j], acc);
            y[r] = acc;
        }
#pragma unroll
        for (int j = 0; j < 7; ++j) v[j] = y[j];
    }
}

// one workgroup per monitor and beam: W and the two sums over the waves' shares (fixed order: thread t takes shares t, t + 256, ...),
// reading = (T)(sum / W) - misalignment, the subtraction in T like `incoming.mu_x - self.misalignment[..., 0]` (bpm.py:80-85)
template <typename T>
__global__ __launch_bounds__(CHX_BLOCK) void lattice_bpm_finalize_kernel(const int64_t* __restrict__ items, int n_items,
                                                                        const int64_t* __restrict__ ptrs, const double* __restrict__ ws,
                                                                        int64_t nw, T* __restrict__ readings, int transport = 0) {
    __shared__ double red[4 * 3];
    const int slot = blockIdx.x;
    if (transport) {                                       // a monitor behind a linear prefix has its reading already
        bool linear = true;
        for (int i = 0; i < n_items; ++i) {
            const int type = (int)items[i * 4];
            if (type == 1 || type == 3) linear = false;
            if (type == 2 && (int)items[i * 4 + 3] == slot) break;
        }
        if (linear) return;
    }
    const int64_t beam = blockIdx.y;                       // nw = waves per beam; readings[slot][beam][2]
    const double* part = ws + ((int64_t)slot * gridDim.y + beam) * nw * 3;
    double v[3] = {0.0, 0.0, 0.0};
    for (int64_t i = threadIdx.x; i < nw; i += CHX_BLOCK) {
        v[0] += part[i * 3];
        v[1] += part[i * 3 + 1];
        v[2] += part[i * 3 + 2];
    }
    chx_block_sum<3>(v, red);
    if (threadIdx.x == 0) {
        const T* mis = nullptr;
        for (int i = 0; i < n_items; ++i)
            if (items[i * 4] == 2 && (int)items[i * 4 + 3] == slot) mis = (const T*)ptrs[items[i * 4 + 2]];
        T* r = readings + ((int64_t)slot * gridDim.y + beam) * 2;
        r[0] = (T)(v[1] / v[0]) - mis[0];
        r[1] = (T)(v[2] / v[0]) - mis[1];
    }
}
}  // namespace

extern "C" size_t chx_lattice_diag_workspace_bytes(int64_t N, int64_t B, int64_t n_bpm) {
    if (N < 1 || B < 1 || n_bpm < 0) return 0;
    const int64_t nw = ((N + CHX_BLOCK - 1) / CHX_BLOCK) * (CHX_BLOCK / 64);
    return (size_t)(nw * B * 3 * n_bpm) * sizeof(double);
}

extern "C" int chx_lattice_track_diag(const int64_t* table, int64_t n_items, int64_t n_elems, int64_t n_ptrs, const void* energy,
                                      double mass_eV, double n_charges, int dtype, void* state, size_t state_bytes, const void* x_in,
                                      void* x_out, int64_t N, int64_t B, int64_t Bx, int64_t Bm, int64_t Bw, int small_runs,
                                      void* energy_out, const void* s_in, void* s_out, const void* survival, void* survival_out,
                                      int64_t n_bpm, void* readings, void* workspace, size_t workspace_bytes, void* stream) {
    return chx_lattice_track_screens(table, n_items, n_elems, n_ptrs, energy, mass_eV, n_charges, dtype, state, state_bytes, x_in, x_out,
                                     N, B, Bx, Bm, Bw, small_runs, energy_out, s_in, s_out, survival, survival_out, n_bpm, readings,
                                     workspace, workspace_bytes, nullptr, nullptr, 0, stream);
}

// the particle pass's grid.x: one particle per lane on small beams, two from 1e6 particle rows on (see chx_lattice_track_screens)
static inline int lattice_ppt(int64_t N, int64_t B) { return (N * B >= 1000000) ? 2 : 1; }

extern "C" int64_t chx_lattice_moment_blocks(int64_t N, int64_t B) {
    if (N < 1 || B < 1) return 0;
    const int64_t tile = (int64_t)CHX_BLOCK * lattice_ppt(N, B);
    const int64_t wgs = (N + tile - 1) / tile;
    return (wgs < kMomSlots ? wgs : kMomSlots) * B;
}

// (chx_moments.hip) partials[29][n_sets] + centre[6] -> out[29] (+ one entry): the second launch of chx_moments_entry on its own
int chx_moments_finalize_sets(const double* partials, int64_t n_sets, const double* centre, int dtype, double* out, int index, int take_sqrt,
                              void* entry_out, void* stream);

extern "C" int chx_lattice_screen_moments(const double* mom_partials, int64_t n_blocks, int dtype, double* out, int index, int take_sqrt,
                                          void* entry_out, void* stream) {
    if (!mom_partials || !out || n_blocks < 1 || n_blocks > 0x7fffffffLL || index >= CHX_MOM_NOUT || (index >= 0 && !entry_out))
        return CHX_ERR_INVALID_ARG;
    if (dtype != CHX_F32 && dtype != CHX_F64) return CHX_ERR_DTYPE;
    if (n_blocks > kMomSlots) return CHX_ERR_INVALID_ARG;
    return chx_moments_finalize_sets(mom_partials, n_blocks, mom_partials + CHX_MOM_NOUT * n_blocks, dtype, out, index, take_sqrt, entry_out, stream);
}

extern "C" int chx_lattice_track_screens(const int64_t* table, int64_t n_items, int64_t n_elems, int64_t n_ptrs, const void* energy,
                                         double mass_eV, double n_charges, int dtype, void* state, size_t state_bytes, const void* x_in,
                                         void* x_out, int64_t N, int64_t B, int64_t Bx, int64_t Bm, int64_t Bw, int small_runs,
                                         void* energy_out, const void* s_in, void* s_out, const void* survival, void* survival_out,
                                         int64_t n_bpm, void* readings, void* workspace, size_t workspace_bytes, const void* charge,
                                         const chx_lattice_screen* screens, int64_t n_screens, void* stream) {
    if (!x_in || !x_out || N < 1 || B < 1 || B > 65535 || n_bpm < 0 || n_bpm > n_items) return CHX_ERR_INVALID_ARG;
    if (!chx_bcast_ok(Bx, B) || !chx_bcast_ok(Bm, B) || !chx_bcast_ok(Bw, B)) return CHX_ERR_INVALID_ARG;
    if (n_bpm > 0 && (!readings || !workspace)) return CHX_ERR_INVALID_ARG;
    if (n_bpm > 0 && workspace_bytes < chx_lattice_diag_workspace_bytes(N, B, n_bpm)) return CHX_ERR_WORKSPACE;
    // screens of a VECTORISED beam (B > 1 beams of N particles under ONE lattice setting, Bm = 1): every record holds the B beams one
    // behind the other ([B][N][7] rows, [B][N] charges and survival probabilities), every image B images; `charge` is ONE row of N
    // charges shared by the beams (particle_beam.py: a vectorised beam's charges broadcast)
    if (n_screens < 0 || n_screens > CHX_LATTICE_MAX_SCREENS || (n_screens > 0 && (!screens || (B != 1 && Bm != 1)))) return CHX_ERR_INVALID_ARG;
    for (int64_t k = 0; k < n_screens; ++k)
        if (B != 1 && screens[k].mom_partials) return CHX_ERR_INVALID_ARG;
    int st = chx_lattice_prepare_screens(table, n_items, n_elems, n_ptrs, Bm, small_runs, energy, mass_eV, n_charges, dtype, state,
                                         state_bytes, energy_out, s_in, s_out, screens, n_screens, stream);
    if (st != CHX_OK) return st;
    ApplyScreens scr;
    for (int k = 0; k < CHX_LATTICE_MAX_SCREENS; ++k) scr.s[k] = k < n_screens ? screens[k] : chx_lattice_screen{};
    scr.charge = charge;
    const double* Rs = (const double*)state;
    const double* coeffs = Rs + n_items * Bm * 49;
    const int shared_in = (Bx == 1 && B > 1) ? 1 : 0, shared_sv = (Bw == 1 && B > 1) ? 1 : 0;
    hipStream_t s = (hipStream_t)stream;
    const int iv = chx_aligned16(x_in) ? 1 : 0, ov = chx_aligned16(x_out) ? 1 : 0;
    // one particle per lane on small beams (a few dozen tiles: two per lane only halve the waves in flight); two per lane from 1e6
    // particle rows on (half the waves: half the wave sums at the monitors, half the scalar loads of the maps; measured: a
    // 16-cavity linac 0.132 -> 0.124 ms at 1e6, 0.418 -> 0.371 at 4e6, no gain at 3e5)
    const int ppt = lattice_ppt(N, B);
    const int64_t tile = (int64_t)CHX_BLOCK * ppt;
    const dim3 grid((unsigned)((N + tile - 1) / tile), (unsigned)B);
    const int64_t nw = (int64_t)grid.x * (CHX_BLOCK / 64);
    const int64_t* ptrs = table + n_items * 4 + 2 * n_elems;
    const int diag = (n_bpm > 0 || survival_out || n_screens > 0) ? 1 : 0;
    // a scan of lattice settings over one shared beam: workgroups that keep their particles and walk a chunk of rows, wave-private
    // output staging (lattice_scan_wave_kernel) — when every row of the output starts on a 16-byte boundary
    // From 8e6 particle rows on (benchmarks/scan_wave_crossover.py: 64 x 1e5 a tie, 1024 x 1e4 0.113 -> 0.100 ms, 64 x 1e6 0.64 -> 0.36;
    // below, its 1024-particle workgroups are too few for a lattice with many items: a 16-cavity linac at 64 energies x 1e4 particles
    // 154 -> 172 us). CHX_TUNE_SCAN_WAVE=0 / 2 (tests, benchmarks): never / whenever the layout allows.
    const char* scan_env = getenv("CHX_TUNE_SCAN_WAVE");
    const int scan_wave = scan_env ? atoi(scan_env) : 1;
    const size_t esz = dtype == CHX_F32 ? 4 : 8;
    if (scan_wave != 0 && (scan_wave == 2 || B * N >= 8000000) && shared_in && Bm == B && B >= 8 && n_screens == 0 &&
        chx_aligned16(x_out) && (N * 7 * (int64_t)esz) % 16 == 0 && (dtype == CHX_F32 || dtype == CHX_F64)) {
        const int wppt = 16 / (int)esz;
        const int64_t wtp = (int64_t)wppt * CHX_BLOCK;
        const int64_t wtiles = (N + wtp - 1) / wtp;
        static const int64_t scan_wgs = [] { const char* e = getenv("CHX_TUNE_SCAN_WGS"); const int v = e ? atoi(e) : 0; return (int64_t)(v > 0 ? v : 8192); }();
        int64_t wchunks = (scan_wgs + wtiles - 1) / wtiles;
        if (wchunks > B) wchunks = B;
        int64_t wrows = (B + wchunks - 1) / wchunks;
        static const int64_t scan_min_rows = [] { const char* e = getenv("CHX_TUNE_SCAN_MIN_ROWS"); const int v = e ? atoi(e) : 0; return (int64_t)(v > 0 ? v : 1); }();
        if (wrows < scan_min_rows) wrows = scan_min_rows;
        wchunks = (B + wrows - 1) / wrows;
        if (wtiles > 0x7fffffffLL || wchunks > 65535) return CHX_ERR_INVALID_ARG;
        const dim3 wgrid((unsigned)wtiles, (unsigned)wchunks);
        // monitors behind a linear prefix by moment transport (float32, one shared row of weights): 8 sums over the shared beam in the
        // part of the workspace the per-wave sums of this grid do not reach. CHX_TUNE_SCAN_TRANSPORT=0: particle sums everywhere.
        const char* transport_env = getenv("CHX_TUNE_SCAN_TRANSPORT");
        const bool transport_on = !(transport_env && transport_env[0] == '0');
        const int64_t ws_used = wtiles * (CHX_BLOCK / 64) * B * 3 * n_bpm;
        const int transport = (transport_on && dtype == CHX_F32 && n_bpm > 0 && (!survival || shared_sv) &&
                               workspace_bytes >= (size_t)(ws_used + kScanSumBlocks * 8) * sizeof(double)) ? 1 : 0;
        double* sum_partials = (double*)workspace + ws_used;
        if (transport) {
            hipLaunchKernelGGL(scan_beam_sums_kernel<float>, dim3(kScanSumBlocks), dim3(CHX_BLOCK), 0, s, (const float*)x_in,
                               (const float*)survival, N, sum_partials);
            CHX_CHECK_LAUNCH();
        }
#define CHX_SCAN_LAUNCH(T, PPT, ...)                                                                        \
    do {                                                                                                    \
        if (small_runs & 1) hipLaunchKernelGGL((lattice_scan_wave_kernel<T, PPT, false>), wgrid, __VA_ARGS__); \
        else hipLaunchKernelGGL((lattice_scan_wave_kernel<T, PPT, true>), wgrid, __VA_ARGS__);              \
    } while (0)
        const int64_t wnw = wtiles * (CHX_BLOCK / 64);
        if (dtype == CHX_F32) {
            CHX_SCAN_LAUNCH(float, 4, dim3(CHX_BLOCK), 0, s, (const float*)x_in, (float*)x_out, table,
                               (int)n_items, Rs, coeffs, N, B, wrows, iv, (const float*)survival, (double*)workspace, diag, ptrs,
                               (float*)survival_out, shared_sv, transport);
            CHX_CHECK_LAUNCH();
            if (n_bpm > 0)
                hipLaunchKernelGGL(lattice_bpm_finalize_kernel<float>, dim3((unsigned)n_bpm, (unsigned)B), dim3(CHX_BLOCK), 0, s, table,
                                   (int)n_items, ptrs, (const double*)workspace, wnw, (float*)readings, transport);
            if (transport) {
                CHX_CHECK_LAUNCH();
                hipLaunchKernelGGL(lattice_scan_bpm_transport_kernel<float>, dim3((unsigned)((B + CHX_BLOCK - 1) / CHX_BLOCK)), dim3(CHX_BLOCK), 0,
                                   s, table, (int)n_items, ptrs, Rs, sum_partials, kScanSumBlocks, B, (float*)readings);
            }
        } else {
            CHX_SCAN_LAUNCH(double, 2, dim3(CHX_BLOCK), 0, s, (const double*)x_in, (double*)x_out,
                               table, (int)n_items, Rs, coeffs, N, B, wrows, iv, (const double*)survival, (double*)workspace, diag, ptrs,
                               (double*)survival_out, shared_sv, 0);
            CHX_CHECK_LAUNCH();
            if (n_bpm > 0)
                hipLaunchKernelGGL(lattice_bpm_finalize_kernel<double>, dim3((unsigned)n_bpm, (unsigned)B), dim3(CHX_BLOCK), 0, s, table,
                                   (int)n_items, ptrs, (const double*)workspace, wnw, (double*)readings);
        }
        CHX_CHECK_LAUNCH();
#undef CHX_SCAN_LAUNCH
        return CHX_OK;
    }
    // small beams walk their items out of LDS (CHX_TUNE_APPLY_STAGED=0: every item's type and map read where they are used)
    static const bool stage_on = [] { const char* e = getenv("CHX_TUNE_APPLY_STAGED"); return !(e && e[0] == '0'); }();
    // (at most 16 KB next to the particle tile and, with screens, the 32 KB combining table: inside the default 64 KB of a workgroup)
    const size_t stage_need = (size_t)n_items * (kApplyMapStride * (dtype == CHX_F32 ? 4 : 8) + CHX_CAV_NCOEF * 8 + 4 * 8);
    // Taken where it was measured to pay (kernel durations, rocprofv3, 1e4 particles): float64 maps — C1's pass with its screen 14.8 -> 8.3 us
    // — and stretches with cavities — a 16-cell float32 linac 25.0 -> 19.0 us; a float32 stretch without cavities is level (100 elements,
    // 25 monitors, 1e5 particles: 25.8 / 26.5 us) or loses (the control step's four items: 7.8 -> 9.7 us) and keeps its scalar loads.
    const bool staged = stage_on && ppt == 1 && n_items <= kApplyStagedItems && stage_need <= 16384 &&
                        (dtype == CHX_F64 || !(small_runs & 1));
    const size_t stage_bytes = staged ? stage_need : 0;
#define CHX_LATTICE_APPLY_S(T, PPT, SCR)                                                                                            \
    do {                                                                                                                            \
        if (PPT == 1 && staged)                                                                                                     \
            hipLaunchKernelGGL((lattice_apply_kernel<T, 1, SCR, true>), grid, dim3(CHX_BLOCK), stage_bytes, s, (const T*)x_in,      \
                               (T*)x_out, table, (int)n_items, Rs, coeffs, N, iv, ov, (const T*)survival, (double*)workspace, diag, \
                               ptrs, (T*)survival_out, shared_in, Bm, shared_sv, scr);                                              \
        else                                                                                                                        \
            hipLaunchKernelGGL((lattice_apply_kernel<T, PPT, SCR, false>), grid, dim3(CHX_BLOCK), 0, s, (const T*)x_in, (T*)x_out,  \
                               table, (int)n_items, Rs, coeffs, N, iv, ov, (const T*)survival, (double*)workspace, diag, ptrs,      \
                               (T*)survival_out, shared_in, Bm, shared_sv, scr);                                                    \
    } while (0)
    bool images = false, sums = false;
    for (int64_t k = 0; k < n_screens; ++k) {
        images = images || screens[k].image != nullptr;
        sums = sums || screens[k].mom_partials != nullptr;
    }
    if (images && sums) return CHX_ERR_INVALID_ARG;     // (one or the other per call: the combining table and the sums' LDS)
#define CHX_LATTICE_APPLY(T, PPT)                  \
    do {                                           \
        if (n_screens > 0 && images) CHX_LATTICE_APPLY_S(T, PPT, 2); \
        else if (n_screens > 0 && sums) CHX_LATTICE_APPLY_S(T, PPT, 3); \
        else if (n_screens > 0) CHX_LATTICE_APPLY_S(T, PPT, 1);      \
        else CHX_LATTICE_APPLY_S(T, PPT, 0);       \
    } while (0)
    if (dtype == CHX_F32) {
        if (ppt == 2) CHX_LATTICE_APPLY(float, 2);
        else CHX_LATTICE_APPLY(float, 1);
        CHX_CHECK_LAUNCH();
        if (n_bpm > 0)
            hipLaunchKernelGGL(lattice_bpm_finalize_kernel<float>, dim3((unsigned)n_bpm, (unsigned)B), dim3(CHX_BLOCK), 0, s, table, (int)n_items, ptrs,
                               (const double*)workspace, nw, (float*)readings);
    } else {
        if (ppt == 2) CHX_LATTICE_APPLY(double, 2);
        else CHX_LATTICE_APPLY(double, 1);
        CHX_CHECK_LAUNCH();
        if (n_bpm > 0)
            hipLaunchKernelGGL(lattice_bpm_finalize_kernel<double>, dim3((unsigned)n_bpm, (unsigned)B), dim3(CHX_BLOCK), 0, s, table, (int)n_items, ptrs,
                               (const double*)workspace, nw, (double*)readings);
    }
#undef CHX_LATTICE_APPLY
#undef CHX_LATTICE_APPLY_S
    CHX_CHECK_LAUNCH();
    return CHX_OK;
}

extern "C" int chx_lattice_track(const int64_t* table, int64_t n_items, int64_t n_elems, int64_t n_ptrs, const void* energy,
                                 double mass_eV, double n_charges, int dtype, void* state, size_t state_bytes, const void* x_in,
                                 void* x_out, int64_t N, void* energy_out, const void* s_in, void* s_out, void* stream) {
    return chx_lattice_track_diag(table, n_items, n_elems, n_ptrs, energy, mass_eV, n_charges, dtype, state, state_bytes, x_in, x_out, N,
                                  1, 1, 1, 1, 0, energy_out, s_in, s_out, nullptr, nullptr, 0, nullptr, nullptr, 0, stream);
}

// ---- several device arrays copied by ONE launch -----------------------------------------------------------------------
// A Screen records a copy of the incoming beam (screen.py:190 `incoming.clone()`): five tensors, two of them scalars — five
// launches of a framework copy kernel, or one of this. Arrays travel by value in the kernel arguments; blockIdx.y picks the
// array, 16-byte chunks when source and destination allow, bytes otherwise.
namespace {
constexpr int kCopyMax = 8;
struct CopyArgs {
    const void* src[kCopyMax];
    void* dst[kCopyMax];
    int64_t bytes[kCopyMax];
};

__global__ __launch_bounds__(CHX_BLOCK) void copy_arrays_kernel(CopyArgs a) {
    const int k = blockIdx.y;
    const char* __restrict__ s = (const char*)a.src[k];
    char* __restrict__ d = (char*)a.dst[k];
    const int64_t n = a.bytes[k];
    const int64_t tid = (int64_t)blockIdx.x * CHX_BLOCK + threadIdx.x, nthreads = (int64_t)gridDim.x * CHX_BLOCK;
    int64_t done = 0;
    if ((((uintptr_t)s | (uintptr_t)d) & 15) == 0) {
        const int64_t n16 = n >> 4;
        const float4* __restrict__ s4 = (const float4*)s;
        float4* __restrict__ d4 = (float4*)d;
        // four independent 16-byte loads per lane in flight before the first store, streaming (nontemporal) both ways: the
        // copy of a 28 MB particle array went from 4.0 to the apply kernels' rate
        int64_t i = tid;
        const chx_v4f* __restrict__ sv = reinterpret_cast<const chx_v4f*>(s);
        chx_v4f* __restrict__ dv = reinterpret_cast<chx_v4f*>(d);
        for (; i + 3 * nthreads < n16; i += 4 * nthreads) {
            const chx_v4f v0 = __builtin_nontemporal_load(sv + i), v1 = __builtin_nontemporal_load(sv + i + nthreads),
                          v2 = __builtin_nontemporal_load(sv + i + 2 * nthreads), v3 = __builtin_nontemporal_load(sv + i + 3 * nthreads);
            __builtin_nontemporal_store(v0, dv + i);
            __builtin_nontemporal_store(v1, dv + i + nthreads);
            __builtin_nontemporal_store(v2, dv + i + 2 * nthreads);
            __builtin_nontemporal_store(v3, dv + i + 3 * nthreads);
        }
        for (; i < n16; i += nthreads) d4[i] = s4[i];
        done = n16 << 4;
    }
    for (int64_t i = done + tid; i < n; i += nthreads) d[i] = s[i];
}
}  // namespace

extern "C" int chx_copy_arrays(const void* const* src, void* const* dst, const int64_t* bytes, int32_t n, void* stream) {
    if (!src || !dst || !bytes || n < 1 || n > kCopyMax) return CHX_ERR_INVALID_ARG;
    CopyArgs a;
    int64_t most = 0;
    for (int k = 0; k < kCopyMax; ++k) {
        a.src[k] = k < n ? src[k] : nullptr;
        a.dst[k] = k < n ? dst[k] : nullptr;
        a.bytes[k] = k < n ? bytes[k] : 0;
        if (k < n) {
            if (bytes[k] < 0 || (bytes[k] > 0 && (!src[k] || !dst[k]))) return CHX_ERR_INVALID_ARG;
            most = bytes[k] > most ? bytes[k] : most;
        }
    }
    if (most == 0) return CHX_OK;
    // 64 bytes per lane and pass: a 28 MB particle array gets 1709 workgroups, a scalar one
    int64_t blocks = (most + CHX_BLOCK * 64 - 1) / (CHX_BLOCK * 64);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(copy_arrays_kernel, dim3((unsigned)blocks, (unsigned)n), dim3(CHX_BLOCK), 0, (hipStream_t)stream, a);
    CHX_CHECK_LAUNCH();
    return CHX_OK;
}

// ---- the extent a stretch derives from a Screen's pixel size (screen.py:139-148), for checking it against the tensor expression ----
namespace {
template <typename T>
__global__ void screen_extent_kernel(const T* __restrict__ ps, int rx, int ry, T* __restrict__ out) {
    if (threadIdx.x == 0) {
        screen_extent_axis<T>(rx, ps[0], out[0], out[1]);
        screen_extent_axis<T>(ry, ps[1], out[2], out[3]);
    }
}
}  // namespace

extern "C" int chx_screen_extent(const void* pixel_size, int32_t resolution_x, int32_t resolution_y, int dtype, void* out, void* stream) {
    if (!pixel_size || !out) return CHX_ERR_INVALID_ARG;
    if (dtype == CHX_F32)
        hipLaunchKernelGGL(screen_extent_kernel<float>, dim3(1), dim3(64), 0, (hipStream_t)stream, (const float*)pixel_size, resolution_x,
                           resolution_y, (float*)out);
    else if (dtype == CHX_F64)
        hipLaunchKernelGGL(screen_extent_kernel<double>, dim3(1), dim3(64), 0, (hipStream_t)stream, (const double*)pixel_size, resolution_x,
                           resolution_y, (double*)out);
    else
        return CHX_ERR_DTYPE;
    CHX_CHECK_LAUNCH();
    return CHX_OK;
}

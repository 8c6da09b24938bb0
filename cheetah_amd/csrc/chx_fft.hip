// chx_fft.hip — Poisson solve of SpaceChargeKick by pruned, symmetry-aware FFTs for gfx950
// (space_charge_kick.py:293-322: phi = irfftn(rfftn(rho_padded) * rfftn(G)) cropped to the first octant).
//
// The reference (and chx_sc_fft_exec) transforms three dense (2g)^3 arrays. Here the structure of the Hockney
// convolution is used instead, with one batched line-FFT kernel that stages tiles of lines through LDS:
//   * rho is non-zero only in [0,g)^3 of the doubled array: the forward transform runs z lines on the g x g
//     non-zero columns, then y lines on g x (g+1), then x lines on 2g x (g+1); the upper half of every input line
//     is an implicit zero (never read). 177 MB of traffic per transform at g = 128 instead of ~400 MB.
//   * only phi in [0,g)^3 is needed: the inverse runs the same three passes backwards and never writes the rest.
//   * the integrated Green function is real and even in every axis, so its spectrum is real and even:
//     it is computed from the compact (g+1)^3 table by three passes of even-extension transforms and stored as
//     (g+1)^3 reals (8.6 MB instead of a 67 MB complex array); the spectral multiply looks it up by |k|.
// Line lengths n = 2g must be powers of two (64 ... 1024); other grids use the hipFFT path (chx_sc_fft_exec).
#include "chx_common.h"
#include "chx_fft_reg.h"
#include "chx_sc_math.h"
#include "chx_sc_tiles.h"
#include "chx_sc_geom_dev.h"

namespace {

constexpr int kTL = 16;        // lines per tile
constexpr int kHalo = 2;       // chx_sc_convolve_halo: nodes of halo around phi (chx_spacecharge.hip reads i-1 .. i+2 of cell i >= -1)
constexpr int kPad = kTL + 1;  // LDS row pitch (complex elements): breaks the power-of-two bank pattern

// LOAD_EVEN_REAL_PAIR / STORE_REAL_PAIR (register kernels only): TWO real, even lines ride one complex transform as a + i b —
// the spectrum of a real even sequence is real, so Re and Im of the result are the two spectra, exactly; a tile is 32 lines
enum LoadMode { LOAD_COMPLEX = 0, LOAD_REAL = 1, LOAD_HERMITIAN = 2, LOAD_EVEN_REAL = 3, LOAD_EVEN_REAL_PAIR = 4 };
enum StoreMode { STORE_COMPLEX = 0, STORE_REAL = 1, STORE_REAL_PAIR = 2 };

struct LineLayout {
    int64_t point_stride;   // elements between consecutive points of a line
    int64_t inner_stride;   // elements between consecutive lines of the inner index
    int64_t outer_stride;   // elements between consecutive outer indices
    int64_t batch_stride;   // elements between batch rows
};

template <typename T> struct alignas(2 * sizeof(T)) cplx { T re, im; };   // one 8/16-byte access per complex number

template <typename T>
__device__ __forceinline__ void sincos_2pi(int k, int n, T& s, T& c);
template <>
__device__ __forceinline__ void sincos_2pi<float>(int k, int n, float& s, float& c) {
    sincospif(2.0f * (float)k / (float)n, &s, &c);
}
template <>
__device__ __forceinline__ void sincos_2pi<double>(int k, int n, double& s, double& c) {
    sincospi(2.0 * (double)k / (double)n, &s, &c);
}

// One workgroup transforms kTL lines of length n (power of two) that are adjacent in the line index.
//   lines: l = outer * inner_count + inner, l in [0, L)
//   n_valid: leading input points that exist (the rest are zeros that are not read)
//   n_keep:  leading output points that are written
//   POINT_FAST: consecutive lanes walk along a line (point_stride == 1 passes), else across lines
template <typename T, int LOADM, int STOREM, bool POINT_FAST>
__global__ __launch_bounds__(CHX_BLOCK) void fft_lines_kernel(const T* __restrict__ in, T* __restrict__ out, int n, int log2n,
                                                             int n_valid, int n_keep, int64_t L, int64_t inner_count,
                                                             LineLayout li, LineLayout lo, int inverse) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cplx<T>* buf = reinterpret_cast<cplx<T>*>(smem_raw);          // [n][kPad]
    cplx<T>* tw = buf + (size_t)n * kPad;                         // [n/2] twiddles exp(-+2 pi i k / n)

    const int tid = threadIdx.x;
    const int64_t l0 = (int64_t)blockIdx.x * kTL;
    const int nl = (int)((L - l0 < kTL) ? (L - l0) : kTL);
    const int64_t b = blockIdx.y;
    // strides are in elements of the array's own type (real or complex): complex arrays are addressed as T pairs
    const T* inb = in + b * li.batch_stride * ((LOADM == LOAD_COMPLEX || LOADM == LOAD_HERMITIAN) ? 2 : 1);
    T* outb = out + b * lo.batch_stride * ((STOREM == STORE_COMPLEX) ? 2 : 1);

    __shared__ int64_t in_base[kTL], out_base[kTL];
    if (tid < kTL) {
        const int64_t l = l0 + (tid < nl ? tid : 0);
        const int64_t outer = l / inner_count, inner = l - outer * inner_count;
        in_base[tid] = outer * li.outer_stride + inner * li.inner_stride;
        out_base[tid] = outer * lo.outer_stride + inner * lo.inner_stride;
    }
    for (int k = tid; k < n / 2; k += CHX_BLOCK) {
        T s, c;
        sincos_2pi<T>(k, n, s, c);
        tw[k].re = c;
        tw[k].im = inverse ? s : -s;
    }
    __syncthreads();
    // ---- load (bit-reversed point index) --------------------------------------------------------------------
    const int total = n * kTL;
    for (int e = tid; e < total; e += CHX_BLOCK) {
        int line, p;
        if (POINT_FAST) { p = e & (n - 1); line = e >> log2n; } else { line = e & (kTL - 1); p = e / kTL; }
        cplx<T> v;
        v.re = (T)0;
        v.im = (T)0;
        if (line < nl) {
            const int64_t base = in_base[line];
            if (LOADM == LOAD_COMPLEX) {
                if (p < n_valid) {
                    const T* q = inb + 2 * (base + (int64_t)p * li.point_stride);
                    v.re = q[0];
                    v.im = q[1];
                }
            } else if (LOADM == LOAD_REAL) {
                if (p < n_valid) v.re = inb[base + (int64_t)p * li.point_stride];
            } else if (LOADM == LOAD_HERMITIAN) {  // half spectrum 0..n/2 given: X[n-k] = conj X[k]
                const int ps = (p <= n / 2) ? p : n - p;
                const T* q = inb + 2 * (base + (int64_t)ps * li.point_stride);
                v.re = q[0];
                v.im = (p <= n / 2) ? q[1] : -q[1];
            } else {  // LOAD_EVEN_REAL: x[n-p] = x[p], values 0..n/2 given
                const int ps = (p <= n / 2) ? p : n - p;
                v.re = inb[base + (int64_t)ps * li.point_stride];
            }
        }
        const int pr = (int)(__brev((unsigned)p) >> (32 - log2n));
        buf[pr * kPad + line] = v;
    }
    __syncthreads();
    // ---- radix-2 decimation-in-time stages, in place ----------------------------------------------------------
    const int half_total = (n / 2) * kTL;
    for (int s = 0; s < log2n; ++s) {
        const int half = 1 << s;
        for (int e = tid; e < half_total; e += CHX_BLOCK) {
            const int line = e & (kTL - 1), bf = e / kTL;  // butterfly index 0 .. n/2-1
            const int j = bf & (half - 1);
            const int i0 = ((bf >> s) << (s + 1)) + j;
            const int i1 = i0 + half;
            const cplx<T> w = tw[j << (log2n - 1 - s)];
            const cplx<T> a = buf[i0 * kPad + line];
            const cplx<T> c = buf[i1 * kPad + line];
            cplx<T> t;
            t.re = c.re * w.re - c.im * w.im;
            t.im = c.re * w.im + c.im * w.re;
            cplx<T> u, d;
            u.re = a.re + t.re; u.im = a.im + t.im;
            d.re = a.re - t.re; d.im = a.im - t.im;
            buf[i0 * kPad + line] = u;
            buf[i1 * kPad + line] = d;
        }
        __syncthreads();
    }
    // ---- store ------------------------------------------------------------------------------------------------
    const int total_out = n_keep * kTL;
    for (int e = tid; e < total_out; e += CHX_BLOCK) {
        int line, p;
        if (POINT_FAST) { p = e % n_keep; line = e / n_keep; } else { line = e % kTL; p = e / kTL; }
        if (line >= nl) continue;
        const int64_t base = out_base[line];
        const cplx<T> v = buf[p * kPad + line];
        if (STOREM == STORE_COMPLEX) {
            T* q = outb + 2 * (base + (int64_t)p * lo.point_stride);
            q[0] = v.re;
            q[1] = v.im;
        } else {
            outb[base + (int64_t)p * lo.point_stride] = v.re;
        }
    }
}

// ---- register-resident variant for n = 16 * M, M in {2, 4, 8, 16} ------------------------------------------------
// Cooley-Tukey split n = 16 x M: a thread runs a whole 16-point FFT in registers on the points (c + M j1), multiplies
// by W_n^(c k1), and after one exchange through LDS a thread runs the M-point FFT over c for its k1 and stores
// X[k1 + 16 k2]. 256 threads = 16 lines x 16 columns: one barrier instead of log2(n); the butterflies are the packed
// radix-4 ones of chx_fft_reg.h (the LDS radix-2 kernel above stays for n = 512, 1024).
//   ZP ("zero padded"): exactly the first n / 2 input points exist — the loads of the upper half and the first butterfly
//                       layer on it are removed at compile time; otherwise complex / real inputs have all n points
//   KH ("keep half"):   exactly the first n / 2 output points are kept — the stores (and, by dead-code elimination, the
//                       arithmetic) of the upper half are removed at compile time
// Lines of a tile beyond L are computed on the tile's first line and not stored: no load is predicated on it.
using chx_fft::cmul;
using chx_fft::cmul_conj;
using chx_fft::fft16;
using chx_fft::fft_small;

template <typename T, int M>
struct RegTile {
    static constexpr int n = 16 * M;
    static constexpr int LP = kTL + 1;           // line pitch
    static constexpr int KP = M * LP + 1;        // pitch between k1 slabs (odd: spreads the banks)
    static constexpr size_t shmem = ((size_t)16 * KP + n) * sizeof(vec2<T>);
};

template <typename T, int M>
__device__ __forceinline__ void fill_twiddles(vec2<T>* tw, bool inverse) {
    for (int k = threadIdx.x; k < 16 * M; k += CHX_BLOCK) {
        T s, c;
        sincos_2pi<T>(k, 16 * M, s, c);
        tw[k] = vec2<T>{c, inverse ? s : -s};
    }
}

// 16-point transforms of x, twiddle W_n^(c k1), into the exchange buffer
template <typename T, int M, bool INV, bool ZP>
__device__ __forceinline__ void pass1_to_lds(vec2<T> (&x)[16], vec2<T>* xch, const vec2<T>* tw, int c, int line) {
    using RT = RegTile<T, M>;
    fft16<T, INV, ZP>(x);
    xch[c * RT::LP + line] = x[0];
#pragma unroll
    for (int k1 = 1; k1 < 16; ++k1) xch[k1 * RT::KP + c * RT::LP + line] = cmul(x[k1], tw[(c * k1) & (RT::n - 1)]);
}

// RIDER: the launch has ONE workgroup more than the lines need; it runs the bookkeeping step behind the tile deposit of a chain of
// space-charge kicks (sc_tile_schedule_block, chx_sc_tiles.h) — a launch of its own otherwise, between the deposit and this pass.
template <typename T, int LOADM, int STOREM, bool POINT_FAST, int M, bool INV, bool ZP, bool KH, bool RIDER = false>
__global__ __launch_bounds__(CHX_BLOCK) void fft_lines_reg_kernel(const T* in, T* __restrict__ out, int n_valid,
                                                                 int n_keep, int64_t L, int64_t inner_count, LineLayout li,
                                                                 LineLayout lo, T* consume /*LOAD_REAL: = in, zeros are written
                                                                 behind the loads (null: the input is left alone)*/,
                                                                 ScScheduleArgs sched) {
    if constexpr (RIDER) {
        if (blockIdx.x == gridDim.x - 1) {
            __shared__ int sched_part[257];
            if (blockIdx.y == 0) sc_tile_schedule_block(sched, sched_part);
            return;
        }
    }
    using RT = RegTile<T, M>;
    constexpr int n = RT::n;
    constexpr bool kCplxIn = LOADM == LOAD_COMPLEX || LOADM == LOAD_HERMITIAN;
    constexpr bool kPair = LOADM == LOAD_EVEN_REAL_PAIR;
    static_assert(kPair == (STOREM == STORE_REAL_PAIR), "paired lines are loaded and stored as pairs");
    constexpr int kLines = kPair ? 2 * kTL : kTL;          // lines of the arrays per tile
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    vec2<T>* xch = reinterpret_cast<vec2<T>*>(smem_raw);  // [k1][c][line], 16 * KP elements
    vec2<T>* tw = xch + 16 * RT::KP;                       // exp(-+2 pi i k / n), n elements
    __shared__ int64_t in_base[kLines], out_base[kLines];

    const int tid = threadIdx.x;
    const int64_t l0 = (int64_t)blockIdx.x * kLines;
    const int nl = (int)((L - l0 < kLines) ? (L - l0) : kLines);
    const int64_t b = blockIdx.y;
    const T* inb = in + b * li.batch_stride * (kCplxIn ? 2 : 1);
    T* outb = out + b * lo.batch_stride * ((STOREM == STORE_COMPLEX) ? 2 : 1);
    if (tid < kLines) {
        const int64_t l = l0 + (tid < nl ? tid : 0);
        const int64_t outer = l / inner_count, inner = l - outer * inner_count;
        in_base[tid] = outer * li.outer_stride + inner * li.inner_stride;
        out_base[tid] = outer * lo.outer_stride + inner * lo.inner_stride;
    }
    fill_twiddles<T, M>(tw, INV);
    __syncthreads();
    const int line = POINT_FAST ? (tid >> 4) : (tid & 15);
    const int c = POINT_FAST ? (tid & 15) : (tid >> 4);
    const int la = kPair ? 2 * line : line;                // first (or only) array line of this transform
    // ---- pass 1: 16-point FFTs over j1 for column c (< M), then the twiddle W_n^(c k1) --------------------------
    if (c < M) {
        vec2<T> x[16];
        const int64_t base = in_base[la];
        const vec2<T>* qc = reinterpret_cast<const vec2<T>*>(inb) + base;
        const T* qr = inb + base;
        const T* qr2 = inb + in_base[kPair ? la + 1 : la];
#pragma unroll
        for (int j1 = 0; j1 < 16; ++j1) {
            const int p = c + M * j1;
            x[j1] = vec2<T>{(T)0, (T)0};
            if (LOADM == LOAD_COMPLEX) {
                if (!ZP || j1 < 8) x[j1] = qc[(int64_t)p * li.point_stride];
            } else if (LOADM == LOAD_REAL) {
                if (!ZP || j1 < 8) x[j1].x = qr[(int64_t)p * li.point_stride];
            } else if (LOADM == LOAD_HERMITIAN) {   // half spectrum 0..n/2 given: X[n-k] = conj X[k]
                const bool lower = j1 < 8 || (j1 == 8 && c == 0);
                x[j1] = qc[(int64_t)(lower ? p : n - p) * li.point_stride];
                if (!lower) x[j1].y = -x[j1].y;
            } else {                                // even, real: x[n-p] = x[p], values 0..n/2 given
                const bool lower = j1 < 8 || (j1 == 8 && c == 0);
                const int64_t off = (int64_t)(lower ? p : n - p) * li.point_stride;
                x[j1].x = qr[off];
                if (kPair) x[j1].y = qr2[off];
            }
        }
        if (LOADM == LOAD_REAL && consume && la < nl) {
            // the accumulation grid of a chain of space-charge kicks: every cell is read exactly once, by this pass — leave it
            // zeroed for the next kick's deposit (only cells that held charge are written)
            T* zr = consume + b * li.batch_stride + base;
#pragma unroll
            for (int j1 = 0; j1 < (ZP ? 8 : 16); ++j1)
                if (x[j1].x != (T)0) zr[(int64_t)(c + M * j1) * li.point_stride] = (T)0;
        }
        pass1_to_lds<T, M, INV, ZP>(x, xch, tw, c, line);
    }
    // POINT_FAST: a wave holds 4 whole lines (every column and every k1 of each), the exchange never leaves the wave
    if (POINT_FAST) chx_wave_sync();
    else __syncthreads();
    // ---- pass 2: M-point FFT over c for k1 = c (16 of them), output index k1 + 16 k2 ----------------------------
    {
        const int k1 = c;
        vec2<T> y[M];
#pragma unroll
        for (int j2 = 0; j2 < M; ++j2) y[j2] = xch[k1 * RT::KP + j2 * RT::LP + line];
        fft_small<T, M, INV>(y);
        if (la < nl) {
            const int64_t base = out_base[la];
            const int64_t base2 = out_base[kPair ? la + 1 : la];
            const bool second = kPair && la + 1 < nl;
#pragma unroll
            for (int k2 = 0; k2 < M; ++k2) {
                const int p = k1 + 16 * k2;
                if (KH ? (k2 < M / 2) : (p < n_keep)) {
                    if (STOREM == STORE_COMPLEX) {
                        reinterpret_cast<vec2<T>*>(outb)[base + (int64_t)p * lo.point_stride] = y[k2];
                    } else {
                        outb[base + (int64_t)p * lo.point_stride] = y[k2].x;
                        if (second) outb[base2 + (int64_t)p * lo.point_stride] = y[k2].y;
                    }
                }
            }
        }
    }
}

template <typename T, int LOADM, int STOREM, bool POINT_FAST, int M, bool INV, bool ZP, bool KH>
int launch_lines_reg(const void* in, void* out, int n_valid, int n_keep, int64_t L, int64_t inner_count, LineLayout li,
                     LineLayout lo, int64_t B, hipStream_t s, void* consume = nullptr, const ScScheduleArgs* sched = nullptr) {
    constexpr int lines = LOADM == LOAD_EVEN_REAL_PAIR ? 2 * kTL : kTL;
    dim3 grid((unsigned)((L + lines - 1) / lines), (unsigned)B);
    constexpr size_t shmem = RegTile<T, M>::shmem;
    auto kern = fft_lines_reg_kernel<T, LOADM, STOREM, POINT_FAST, M, INV, ZP, KH>;
    if constexpr (LOADM == LOAD_REAL) {
        if (sched) {
            kern = fft_lines_reg_kernel<T, LOADM, STOREM, POINT_FAST, M, INV, ZP, KH, true>;
            grid.x += 1;
        }
    } else if (sched) {
        return CHX_ERR_INVALID_ARG;
    }
    if (shmem > 64 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem) !=
            hipSuccess)
            return CHX_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(kern, grid, dim3(CHX_BLOCK), shmem, s, (const T*)in, (T*)out, n_valid, n_keep, L, inner_count, li, lo,
                       (T*)consume, sched ? *sched : ScScheduleArgs());
    CHX_CHECK_LAUNCH();
    return CHX_OK;
}

// ---- middle pass of the convolution, fused: forward FFT -> multiply by the Green spectrum -> inverse FFT, in place ---------
// The z lines of the twice-transformed charge data[kx <= gx][ky < 2 gy][z < gz] (z >= gz: implicit zeros) are transformed,
// multiplied by the real, even Green spectrum and transformed back without leaving the CU: the full [gx+1][2gy][2gz] complex
// spectrum of the three-kernel form (forward, multiply, inverse) never exists. Line l = kx * ny + ky is contiguous;
// consecutive lanes walk along it, so every access is a whole 128-byte (M = 16) run.
// The inverse runs the forward factorisation backwards — inverse M-point FFTs over k2 in the thread that holds
// X[k1 + 16 k2], conjugate twiddle, one LDS exchange, inverse 16-point FFTs — so no re-ordering pass is needed:
//   X[k1 + 16 k2] = sum_c W_M^(c k2) W_n^(c k1) sum_j1 x[c + M j1] W_16^(j1 k1)
//   x[c + M j1]   = sum_k1 W_16^(-j1 k1) W_n^(-c k1) sum_k2 X[k1 + 16 k2] W_M^(-c k2)
template <typename T, int M>
__global__ __launch_bounds__(CHX_BLOCK) void fft_z_fused_kernel(T* __restrict__ data, const T* __restrict__ gh,
                                                               const double* __restrict__ scale, int64_t L, int ny, int gx) {
    using RT = RegTile<T, M>;
    constexpr int n = RT::n, gz = n / 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    vec2<T>* xch = reinterpret_cast<vec2<T>*>(smem_raw);  // [k1][c][line]
    vec2<T>* tw = xch + 16 * RT::KP;                       // exp(-2 pi i k / n)
    __shared__ int g_base[kTL];

    const int tid = threadIdx.x;
    const int64_t l0 = (int64_t)blockIdx.x * kTL;
    const int nl = (int)((L - l0 < kTL) ? (L - l0) : kTL);
    const int64_t b = blockIdx.y;
    const int gy1 = ny / 2 + 1;
    vec2<T>* db = reinterpret_cast<vec2<T>*>(data) + b * L * gz;
    const T* gb = gh + b * (int64_t)(gx + 1) * gy1 * (gz + 1);
    const T sc = (T)scale[b];
    if (tid < kTL) {
        const int64_t l = l0 + (tid < nl ? tid : 0);        // l = kx * ny + ky -> Ghat[kx][|ky|][.]
        const int kx = (int)(l / ny), ky = (int)(l - (int64_t)kx * ny);
        g_base[tid] = (kx * gy1 + ((ky <= ny / 2) ? ky : ny - ky)) * (gz + 1);
    }
    fill_twiddles<T, M>(tw, false);
    __syncthreads();
    const int line = tid >> 4, c = tid & 15;
    const bool live = line < nl;
    vec2<T>* q = db + (l0 + (live ? line : 0)) * gz + c;    // point c of this line
    // ---- forward, pass 1 (points c + M j1, j1 < 8 exist)
    if (c < M) {
        vec2<T> x[16];
#pragma unroll
        for (int j1 = 0; j1 < 16; ++j1) x[j1] = j1 < 8 ? q[M * j1] : vec2<T>{(T)0, (T)0};
        pass1_to_lds<T, M, false, true>(x, xch, tw, c, line);
    }
    // a wave holds 4 whole lines (all 16 columns c and all 16 k1 of each): both exchanges stay inside the wave, so the waves
    // of a workgroup need no barrier between them and drift apart (loads of one overlap the butterflies of another)
    chx_wave_sync();
    // ---- forward pass 2, multiply, inverse pass 1 (all in the registers of the thread that owns k1)
    {
        const int k1 = c;
        vec2<T> y[M];
#pragma unroll
        for (int j2 = 0; j2 < M; ++j2) y[j2] = xch[k1 * RT::KP + j2 * RT::LP + line];
        fft_small<T, M, false>(y);
        const T* gl = gb + g_base[line];
#pragma unroll
        for (int k2 = 0; k2 < M; ++k2) {
            const int p = k1 + 16 * k2;                     // p <= n/2  <=>  k2 < M/2 or (k2 == M/2 and k1 == 0)
            const int sk = (k2 < M / 2 || (k2 == M / 2 && k1 == 0)) ? p : n - p;
            const T g = gl[sk] * sc;
            y[k2] *= vec2<T>{g, g};
        }
        fft_small<T, M, true>(y);
        xch[k1 * RT::KP + line] = y[0];
#pragma unroll
        for (int cc = 1; cc < M; ++cc)                       // conjugate twiddle W_n^(-c k1); this thread's own slab
            xch[k1 * RT::KP + cc * RT::LP + line] = cmul_conj(y[cc], tw[(cc * k1) & (n - 1)]);
    }
    chx_wave_sync();
    // ---- inverse pass 2: 16-point inverse FFTs over k1, outputs x[c + M j1], j1 < 8 kept
    if (c < M) {
        vec2<T> x[16];
#pragma unroll
        for (int k1 = 0; k1 < 16; ++k1) x[k1] = xch[k1 * RT::KP + c * RT::LP + line];
        fft16<T, true>(x);
        if (live) {
#pragma unroll
            for (int j1 = 0; j1 < 8; ++j1) q[M * j1] = x[j1];
        }
    }
}

// z lines of data[kx <= gx][ky < ny][z < gz] (complex), Green spectrum gh[(gx+1)][(gy+1)][(gz+1)]
template <typename T, int M>
int launch_z_fused(void* data, const void* gh, const double* scale, int gx, int gy, int64_t B, hipStream_t s) {
    const int ny = 2 * gy;
    const int64_t L = (int64_t)(gx + 1) * ny;
    dim3 grid((unsigned)((L + kTL - 1) / kTL), (unsigned)B);
    constexpr size_t shmem = RegTile<T, M>::shmem;
    auto kern = fft_z_fused_kernel<T, M>;
    if (shmem > 64 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem) !=
            hipSuccess)
            return CHX_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(kern, grid, dim3(CHX_BLOCK), shmem, s, (T*)data, (const T*)gh, scale, L, ny, gx);
    CHX_CHECK_LAUNCH();
    return CHX_OK;
}

// ZP / KH are promises of the call site about n_valid / n_keep (checked here)
template <typename T, int LOADM, int STOREM, bool POINT_FAST, bool INV, bool ZP = false, bool KH = false>
int launch_lines(const void* in, void* out, int n, int n_valid, int n_keep, int64_t L, int64_t inner_count, LineLayout li,
                 LineLayout lo, int64_t B, hipStream_t s, void* consume = nullptr, const ScScheduleArgs* sched = nullptr) {
    if ((ZP && 2 * n_valid != n) || (KH && 2 * n_keep != n)) return CHX_ERR_INVALID_ARG;
    if ((LOADM == LOAD_COMPLEX || LOADM == LOAD_REAL) && !ZP && n_valid != n) return CHX_ERR_INVALID_ARG;
    switch (n) {
        case 32: return launch_lines_reg<T, LOADM, STOREM, POINT_FAST, 2, INV, ZP, KH>(in, out, n_valid, n_keep, L, inner_count, li, lo, B, s, consume, sched);
        case 64: return launch_lines_reg<T, LOADM, STOREM, POINT_FAST, 4, INV, ZP, KH>(in, out, n_valid, n_keep, L, inner_count, li, lo, B, s, consume, sched);
        case 128: return launch_lines_reg<T, LOADM, STOREM, POINT_FAST, 8, INV, ZP, KH>(in, out, n_valid, n_keep, L, inner_count, li, lo, B, s, consume, sched);
        case 256: return launch_lines_reg<T, LOADM, STOREM, POINT_FAST, 16, INV, ZP, KH>(in, out, n_valid, n_keep, L, inner_count, li, lo, B, s, consume, sched);
        default: break;
    }
    if (sched) return CHX_ERR_INVALID_ARG;          // (chx_sc_convolve_carries_schedule: lines of at most 256 points)
    int log2n = 0;
    while ((1 << log2n) < n) ++log2n;
    const size_t shmem = ((size_t)n * kPad + n / 2) * sizeof(cplx<T>);
    // the LDS radix-2 kernel (n = 512, 1024) transforms the lines of a pair one by one
    constexpr int LM = LOADM == LOAD_EVEN_REAL_PAIR ? (int)LOAD_EVEN_REAL : LOADM;
    constexpr int SM = STOREM == STORE_REAL_PAIR ? (int)STORE_REAL : STOREM;
    auto kern = fft_lines_kernel<T, LM, SM, POINT_FAST>;
    if (shmem > 64 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem) !=
            hipSuccess)
            return CHX_ERR_LAUNCH;
    }
    dim3 grid((unsigned)((L + kTL - 1) / kTL), (unsigned)B);
    hipLaunchKernelGGL(kern, grid, dim3(CHX_BLOCK), shmem, s, (const T*)in, (T*)out, n, log2n, n_valid, n_keep, L, inner_count,
                       li, lo, INV ? 1 : 0);
    CHX_CHECK_LAUNCH();
    return CHX_OK;
}

// Compact Green function Gc[b][i][j][k], i, j, k in [0, g]: the signed 8-term difference of the primitive table
// (chx_spacecharge.hip igf_fill_kernel) for indices < g and 0 on the index-g planes (space_charge_kick.py:249-289).
template <typename T>
__global__ __launch_bounds__(CHX_BLOCK) void igf_compact_kernel(const double* __restrict__ table, int gx, int gy, int gz,
                                                               T* __restrict__ Gc) {
    const int64_t b = blockIdx.y;
    const int64_t npts = (int64_t)(gx + 1) * (gy + 1) * (gz + 1);
    const double* tb = table + b * npts;
    const int64_t sy = gz + 1, sx = (int64_t)(gy + 1) * (gz + 1);
    // 32-bit index arithmetic: (g + 1)^3 <= 513^3 < 2^31 (64-bit integer division is ~100 instructions on this target and
    // used to be most of this kernel's 38 us)
    const unsigned n1u = (unsigned)(gz + 1), n2u = (unsigned)(gy + 1);
    for (unsigned idx = blockIdx.x * CHX_BLOCK + threadIdx.x; idx < (unsigned)npts; idx += gridDim.x * CHX_BLOCK) {
        const unsigned q1 = idx / n1u;
        const int k = (int)(idx - q1 * n1u);
        const unsigned q2 = q1 / n2u;
        const int j = (int)(q1 - q2 * n2u);
        const int i = (int)q2;
        double g = 0.0;
        if (i < gx && j < gy && k < gz) {
            const double* p = tb + i * sx + j * sy + k;
            g = p[sx + sy + 1] - p[sy + 1] - p[sx + 1] - p[sx + sy] + p[sx] + p[sy] + p[1] - p[0];
        }
        Gc[b * npts + idx] = (T)g;
    }
}

// ---- far-field form of the integrated Green function (fp32 grids) ------------------------------------------------
// G_ijk = integral of 1/r over the cell centred at R = (i dx, j dy, k dt) (space_charge_kick.py:170-236 evaluates it as
// the signed 8-corner difference of the primitive F, 48 fp64 transcendentals per cell there, 6 per corner point here).
// For |R| >= kFarRatio * max(dx, dy, dt) the Taylor expansion of 1/|R + s| about the cell centre, integrated over the
// cell, is used instead (odd orders vanish by symmetry; with c_a = R_a^2 / R^2, e_a = h_a^2 / R^2):
//   G = V / R * [ 1 + sum_a e_a (3 c_a - 1) / 24
//                   + sum_a e_a^2 (105 c_a^2 - 90 c_a + 9) / 1920
//                   + sum_{a<b} e_a e_b (105 c_a c_b - 15 (c_a + c_b) + 3) / 576 ]  + O(e^3)
// Relative error <= 1e-8 at ratio 8 (checked against the 8-corner form in 80-bit arithmetic for isotropic and 1 : 1.3 : 440
// cells) — six times below the fp32 rounding of the stored value, and smaller than the cancellation error of the
// reference's own fp64 corner differences for stretched cells (5e-7 at 24 cells' distance for the 1 : 1.3 : 440 case).
// Only the fp32 path uses it; fp64 grids keep the corner table everywhere.
constexpr double kFarRatio = 8.0;

struct FarGeom {
    double dx, dy, dt, thr2;
};

template <typename T>
__device__ __forceinline__ FarGeom far_geom(const T* __restrict__ cell, const T* __restrict__ gamma, int64_t b) {
    FarGeom f;
    f.dx = (double)cell[b * 3 + 0];
    f.dy = (double)cell[b * 3 + 1];
    f.dt = (double)(T)(cell[b * 3 + 2] * gamma[b]);   // space_charge_kick.py:170-176, product in the working dtype
    const double hm = fmax(f.dx, fmax(f.dy, f.dt));
    f.thr2 = (kFarRatio * hm) * (kFarRatio * hm);
    return f;
}

// margin: the table keeps points up to 1.001 x the threshold, the compact kernel (fp32 test, 1.0001 x) calls a cell near
// only well inside that — every corner a near cell reads is therefore in the table
__device__ __forceinline__ bool far_for_table(const FarGeom& f, int i, int j, int k) {
    const double x = i * f.dx, y = j * f.dy, z = k * f.dt;
    return x * x + y * y + z * z >= f.thr2 * 1.001;
}

// Corner table restricted to the points a near cell needs: point (i, j, k) is a corner of the cells (i-1..i, j-1..j,
// k-1..k); |R| grows with every index, so the point is needed iff its smallest adjacent cell is near. The thread index
// runs fastest along the axis with the SMALLEST cell (the near region is longest there: for a relativistic bunch
// dt = gamma * dz dominates and the near set is the slab k < 8): whole waves are skipped or kept.
template <typename T>
__global__ __launch_bounds__(CHX_BLOCK) void igf_table_near_kernel(const T* cell, const T* gamma,
                                                                  int gx, int gy, int gz, double* __restrict__ table, ScGeoSums rider) {
    const int64_t b = blockIdx.y;
    // rider (chx_sc_geom_dev.h; B = 1): every workgroup forms the kick's geometry from the sums the previous gather pass left and takes
    // cell / gamma from it; workgroup 0 stores it as the side stream's copy for the kernels behind this one
    __shared__ double geo_red[(CHX_BLOCK / 16 + 1) * 8];
    __shared__ T geo_s[kScGeoValues];
    __shared__ double pot_s[1];
    if (rider.sums) {
        sc_geo_from_sums<T, CHX_BLOCK>(rider, geo_red, geo_s, pot_s, blockIdx.x == 0 && blockIdx.y == 0);
        cell = geo_s + 3;
        gamma = geo_s + 6;
    }
    const FarGeom f = far_geom<T>(cell, gamma, b);
    const int64_t npts = (int64_t)(gx + 1) * (gy + 1) * (gz + 1);
    const int fast = (f.dx <= f.dy && f.dx <= f.dt) ? 0 : (f.dy <= f.dt ? 1 : 2);
    const int n1 = gy + 1, n2 = gz + 1;
    // bounding box of the needed points: point p is needed only if (p - 1) h < 1.001^(1/2) x threshold along its axis (the
    // threads run over the box, not over the (g + 1)^3 points: for a relativistic bunch it is the slab k <= 9 of 129)
    const double reach = sqrt(f.thr2 * 1.001);
    const double h[3] = {f.dx, f.dy, f.dt};
    const int full[3] = {gx + 1, gy + 1, gz + 1};
    int box[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const double m = reach / h[d] + 2.0;
        box[d] = m < (double)full[d] ? (int)m : full[d];
    }
    // extents in thread order (fast, mid, slow)
    const int nf = box[fast];
    const int nm = fast == 0 ? box[1] : box[0];                       // mid = x unless x is the fast axis
    const unsigned nbox = (unsigned)box[0] * (unsigned)box[1] * (unsigned)box[2];
    for (unsigned idx = blockIdx.x * CHX_BLOCK + threadIdx.x; idx < nbox; idx += gridDim.x * CHX_BLOCK) {
        const unsigned q1 = idx / (unsigned)nf;
        const int pf = (int)(idx - q1 * (unsigned)nf);
        const unsigned q2 = q1 / (unsigned)nm;
        const int pm = (int)(q1 - q2 * (unsigned)nm);
        const int ps = (int)q2;
        // fast 0: (x, y, z) = (pf, pm, ps); fast 1: (pm, pf, ps); fast 2: (pm, ps, pf)
        const int i = fast == 0 ? pf : pm;
        const int j = fast == 0 ? pm : (fast == 1 ? pf : ps);
        const int k = fast == 2 ? pf : ps;
        if (far_for_table(f, i > 0 ? i - 1 : 0, j > 0 ? j - 1 : 0, k > 0 ? k - 1 : 0)) continue;
        table[b * npts + ((int64_t)i * n1 + j) * n2 + k] = igf_primitive<double>((i - 0.5) * f.dx, (j - 0.5) * f.dy, (k - 0.5) * f.dt);
    }
}

// Compact Green function like igf_compact_kernel, far cells by the expansion above, near cells from the corner table.
template <typename T>
__global__ __launch_bounds__(CHX_BLOCK) void igf_compact_far_kernel(const double* __restrict__ table, const T* __restrict__ cell,
                                                                   const T* __restrict__ gamma, int gx, int gy, int gz,
                                                                   unsigned long long magic_z, unsigned long long magic_y,
                                                                   T* __restrict__ Gc) {
    const int64_t b = blockIdx.y;
    const FarGeom f = far_geom<T>(cell, gamma, b);
    const int64_t npts = (int64_t)(gx + 1) * (gy + 1) * (gz + 1);
    const double* tb = table + b * npts;
    const int64_t sy = gz + 1, sx = (int64_t)(gy + 1) * (gz + 1);
    // The far cells are evaluated in fp32 (this kernel only serves fp32 grids: the stored value is rounded to fp32 anyway):
    // the leading term V / R comes from v_rsq_f32 refined by one Newton step, the bracket is 1 + O(1e-2) — about 1e-7
    // relative in total, where the fp64 form (one division, one square root and ~70 flops per cell) cost 33 us next to the
    // deposit kernels it shares the CUs with.
    const float dxf = (float)f.dx, dyf = (float)f.dy, dtf = (float)f.dt;
    const float Vf = (float)(f.dx * f.dy * f.dt), hx2 = dxf * dxf, hy2 = dyf * dyf, hz2 = dtf * dtf;
    const float thr2f = (float)f.thr2;
    // index arithmetic: (g + 1)^3 <= 513^3 < 2^28, and idx / d = (idx * (2^38 / d + 1)) >> 38 exactly for idx < 2^28, d < 2^10
    // (two unsigned divisions by run-time values were half of this kernel's 10 us)
    const unsigned n1u = (unsigned)(gz + 1), n2u = (unsigned)(gy + 1);
    for (unsigned idx = blockIdx.x * CHX_BLOCK + threadIdx.x; idx < (unsigned)npts; idx += gridDim.x * CHX_BLOCK) {
        const unsigned q1 = (unsigned)(((unsigned long long)idx * magic_z) >> 38);
        const int k = (int)(idx - q1 * n1u);
        const unsigned q2 = (unsigned)(((unsigned long long)q1 * magic_y) >> 38);
        const int j = (int)(q1 - q2 * n2u);
        const int i = (int)q2;
        T g = (T)0;
        if (i < gx && j < gy && k < gz) {
            const float x = (float)i * dxf, y = (float)j * dyf, z = (float)k * dtf;
            const float x2 = x * x, y2 = y * y, z2 = z * z, R2 = x2 + y2 + z2;
            // fp32 test with a 1.0001 margin; the table was filled up to 1.001 x the threshold (far_for_table), so every
            // corner of a cell taken as near here is present
            if (R2 >= thr2f * 1.0001f) {
                float rinv = __frsqrt_rn(R2);
                rinv = rinv * (1.5f - 0.5f * R2 * rinv * rinv);
                const float inv = rinv * rinv;
                const float cx = x2 * inv, cy = y2 * inv, cz = z2 * inv;
                const float ex = hx2 * inv, ey = hy2 * inv, ez = hz2 * inv;
                const float t2 = (ex * (3.0f * cx - 1.0f) + ey * (3.0f * cy - 1.0f) + ez * (3.0f * cz - 1.0f)) * (1.0f / 24.0f);
                const float t4a = ex * ex * ((105.0f * cx - 90.0f) * cx + 9.0f) + ey * ey * ((105.0f * cy - 90.0f) * cy + 9.0f) +
                                  ez * ez * ((105.0f * cz - 90.0f) * cz + 9.0f);
                const float t4b = ex * ey * (105.0f * cx * cy - 15.0f * (cx + cy) + 3.0f) +
                                  ex * ez * (105.0f * cx * cz - 15.0f * (cx + cz) + 3.0f) +
                                  ey * ez * (105.0f * cy * cz - 15.0f * (cy + cz) + 3.0f);
                g = (T)(Vf * rinv * (1.0f + t2 + t4a * (1.0f / 1920.0f) + t4b * (1.0f / 576.0f)));
            } else {
                const double* p = tb + i * sx + j * sy + k;
                g = (T)(p[sx + sy + 1] - p[sy + 1] - p[sx + 1] - p[sx + sy] + p[sx] + p[sy] + p[1] - p[0]);
            }
        }
        Gc[b * npts + idx] = g;
    }
}

// rho_hat[b][kx <= gx][ky][kz] *= Ghat[b][kx][min(ky, ny-ky)][min(kz, nz-kz)] * scale[b]
template <typename T>
__global__ __launch_bounds__(CHX_BLOCK) void spectral_mul_sym_kernel(T* __restrict__ a, const T* __restrict__ gh,
                                                                    const double* __restrict__ scale, int nxc, int ny, int nz) {
    const int64_t b = blockIdx.y;
    const int64_t ntot = (int64_t)nxc * ny * nz;
    const T sc = (T)scale[b];
    T* ab = a + b * ntot * 2;
    const int gy1 = ny / 2 + 1, gz1 = nz / 2 + 1;
    const T* gb = gh + b * (int64_t)nxc * gy1 * gz1;
    for (int64_t idx = (int64_t)blockIdx.x * CHX_BLOCK + threadIdx.x; idx < ntot; idx += (int64_t)gridDim.x * CHX_BLOCK) {
        const int kz = (int)(idx % nz);
        const int ky = (int)((idx / nz) % ny);
        const int kx = (int)(idx / ((int64_t)nz * ny));
        const int syk = (ky <= ny / 2) ? ky : ny - ky, szk = (kz <= nz / 2) ? kz : nz - kz;
        const T g = gb[((int64_t)kx * gy1 + syk) * gz1 + szk] * sc;
        ab[2 * idx] *= g;
        ab[2 * idx + 1] *= g;
    }
}

// the tile [2g][17] complex + twiddles must fit the 160 KB of LDS of a CU: 2g <= 1024 (fp32) / 512 (fp64)
bool pow2_ok(int g, int dtype) { return g >= 16 && g <= (dtype == CHX_F32 ? 512 : 256) && (g & (g - 1)) == 0; }

}  // namespace

extern "C" int chx_sc_pruned_supported(const int32_t* bins, int dtype) {
    return (dtype == CHX_F32 || dtype == CHX_F64) && bins && pow2_ok(bins[0], dtype) && pow2_ok(bins[1], dtype) &&
           pow2_ok(bins[2], dtype);
}

// workspace layout of chx_sc_green_spectrum: [Gc (g+1)^3 T][H (g+1)^3 T] per batch row
extern "C" size_t chx_sc_green_workspace_bytes(int64_t B, const int32_t* bins, int dtype) {
    if (B < 1 || !chx_sc_pruned_supported(bins, dtype)) return 0;
    const size_t esz = dtype == CHX_F32 ? 4 : 8;
    return (size_t)B * 2 * (size_t)(bins[0] + 1) * (bins[1] + 1) * (bins[2] + 1) * esz;
}

// cell / gamma non-null: far-field compact kernel (the table then only holds the near points)
template <typename T>
static int green_spectrum_impl(const double* table, const T* cell, const T* gamma, int64_t B, const int32_t* bins, T* Ghat,
                               T* ws, hipStream_t s) {
    const int gx = bins[0], gy = bins[1], gz = bins[2];
    const int64_t n1 = (int64_t)(gx + 1) * (gy + 1) * (gz + 1);
    T* Gc = ws;
    T* H = ws + B * n1;
    int grid = chx_grid_for(n1, CHX_BLOCK, 4096);
    if (cell)
        hipLaunchKernelGGL(igf_compact_far_kernel<T>, dim3((unsigned)grid, (unsigned)B), dim3(CHX_BLOCK), 0, s, table, cell,
                           gamma, gx, gy, gz, (1ull << 38) / (unsigned)(gz + 1) + 1, (1ull << 38) / (unsigned)(gy + 1) + 1, Gc);
    else
        hipLaunchKernelGGL(igf_compact_kernel<T>, dim3((unsigned)grid, (unsigned)B), dim3(CHX_BLOCK), 0, s, table, gx, gy, gz, Gc);
    CHX_CHECK_LAUNCH();
    const int64_t sy = gz + 1, sx = (int64_t)(gy + 1) * (gz + 1);
    // z: lines (x, y), points contiguous; even extension in, real out (kz <= gz)           Gc -> H
    LineLayout lz{1, sy, sx, n1};
    int st = launch_lines<T, LOAD_EVEN_REAL_PAIR, STORE_REAL_PAIR, true, false>(Gc, H, 2 * gz, gz + 1, gz + 1,
                                                                      (int64_t)(gx + 1) * (gy + 1), gy + 1, lz, lz, B, s);
    if (st != CHX_OK) return st;
    // y: lines (x, kz), point stride sy                                                    H -> Gc
    LineLayout ly{sy, 1, sx, n1};
    st = launch_lines<T, LOAD_EVEN_REAL_PAIR, STORE_REAL_PAIR, false, false>(H, Gc, 2 * gy, gy + 1, gy + 1,
                                                                   (int64_t)(gx + 1) * (gz + 1), gz + 1, ly, ly, B, s);
    if (st != CHX_OK) return st;
    // x: lines (ky, kz) = one contiguous index, point stride sx                            Gc -> Ghat
    LineLayout lx{sx, 1, 0, n1};
    return launch_lines<T, LOAD_EVEN_REAL_PAIR, STORE_REAL_PAIR, false, false>(Gc, Ghat, 2 * gx, gx + 1, gx + 1, sx, sx, lx, lx, B, s);
}

// Real, even spectrum of the integrated Green function on the doubled grid, stored on (gx+1)(gy+1)(gz+1) points.
// `table` is the corner table written by chx_sc_igf_table (double [B][(gx+1)(gy+1)(gz+1)]).
extern "C" int chx_sc_green_spectrum(const double* table, int64_t B, const int32_t* bins, int dtype, void* Ghat,
                                     void* workspace, size_t workspace_bytes, void* stream) {
    if (!table || !Ghat || B < 1 || B > 65535 || !chx_sc_pruned_supported(bins, dtype)) return CHX_ERR_INVALID_ARG;
    if (dtype != CHX_F32 && dtype != CHX_F64) return CHX_ERR_DTYPE;
    if (!workspace || workspace_bytes < chx_sc_green_workspace_bytes(B, bins, dtype)) return CHX_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    return dtype == CHX_F32 ? green_spectrum_impl<float>(table, nullptr, nullptr, B, bins, (float*)Ghat, (float*)workspace, s)
                            : green_spectrum_impl<double>(table, nullptr, nullptr, B, bins, (double*)Ghat, (double*)workspace, s);
}

// workspace of chx_sc_green_spectrum_fast: the corner table (double) followed by chx_sc_green_spectrum's workspace
extern "C" size_t chx_sc_green_fast_workspace_bytes(int64_t B, const int32_t* bins, int dtype) {
    if (B < 1 || !chx_sc_pruned_supported(bins, dtype)) return 0;
    const size_t npts = (size_t)(bins[0] + 1) * (bins[1] + 1) * (bins[2] + 1);
    return (((size_t)B * npts * sizeof(double) + 255) & ~(size_t)255) + chx_sc_green_workspace_bytes(B, bins, dtype);
}

// Green spectrum straight from the cell sizes: chx_sc_igf_table + chx_sc_green_spectrum in one call. fp32 grids evaluate
// the primitive only where the far-field expansion (above) is not accurate to fp32 rounding; fp64 grids are exact everywhere.
extern "C" int chx_sc_green_spectrum_fast(const void* cell, const void* gamma, int64_t B, const int32_t* bins, int dtype,
                                          void* Ghat, void* workspace, size_t workspace_bytes, void* stream) {
    if (!cell || !gamma || !Ghat || B < 1 || B > 65535 || !chx_sc_pruned_supported(bins, dtype)) return CHX_ERR_INVALID_ARG;
    if (!workspace || workspace_bytes < chx_sc_green_fast_workspace_bytes(B, bins, dtype)) return CHX_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const size_t npts = (size_t)(bins[0] + 1) * (bins[1] + 1) * (bins[2] + 1);
    double* table = (double*)workspace;
    char* rest = (char*)workspace + (((size_t)B * npts * sizeof(double) + 255) & ~(size_t)255);
    if (dtype == CHX_F64) {
        int st = chx_sc_igf_table(cell, gamma, B, bins, dtype, table, stream);
        if (st != CHX_OK) return st;
        return green_spectrum_impl<double>(table, nullptr, nullptr, B, bins, (double*)Ghat, (double*)rest, s);
    }
    const int grid = chx_grid_for((int64_t)npts, CHX_BLOCK, 2048);
    hipLaunchKernelGGL(igf_table_near_kernel<float>, dim3((unsigned)grid, (unsigned)B), dim3(CHX_BLOCK), 0, s, (const float*)cell,
                       (const float*)gamma, bins[0], bins[1], bins[2], table, ScGeoSums());
    CHX_CHECK_LAUNCH();
    return green_spectrum_impl<float>(table, (const float*)cell, (const float*)gamma, B, bins, (float*)Ghat, (float*)rest, s);
}

// chx_sc_green_spectrum_fast inside a chain kick (chx_sc_tiles.h): B = 1; with a rider the corner-table launch forms cell / gamma itself
int chx_sc_green_spectrum_chain(const void* cell, const void* gamma, const int32_t* bins, int dtype, void* Ghat, void* workspace,
                                size_t workspace_bytes, const ScGeoSums* rider, void* stream) {
    if (!rider || !rider->sums) return chx_sc_green_spectrum_fast(cell, gamma, 1, bins, dtype, Ghat, workspace, workspace_bytes, stream);
    if (dtype != CHX_F32 || !Ghat || !chx_sc_pruned_supported(bins, dtype)) return CHX_ERR_INVALID_ARG;
    if (!workspace || workspace_bytes < chx_sc_green_fast_workspace_bytes(1, bins, dtype)) return CHX_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const size_t npts = (size_t)(bins[0] + 1) * (bins[1] + 1) * (bins[2] + 1);
    double* table = (double*)workspace;
    char* rest = (char*)workspace + ((npts * sizeof(double) + 255) & ~(size_t)255);
    const float* geo = (const float*)rider->geo_out;      // the kernels behind the corner table read the copy its workgroup 0 stores
    const int grid = chx_grid_for((int64_t)npts, CHX_BLOCK, 2048);
    hipLaunchKernelGGL(igf_table_near_kernel<float>, dim3((unsigned)grid, 1), dim3(CHX_BLOCK), 0, s, geo + 3, geo + 6, bins[0], bins[1],
                       bins[2], table, *rider);
    CHX_CHECK_LAUNCH();
    return green_spectrum_impl<float>(table, geo + 3, geo + 6, 1, bins, (float*)Ghat, (float*)rest, s);
}

// workspace of chx_sc_convolve per batch row: A [gx+1][gy][gz], Bf [gx+1][2gy][gz] complex, and for line lengths the fused
// middle pass does not cover (2 gz > 256) the full z spectrum C [gx+1][2gy][2gz]
static bool z_fused_ok(int gz) { return 2 * gz <= 256; }

extern "C" size_t chx_sc_convolve_workspace_bytes(int64_t B, const int32_t* bins, int dtype) {
    if (B < 1 || !chx_sc_pruned_supported(bins, dtype)) return 0;
    const size_t esz = dtype == CHX_F32 ? 4 : 8;
    const size_t nxc = bins[0] + 1, gy = bins[1], gz = bins[2];
    return (size_t)B * 2 * esz * (nxc * gy * gz + nxc * 2 * gy * gz + (z_fused_ok(bins[2]) ? 0 : nxc * 2 * gy * 2 * gz));
}

// Pass order x (real -> half spectrum), y, z: the half-spectrum axis (g + 1 planes, an odd count) is the SLOWEST one, so every
// strided pass moves 16-line tiles that are whole, 128-byte aligned cache lines (gz is a multiple of 16), and the heaviest pass
// — forward z, multiply, inverse z — runs in place along the contiguous axis.
template <typename T>
static int convolve_impl(const T* rho, const T* Ghat, const double* scale, int64_t B, const int32_t* bins, T* phi, T* ws,
                         hipStream_t s, bool halo, hipEvent_t ghat_ready = nullptr, bool consume_rho = false,
                         const ScScheduleArgs* sched = nullptr) {
    const int gx = bins[0], gy = bins[1], gz = bins[2];
    const int nx = 2 * gx, ny = 2 * gy, nz = 2 * gz, nxc = gx + 1;
    const int64_t nA = (int64_t)nxc * gy * gz, nB = (int64_t)nxc * ny * gz, nC = (int64_t)nxc * ny * nz;  // complex elements
    T* A = ws;
    T* Bf = A + 2 * B * nA;
    T* C = Bf + 2 * B * nB;
    const int64_t g3 = (int64_t)gx * gy * gz, yz = (int64_t)gy * gz;
    // forward x: rho[x < gx][y][z] real, lines (y, z) -> A[kx <= gx][y][z]
    LineLayout rx{yz, 1, 0, g3};
    LineLayout ax{yz, 1, 0, nA};
    // consume_rho: rho is the accumulation grid of a chain of kicks — the pass that reads it leaves zeros behind
    const bool in_pass = consume_rho && nx <= 256;
    int st = launch_lines<T, LOAD_REAL, STORE_COMPLEX, false, false, true>(rho, A, nx, gx, nxc, yz, yz, rx, ax, B, s,
                                                                           in_pass ? (void*)rho : nullptr, sched);
    if (st != CHX_OK) return st;
    if (consume_rho && !in_pass && hipMemsetAsync((void*)rho, 0, (size_t)B * g3 * sizeof(T), s) != hipSuccess) return CHX_ERR_LAUNCH;
    // forward y: A lines (kx, z), y < gy valid -> Bf[kx][ky < ny][z]
    LineLayout ay{gz, 1, yz, nA};
    LineLayout by{gz, 1, (int64_t)ny * gz, nB};
    st = launch_lines<T, LOAD_COMPLEX, STORE_COMPLEX, false, false, true>(A, Bf, ny, gy, ny, (int64_t)nxc * gz, gz, ay, by, B, s);
    if (st != CHX_OK) return st;
    // z: forward, multiply by the Green spectrum, inverse, z < gz kept — in place on Bf. The spectrum is first needed HERE: a
    // caller that computes it on another stream hands over its completion event, and the two forward passes above run
    // without waiting for it (a cross-queue dependency costs ~13 us to resolve on MI355X: waiting before the first pass
    // left the main queue idle for 17 us per kick)
    if (ghat_ready && hipStreamWaitEvent(s, ghat_ready, 0) != hipSuccess) return CHX_ERR_LAUNCH;
    switch (nz) {
        case 32: st = launch_z_fused<T, 2>(Bf, Ghat, scale, gx, gy, B, s); break;
        case 64: st = launch_z_fused<T, 4>(Bf, Ghat, scale, gx, gy, B, s); break;
        case 128: st = launch_z_fused<T, 8>(Bf, Ghat, scale, gx, gy, B, s); break;
        case 256: st = launch_z_fused<T, 16>(Bf, Ghat, scale, gx, gy, B, s); break;
        default: {
            const int64_t L = (int64_t)nxc * ny;
            LineLayout bz{1, gz, 0, nB};
            LineLayout cz{1, nz, 0, nC};
            st = launch_lines<T, LOAD_COMPLEX, STORE_COMPLEX, true, false, true>(Bf, C, nz, gz, nz, L, L, bz, cz, B, s);
            if (st != CHX_OK) return st;
            const int grid = chx_grid_for(nC, CHX_BLOCK * 4, 8192);
            hipLaunchKernelGGL(spectral_mul_sym_kernel<T>, dim3((unsigned)grid, (unsigned)B), dim3(CHX_BLOCK), 0, s, C, Ghat, scale,
                               nxc, ny, nz);
            CHX_CHECK_LAUNCH();
            st = launch_lines<T, LOAD_COMPLEX, STORE_COMPLEX, true, true, false, true>(C, Bf, nz, nz, gz, L, L, cz, bz, B, s);
        }
    }
    if (st != CHX_OK) return st;
    // inverse y: Bf -> A[kx][y < gy][z]
    st = launch_lines<T, LOAD_COMPLEX, STORE_COMPLEX, false, true, false, true>(Bf, A, ny, ny, gy, (int64_t)nxc * gz, gz, by, ay, B, s);
    if (st != CHX_OK) return st;
    // inverse x: half spectra A[kx <= gx], lines (y, z) -> phi[x < gx][y][z] real, compact or inside a halo of kHalo nodes
    const int64_t pz = gz + (halo ? 2 * kHalo : 0), py = (gy + (halo ? 2 * kHalo : 0)) * pz;
    LineLayout ai{yz, 1, gz, nA};
    LineLayout po{py, 1, pz, (gx + (halo ? 2 * kHalo : 0)) * py};
    T* origin = phi + (halo ? (kHalo * py + kHalo * pz + kHalo) : 0);
    return launch_lines<T, LOAD_HERMITIAN, STORE_REAL, false, true, false, true>(A, origin, nx, nxc, gx, yz, gz, ai, po, B, s);
}

// phi[B][gx][gy][gz] = crop( ifft( fft(pad(rho)) * Ghat * scale ) ), unnormalised transforms (fold 1/(8 gx gy gz) into
// scale). rho[B][gx][gy][gz] compact (not padded), Ghat from chx_sc_green_spectrum.
extern "C" int chx_sc_convolve(const void* rho, const void* Ghat, const double* scale, int64_t B, const int32_t* bins,
                               int dtype, void* phi, void* workspace, size_t workspace_bytes, void* stream) {
    if (!rho || !Ghat || !scale || !phi || B < 1 || B > 65535 || !chx_sc_pruned_supported(bins, dtype)) return CHX_ERR_INVALID_ARG;
    if (dtype != CHX_F32 && dtype != CHX_F64) return CHX_ERR_DTYPE;
    if (!workspace || workspace_bytes < chx_sc_convolve_workspace_bytes(B, bins, dtype)) return CHX_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    return dtype == CHX_F32 ? convolve_impl<float>((const float*)rho, (const float*)Ghat, scale, B, bins, (float*)phi,
                                                   (float*)workspace, s, false)
                            : convolve_impl<double>((const double*)rho, (const double*)Ghat, scale, B, bins, (double*)phi,
                                                    (double*)workspace, s, false);
}

// The same convolution with phi stored inside a halo of 2 nodes: phi_halo[B][gx+4][gy+4][gz+4], node (i, j, k) at
// [i+2][j+2][k+2]. The halo is NOT written (its content is undefined): chx_sc_gather_kick_phi reads it only where the
// result is discarded. Same workspace as chx_sc_convolve.
extern "C" size_t chx_sc_phi_halo_elements(int64_t B, const int32_t* bins) {
    if (B < 1 || !bins) return 0;
    return (size_t)B * (bins[0] + 2 * kHalo) * (bins[1] + 2 * kHalo) * (bins[2] + 2 * kHalo);
}

extern "C" int chx_sc_convolve_halo_after(const void* rho, const void* Ghat, const double* scale, int64_t B, const int32_t* bins,
                                          int dtype, void* phi_halo, void* workspace, size_t workspace_bytes, void* stream,
                                          void* ghat_ready_event) {
    if (!rho || !Ghat || !scale || !phi_halo || B < 1 || B > 65535 || !chx_sc_pruned_supported(bins, dtype))
        return CHX_ERR_INVALID_ARG;
    if (!workspace || workspace_bytes < chx_sc_convolve_workspace_bytes(B, bins, dtype)) return CHX_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    hipEvent_t ev = (hipEvent_t)ghat_ready_event;
    return dtype == CHX_F32 ? convolve_impl<float>((const float*)rho, (const float*)Ghat, scale, B, bins, (float*)phi_halo,
                                                   (float*)workspace, s, true, ev)
                            : convolve_impl<double>((const double*)rho, (const double*)Ghat, scale, B, bins, (double*)phi_halo,
                                                    (double*)workspace, s, true, ev);
}

// chx_sc_convolve_halo_after for a charge grid that is an accumulation buffer (chx_sc_tile_deposit_acc): rho is all zeros when
// the call has run
extern "C" int chx_sc_convolve_halo_consume(void* rho, const void* Ghat, const double* scale, int64_t B, const int32_t* bins, int dtype,
                                            void* phi_halo, void* workspace, size_t workspace_bytes, void* stream,
                                            void* ghat_ready_event) {
    if (!rho || !Ghat || !scale || !phi_halo || B < 1 || B > 65535 || !chx_sc_pruned_supported(bins, dtype))
        return CHX_ERR_INVALID_ARG;
    if (!workspace || workspace_bytes < chx_sc_convolve_workspace_bytes(B, bins, dtype)) return CHX_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    hipEvent_t ev = (hipEvent_t)ghat_ready_event;
    return dtype == CHX_F32 ? convolve_impl<float>((const float*)rho, (const float*)Ghat, scale, B, bins, (float*)phi_halo,
                                                   (float*)workspace, s, true, ev, true)
                            : convolve_impl<double>((const double*)rho, (const double*)Ghat, scale, B, bins, (double*)phi_halo,
                                                    (double*)workspace, s, true, ev, true);
}

// chx_sc_convolve_halo_consume inside a chain kick (chx_sc_tiles.h): B = 1, the first pass carries the deposit's bookkeeping step
int chx_sc_convolve_carries_schedule(const int32_t* bins) { return bins && 2 * bins[0] <= 256 && 2 * bins[0] >= 32; }

int chx_sc_convolve_halo_chain(void* rho, const void* Ghat, const double* scale, const int32_t* bins, int dtype, void* phi_halo,
                               void* workspace, size_t workspace_bytes, void* stream, void* ghat_ready_event,
                               const ScScheduleArgs* schedule) {
    if (!rho || !Ghat || !scale || !phi_halo || !chx_sc_pruned_supported(bins, dtype)) return CHX_ERR_INVALID_ARG;
    if (schedule && !chx_sc_convolve_carries_schedule(bins)) return CHX_ERR_INVALID_ARG;
    if (!workspace || workspace_bytes < chx_sc_convolve_workspace_bytes(1, bins, dtype)) return CHX_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    hipEvent_t ev = (hipEvent_t)ghat_ready_event;
    return dtype == CHX_F32 ? convolve_impl<float>((const float*)rho, (const float*)Ghat, scale, 1, bins, (float*)phi_halo,
                                                   (float*)workspace, s, true, ev, true, schedule)
                            : convolve_impl<double>((const double*)rho, (const double*)Ghat, scale, 1, bins, (double*)phi_halo,
                                                    (double*)workspace, s, true, ev, true, schedule);
}

extern "C" int chx_sc_convolve_halo(const void* rho, const void* Ghat, const double* scale, int64_t B, const int32_t* bins,
                                    int dtype, void* phi_halo, void* workspace, size_t workspace_bytes, void* stream) {
    return chx_sc_convolve_halo_after(rho, Ghat, scale, B, bins, dtype, phi_halo, workspace, workspace_bytes, stream, nullptr);
}

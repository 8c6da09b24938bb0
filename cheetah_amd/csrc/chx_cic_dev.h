// chx_cic_dev.h — device side of the cloud-in-cell arithmetic shared by the deposit kernels (chx_cic.hip) and the particle
// pass of a lattice stretch that deposits a Screen's image from registers (lattice_apply_kernel, chx_apply.hip).
// cloud_in_cell.py:150-311 in the working dtype, operation by operation (both files are compiled with -ffp-contract=off): the
// same cell indices and fractions wherever these functions are used.
#pragma once
#include "chx_common.h"

namespace {

struct CicDev {
    int ndim;
    int cols[3];
    int bins[3];
    int64_t gstride[3];
    int64_t gbatch;
    int64_t B, Bx, Bq, Bs, Be, Bsc, Bsh, N;
    int abs_charge;
};

template <typename T>
struct CicPoint {
    bool inside;
    long long i[3];  // floor(p)
    T f[3];          // p - i
    T bw[3];         // bin width
};

// one axis of cloud_in_cell.py:150-172: in-extent test, bin-space position, floor, fraction (v = the coordinate after scale / shift)
template <typename T>
__device__ __forceinline__ bool cic_axis(T v, T l, T rgt, int bins, long long& i_out, T& f_out, T& bw_out) {
    const bool inside = (v >= l) && (v <= rgt);
    const T bw = (rgt - l) / (T)bins;
    const T pb = (v - l) / bw - (T)0.5;
    T fl = floor(pb);
    // clamp before the integer conversion (only reachable outside the extent,
    // where the charge is masked to zero anyway)
    const T lim = (T)4.0e18;
    fl = fl > lim ? lim : (fl < -lim ? -lim : fl);
    const long long i = (long long)fl;
    i_out = i;
    f_out = pb - (T)i;
    bw_out = bw;
    return inside;
}

// A Screen's extent from its pixel size (screen.py:139-148: `-resolution * pixel_size / 2`, `resolution * pixel_size / 2` as
// torch evaluates them on the device: the integer becomes a scalar of the tensor's dtype, the division by the scalar 2 a
// multiplication by its reciprocal)
template <typename T>
__device__ __forceinline__ void screen_extent_axis(int resolution, T pixel_size, T& left, T& right) {
    left = ((T)(-resolution) * pixel_size) * (T)0.5;
    right = ((T)resolution * pixel_size) * (T)0.5;
}

// cloud_in_cell.py:150-172 (1-D), :262-311 (3-D): in-extent mask, bin-space position, floor, frac
// Rmap (optional): a 7x7 map applied to the particle on the fly — coordinate cols[d] of R x, evaluated as the fma chain of
// chx_apply_affine7 (bit-identical to tracking first and depositing afterwards), without the tracked particles ever
// being written (chx_cic_deposit_mapped: Screen images of a scan of lattice settings).
template <typename T>
__device__ __forceinline__ CicPoint<T> cic_locate(const CicDev& a, const T* __restrict__ x,
                                                  const T* __restrict__ extent,
                                                  const T* __restrict__ scale,
                                                  const T* __restrict__ shift, int64_t b, int64_t n,
                                                  const T* __restrict__ Rmap = nullptr) {
    CicPoint<T> r;
    r.inside = true;
    const int64_t xrow = (a.Bx == 1 ? 0 : b) * a.N + n;
    const T* ext = extent + (a.Be == 1 ? 0 : b) * a.ndim * 2;
    T row[7];
    if (Rmap) {
#pragma unroll
        for (int j = 0; j < 7; ++j) row[j] = x[xrow * 7 + j];
    }
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        if (d < a.ndim) {
            T v;
            if (Rmap) {
                const T* Rr = Rmap + a.cols[d] * 7;
                v = Rr[0] * row[0];
#pragma unroll
                for (int j = 1; j < 7; ++j) v = fma(Rr[j], row[j], v);
            } else {
                v = x[xrow * 7 + a.cols[d]];
            }
            if (scale) v = v * scale[(a.Bsc == 1 ? 0 : b) * a.ndim + d];
            if (shift) v = v - shift[(a.Bsh == 1 ? 0 : b) * a.ndim + d];
            r.inside = cic_axis<T>(v, ext[d * 2], ext[d * 2 + 1], a.bins[d], r.i[d], r.f[d], r.bw[d]) && r.inside;
        } else {
            r.i[d] = 0;
            r.f[d] = (T)0;
            r.bw[d] = (T)1;
        }
    }
    return r;
}

template <typename T>
__device__ __forceinline__ T cic_charge(const CicDev& a, const T* __restrict__ q,
                                        const T* __restrict__ s, int64_t b, int64_t n) {
    T c = q ? q[(a.Bq == 1 ? 0 : b) * a.N + n] : (T)1;
    if (a.abs_charge) c = fabs(c);
    if (s) c = c * s[(a.Bs == 1 ? 0 : b) * a.N + n];
    return c;
}

// Per-workgroup combining table in LDS: a focused beam puts thousands of particles on a handful of cells, and
// same-address global atomics serialise in L2 (measured: 159 us for 1e4 particles of the ARES example on its screen,
// 3-4 ms for 1e6 particles focused to half a pixel). Contributions are first summed per cell in an open-addressed LDS
// table (fp64 values: ds_add_f64 is the fast LDS atomic on gfx950), one global atomic per occupied slot at the end; a
// cell that finds no slot within kCombProbes goes to global memory directly, so a diffuse beam loses nothing.
constexpr int kCombSlots = 2048;  // 32 KiB of LDS per workgroup
constexpr int kCombProbes = 4;

template <typename T>
struct CombTable {
    long long* keys;
    double* vals;
    T* g;
    __device__ __forceinline__ void init(long long* k, double* v, T* grid) {
        keys = k; vals = v; g = grid;
        for (int i = threadIdx.x; i < kCombSlots; i += blockDim.x) { keys[i] = -1; vals[i] = 0.0; }
        __syncthreads();
    }
    __device__ __forceinline__ void add(int64_t off, T v) {
        unsigned h = (unsigned)((unsigned long long)off * 0x9E3779B97F4A7C15ull >> 40) & (kCombSlots - 1);
#pragma unroll
        for (int probe = 0; probe < kCombProbes; ++probe) {
            const long long prev = (long long)atomicCAS(reinterpret_cast<unsigned long long*>(&keys[h]),
                                                        (unsigned long long)-1LL, (unsigned long long)off);
            if (prev == -1LL || prev == (long long)off) {
                unsafeAtomicAdd(&vals[h], (double)v);
                return;
            }
            h = (h + 1) & (kCombSlots - 1);
        }
        unsafeAtomicAdd(g + off, v);
    }
    __device__ __forceinline__ void flush() {
        __syncthreads();
        for (int i = threadIdx.x; i < kCombSlots; i += blockDim.x)
            if (keys[i] != -1) unsafeAtomicAdd(g + keys[i], (T)vals[i]);
    }
};

// ---- Screen histogram (screen.py:305-311; ATen histogramdd with explicit edges:
// skip if v < e_0 or e_last < v; pos = upper_bound(edges, v) - 1; pos == nbins -> nbins - 1) ----
template <typename T>
__device__ __forceinline__ int hist_bin(const T* __restrict__ edges, int nbins, T v) {
    if (!(v >= edges[0]) || !(v <= edges[nbins])) return -1;
    int lo = 0, hi = nbins + 1;  // first index with edges[idx] > v
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (edges[mid] > v) hi = mid;
        else lo = mid + 1;
    }
    int pos = lo - 1;
    if (pos == nbins) pos -= 1;
    return pos;
}

}  // namespace

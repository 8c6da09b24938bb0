// chx_sc_geom_dev.h — the grid geometry of a chain kick formed INSIDE the kernels that need it first (space_charge_kick.py:531-550).
//
// A kick of a chain (chx_sc_kick_sorted) builds its grid from the beam sizes of the rows the previous kick's gather pass wrote. With
// one partial sum per gather workgroup (~4000 of them) a one-workgroup kernel has to reduce them between the gather and the two
// kernels that need the geometry first — the tile deposit on the main stream, the Green function's corner table on the side stream:
// a launch on BOTH streams (8.5 us each behind a 6.5 us write-back gap, the side stream's also behind a cross-queue hop), all of it
// on the critical path of the kick (profiles/r04_c4_critical_path.md).
//
// Here the gather pass adds its workgroups' sums into kScSumRows rows of 8 doubles with fp64 atomics (workgroup b -> row b mod 256:
// fifteen adds per address, spread over 2048 addresses), and EVERY workgroup of the two consumer launches reduces those 16 KB itself
// — one round of loads, a workgroup sum, the arithmetic of sc_geometry_row on four lanes: ~4 us at the head of each workgroup, no
// launch, no flag, nothing to wait for. Workgroup 0 also stores the result as the stream's copy of the geometry for the kernels that
// follow on that stream. Both launches compute the same numbers from the same sums: identical copies.
// (First tried: workgroup 0 reduces the ~4000 partial sums and publishes the geometry through agent-scope stores and a flag, the
// other workgroups of the launch poll the flag. Correct, but the chain reduce -> arithmetic on one lane -> publish -> poll -> load is
// ~15 us long — as long as the launch it replaces: C4 1.97 ms either way.)
// The order of the atomic adds is not fixed: the sums differ in the last bits from run to run, like the charge grid the deposit's
// float atomics build. Two sets of rows alternate with the kick's index; the gather pass that fills one clears the other.
#pragma once
#include "chx_common.h"
#include "chx_sc_math.h"

constexpr int kScGeoValues = 17;        // half 3 | cell 3 | gamma | dt | scale 3 | extent 6   (of the beam dtype T)
constexpr int kScGeoPotOffset = 192;    // byte offset of the potential factor (double) inside a geometry copy
constexpr int kScGeoBytes = 256;        // one copy
constexpr int kScSumRows = 256;         // rows of sums[8][kScSumRows]: W, W2, s_x, s_y, s_tau, m_xx, m_yy, m_tautau about the origin

struct ScGeoSums {
    const double* sums;       // [8][kScSumRows] of the previous gather pass; nullptr: the geometry is in place already
    const void* grid_extent;  // T[3]
    const void* energy;       // T[1]
    const void* length;       // T[1]
    double mass, pot_factor;
    int gx, gy, gz;
    void* geo_out;            // where workgroup 0 of the launch leaves the copy for the kernels behind it
};

// all TH >= 256 threads of a workgroup. red: (TH / 16 + 1) * 8 doubles of LDS; geo_s[kScGeoValues], pot_s[1]: LDS, valid on return
// (the function ends with a barrier). Bit for bit the arithmetic of sc_geometry_row, the three axes and the energy terms on a lane each.
template <typename T, int TH>
__device__ __forceinline__ void sc_geo_from_sums(const ScGeoSums& r, double* red, T* geo_s, double* pot_s, bool store) {
    constexpr int K = 8;
    static_assert(TH >= kScSumRows && TH % 64 == 0, "one row per thread");
    double* tot = red + (TH / 16) * K;
    double a[K];
#pragma unroll
    for (int k = 0; k < K; ++k) a[k] = threadIdx.x < kScSumRows ? r.sums[k * kScSumRows + threadIdx.x] : 0.0;
#pragma unroll
    for (int k = 0; k < K; ++k) a[k] = chx_row16_sum(a[k]);
    if ((threadIdx.x & 15) == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) red[(threadIdx.x >> 4) * K + k] = a[k];
    }
    __syncthreads();
    if (threadIdx.x < K) {
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < kScSumRows / 16; ++q) t += red[q * K + threadIdx.x];
        tot[threadIdx.x] = t;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int d = threadIdx.x;
        const double W = tot[0], W2 = tot[1];
        const double cf = W - W2 / W;
        const double m = tot[2 + d] / W;
        const double var = (tot[5 + d] - W * m * m) / cf;
        const T g = (T)(d == 0 ? r.gx : (d == 1 ? r.gy : r.gz));
        const T sig = (T)sqrt(var);
        const T h = ((const T*)r.grid_extent)[d] * sig;
        const T c = ((T)2 * h) / g;
        geo_s[d] = h;
        geo_s[3 + d] = c;
        geo_s[11 + d * 2] = -h;
        geo_s[12 + d * 2] = h;
    } else if (threadIdx.x == 3) {
        const T gam = ((const T*)r.energy)[0] / (T)r.mass;
        const T ig2 = (T)1 / (gam * gam);
        T one_minus = (T)1 - ig2;
        if (one_minus < (T)0) one_minus = (T)0;
        const T beta = (fabs((double)gam) > 0.0) ? (T)sqrt(one_minus) : (T)1;
        geo_s[6] = gam;
        geo_s[7] = ((const T*)r.length)[0] / ((T)299792458.0 * beta);
        geo_s[8] = (T)1;
        geo_s[9] = (T)1;
        geo_s[10] = -beta;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double vol = 1.0;
        for (int d = 0; d < 3; ++d) vol *= (double)geo_s[3 + d];
        const double pot = (1.0 / vol) * r.pot_factor;
        pot_s[0] = pot;
        if (store) *(double*)((char*)r.geo_out + kScGeoPotOffset) = pot;
    }
    if (store && threadIdx.x < kScGeoValues) ((T*)r.geo_out)[threadIdx.x] = geo_s[threadIdx.x];
    __syncthreads();
}

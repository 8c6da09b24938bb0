// chx_common.h — shared device/host helpers for the libchx HIP kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "chx.h"

#define CHX_WAVE 64
#define CHX_BLOCK 256

#define CHX_CHECK_LAUNCH()                                   \
    do {                                                     \
        hipError_t e__ = hipGetLastError();                  \
        if (e__ != hipSuccess) return CHX_ERR_LAUNCH;        \
    } while (0)

static inline bool chx_bcast_ok(int64_t b, int64_t B) { return b == 1 || b == B; }
static __host__ __device__ inline bool chx_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// 16-byte vector type per element type: float -> 4 lanes, double -> 2 lanes.
template <typename T> struct chx_vec16;
template <> struct chx_vec16<float> { using type = float4; static constexpr int n = 4; };
template <> struct chx_vec16<double> { using type = double2; static constexpr int n = 2; };

// v + (v of the lane selected by the DPP control CTRL): a VALU move with a data-parallel-primitive lane pattern — no LDS
// round trip, unlike __shfl_xor (ds_bpermute_b32 + s_waitcnt per 32-bit half).
template <int CTRL>
__device__ __forceinline__ double chx_dpp_add(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int lo2 = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
    const int hi2 = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
    return v + __hiloint2double(hi2, lo2);
}

// Sum over the 64 lanes of a wavefront (all lanes receive the total). The four steps inside a row of 16 lanes are DPP
// moves (quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror: after each step the lanes being paired
// already hold equal partial sums, so the mirrors act like xor 4 / xor 8); only the two steps across rows go through
// ds_bpermute. Measured motive (moments_onepass_kernel, 29 accumulators): 348 dependent ds_bpermute + s_waitcnt per wave
// were half of the kernel's 19.6 us.
__device__ __forceinline__ double chx_wave_sum(double v) {
    v = chx_dpp_add<0xB1>(v);   // quad_perm [1,0,3,2]
    v = chx_dpp_add<0x4E>(v);   // quad_perm [2,3,0,1]
    v = chx_dpp_add<0x141>(v);  // row_half_mirror
    v = chx_dpp_add<0x140>(v);  // row_mirror
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

// The same sum, bit for bit, delivered in lane 63 ONLY (the other lanes hold partial sums): the two steps across rows as DPP
// row broadcasts (row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3) instead of four dependent ds_bpermute —
// for passes that take many wave sums per wave and let one lane write them (the monitors of lattice_apply_kernel).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double chx_dpp_add_rows(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int lo2 = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, false);
    const int hi2 = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, false);
    return v + __hiloint2double(hi2, lo2);
}
__device__ __forceinline__ double chx_wave_sum_lane63(double v) {
    v = chx_dpp_add<0xB1>(v);
    v = chx_dpp_add<0x4E>(v);
    v = chx_dpp_add<0x141>(v);
    v = chx_dpp_add<0x140>(v);
    v = chx_dpp_add_rows<0x142, 0xA>(v);   // row_bcast:15 -> rows 1 and 3: r1 + r0, r3 + r2
    v = chx_dpp_add_rows<0x143, 0xC>(v);   // row_bcast:31 -> rows 2 and 3: (r3 + r2) + (r1 + r0)
    return v;
}

// Sum over each row of 16 lanes only (every lane of a row receives its row's total): the DPP part of chx_wave_sum.
__device__ __forceinline__ double chx_row16_sum(double v) {
    v = chx_dpp_add<0xB1>(v);
    v = chx_dpp_add<0x4E>(v);
    v = chx_dpp_add<0x141>(v);
    v = chx_dpp_add<0x140>(v);
    return v;
}

// Block-wide sum of K doubles per thread (CHX_BLOCK threads = 4 waves). Result valid in thread 0.
template <int K>
__device__ __forceinline__ void chx_block_sum(double (&v)[K], double* smem /* [4*K] */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = chx_wave_sum(v[k]);
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) smem[wave * K + k] = v[k];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k)
            v[k] = smem[k] + smem[K + k] + smem[2 * K + k] + smem[3 * K + k];
    }
    __syncthreads();
}

// Block-wide sums of EIGHT doubles per thread, folded: instead of eight full wave sums (8 x (8 DPP moves + 4 ds_bpermute + 6
// adds) per wave) the lanes first split the eight values between them — after exchanging with lane ^ 1 a lane carries four of
// them (summed over the pair), after lane ^ 2 two (summed over its quad) — and only those two are summed over the four quads of
// the row of 16 (row_ror:8, row_ror:4). Lane p < 4 of every row then holds the row's totals of values {0,1}, {4,5}, {2,3}, {6,7}
// (p = 0..3) and leaves them in LDS; after one barrier threads k < 8 add the 16 rows of the workgroup: v[0] of thread k is the
// total of value k. ~60 instructions per wave instead of ~200. smem: 16 rows x 8 doubles.
template <int CTRL>
__device__ __forceinline__ double chx_dpp_get(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int lo2 = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
    const int hi2 = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi2, lo2);
}
__device__ __forceinline__ void chx_block_sum8_folded(double (&v)[8], double* smem /* [16 * 8] */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool b0 = lane & 1, b1 = lane & 2;
    double t[4], u[2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const double mine = b0 ? v[j + 4] : v[j], give = b0 ? v[j] : v[j + 4];
        t[j] = mine + chx_dpp_get<0xB1>(give);          // quad_perm [1,0,3,2]
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const double mine = b1 ? t[j + 2] : t[j], give = b1 ? t[j] : t[j + 2];
        u[j] = mine + chx_dpp_get<0x4E>(give);          // quad_perm [2,3,0,1]
        u[j] += chx_dpp_get<0x128>(u[j]);               // row_ror:8
        u[j] += chx_dpp_get<0x124>(u[j]);               // row_ror:4
    }
    if ((lane & 15) < 4) {
        const int row = wave * 4 + (lane >> 4);
        const int first = 4 * (lane & 1) + (lane & 2);  // value index of u[0]
        smem[row * 8 + first] = u[0];
        smem[row * 8 + first + 1] = u[1];
    }
    __syncthreads();
    if (threadIdx.x < 8) {
        double acc = 0.0;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc += smem[r * 8 + threadIdx.x];
        v[0] = acc;
    }
}

// THIRTY-TWO doubles per thread summed over the workgroup's rows of 16 lanes, folded the same way: after lane ^ 1 a lane carries 16
// of the values, after lane ^ 2 eight, and those eight are summed over the four quads of its row (row_ror:8, row_ror:4): ~216
// instructions per wave against 32 x 12 for full row sums. Lane p < 4 of a row ends with the row totals of values
// 16 (p & 1) + 8 (p >> 1) + j, j < 8, and stores them to smem[row][32] (row = wave * 4 + lane / 16). No barrier in here.
__device__ __forceinline__ void chx_row16_sum32_folded(double (&v)[32], double* smem /* [rows][32] */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool b0 = lane & 1, b1 = lane & 2;
    double t[16], u[8];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const double mine = b0 ? v[j + 16] : v[j], give = b0 ? v[j] : v[j + 16];
        t[j] = mine + chx_dpp_get<0xB1>(give);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const double mine = b1 ? t[j + 8] : t[j], give = b1 ? t[j] : t[j + 8];
        u[j] = mine + chx_dpp_get<0x4E>(give);
        u[j] += chx_dpp_get<0x128>(u[j]);
        u[j] += chx_dpp_get<0x124>(u[j]);
    }
    if ((lane & 15) < 4) {
        double* dst = smem + (wave * 4 + (lane >> 4)) * 32 + 16 * (lane & 1) + 4 * (lane & 2);
#pragma unroll
        for (int j = 0; j < 8; ++j) dst[j] = u[j];
    }
}

// SIXTEEN doubles per thread, the same folding (~108 instructions per wave): lane p < 4 of a row ends with the row totals of values
// 8 (p & 1) + 4 (p >> 1) + j, j < 4, and stores them to smem[row * stride + j...]. No barrier in here.
__device__ __forceinline__ void chx_row16_sum16_folded(double (&v)[16], double* smem /* [rows][stride] */, int stride) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool b0 = lane & 1, b1 = lane & 2;
    double t[8], u[4];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const double mine = b0 ? v[j + 8] : v[j], give = b0 ? v[j] : v[j + 8];
        t[j] = mine + chx_dpp_get<0xB1>(give);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const double mine = b1 ? t[j + 4] : t[j], give = b1 ? t[j] : t[j + 4];
        u[j] = mine + chx_dpp_get<0x4E>(give);
        u[j] += chx_dpp_get<0x128>(u[j]);
        u[j] += chx_dpp_get<0x124>(u[j]);
    }
    if ((lane & 15) < 4) {
        double* dst = smem + (wave * 4 + (lane >> 4)) * stride + 8 * (lane & 1) + 2 * (lane & 2);
#pragma unroll
        for (int j = 0; j < 4; ++j) dst[j] = u[j];
    }
}

// ---- LDS tile staging: contiguous 16-byte vector transfers between global memory and an LDS
// tile (coalesced global_load/store_dwordx4); scalar fallback when the tile start is unaligned.
// Non-temporal 16-byte accesses (the `nt` cache policy of global_load/store_dwordx4): a streaming pass reads every
// byte once and writes every byte once, so nothing is gained by allocating the lines in L2. Measured on MI355X
// (benchmarks/apply_variants.hip, two buffers ping-ponged like a tracked lattice): 5.69 -> 6.71 TB/s at 1e6 particles,
// 5.44 -> 5.90 TB/s at 1.6e7. Not used where other workgroups re-read the same input (a beam shared by a batch).
typedef float chx_v2f __attribute__((ext_vector_type(2)));
typedef float chx_v4f __attribute__((ext_vector_type(4)));
typedef double chx_v2d __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float4 chx_nt_load(const float4* p) {
    const chx_v4f v = __builtin_nontemporal_load(reinterpret_cast<const chx_v4f*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ double2 chx_nt_load(const double2* p) {
    const chx_v2d v = __builtin_nontemporal_load(reinterpret_cast<const chx_v2d*>(p));
    return make_double2(v.x, v.y);
}
__device__ __forceinline__ void chx_nt_store(float4 v, float4* p) {
    const chx_v4f w = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(w, reinterpret_cast<chx_v4f*>(p));
}
__device__ __forceinline__ void chx_nt_store(double2 v, double2* p) {
    const chx_v2d w = {v.x, v.y};
    __builtin_nontemporal_store(w, reinterpret_cast<chx_v2d*>(p));
}

// TP = particles per tile (compile time, for documentation of the call sites). The copy loops are deliberately left
// rolled: issuing all ceil(TP * 7 / (VN * 256)) = 4 loads of a lane before the first wait was measured SLOWER on MI355X
// (apply at 1e6 particles: 8.4 -> 9.4 us repeated, 9.3 -> 11.3 us ping-ponged) — with 8 workgroups per CU the memory
// system is already oversubscribed and the extra 16 VGPRs cost occupancy.
template <typename T, int TP>
__device__ __forceinline__ void tile_load(const T* __restrict__ g, T* __restrict__ lds, int n_elem,
                                          bool vec_ok, bool nt = false) {
    using V = typename chx_vec16<T>::type;
    constexpr int VN = chx_vec16<T>::n;
    if (vec_ok) {
        const int nvec = n_elem / VN;
        const V* __restrict__ gv = reinterpret_cast<const V*>(g);
        V* lv = reinterpret_cast<V*>(lds);
        if (nt) {
            for (int v = threadIdx.x; v < nvec; v += CHX_BLOCK) lv[v] = chx_nt_load(gv + v);
        } else {
            for (int v = threadIdx.x; v < nvec; v += CHX_BLOCK) lv[v] = gv[v];
        }
        for (int e = nvec * VN + threadIdx.x; e < n_elem; e += CHX_BLOCK) lds[e] = g[e];
    } else {
        for (int e = threadIdx.x; e < n_elem; e += CHX_BLOCK) lds[e] = g[e];
    }
}

template <typename T, int TP>
__device__ __forceinline__ void tile_store(T* __restrict__ g, const T* __restrict__ lds, int n_elem,
                                           bool vec_ok, bool nt = false) {
    using V = typename chx_vec16<T>::type;
    constexpr int VN = chx_vec16<T>::n;
    if (vec_ok) {
        const int nvec = n_elem / VN;
        V* __restrict__ gv = reinterpret_cast<V*>(g);
        const V* lv = reinterpret_cast<const V*>(lds);
        if (nt) {
            for (int v = threadIdx.x; v < nvec; v += CHX_BLOCK) chx_nt_store(lv[v], gv + v);
        } else {
            for (int v = threadIdx.x; v < nvec; v += CHX_BLOCK) gv[v] = lv[v];
        }
        for (int e = nvec * VN + threadIdx.x; e < n_elem; e += CHX_BLOCK) g[e] = lds[e];
    } else {
        for (int e = threadIdx.x; e < n_elem; e += CHX_BLOCK) g[e] = lds[e];
    }
}

// ---- the same staging per WAVE: a wave moves its own rows between global memory and its own LDS slice (64 rows of 28 / 56
// bytes are a whole number of 16-byte chunks), so no workgroup barrier is needed around the transfer — chx_wave_sync()
// orders the wave's LDS accesses — and the waves of a workgroup run independently.
__device__ __forceinline__ void chx_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <typename T>
__device__ __forceinline__ void wave_tile_load(const T* __restrict__ g, T* __restrict__ wl, int n_elem, bool vec_ok,
                                                bool nt = false) {
    using V = typename chx_vec16<T>::type;
    constexpr int VN = chx_vec16<T>::n;
    const int lane = threadIdx.x & 63;
    if (vec_ok) {
        const int nvec = n_elem / VN;
        const V* __restrict__ gv = reinterpret_cast<const V*>(g);
        V* lv = reinterpret_cast<V*>(wl);
        if (nt) {
            for (int v = lane; v < nvec; v += 64) lv[v] = chx_nt_load(gv + v);
        } else {
            for (int v = lane; v < nvec; v += 64) lv[v] = gv[v];
        }
        for (int e = nvec * VN + lane; e < n_elem; e += 64) wl[e] = g[e];
    } else {
        for (int e = lane; e < n_elem; e += 64) wl[e] = g[e];
    }
}

template <typename T>
__device__ __forceinline__ void wave_tile_store(T* __restrict__ g, const T* __restrict__ wl, int n_elem, bool vec_ok,
                                                 bool nt = false) {
    using V = typename chx_vec16<T>::type;
    constexpr int VN = chx_vec16<T>::n;
    const int lane = threadIdx.x & 63;
    if (vec_ok) {
        const int nvec = n_elem / VN;
        V* __restrict__ gv = reinterpret_cast<V*>(g);
        const V* lv = reinterpret_cast<const V*>(wl);
        if (nt) {
            for (int v = lane; v < nvec; v += 64) chx_nt_store(lv[v], gv + v);
        } else {
            for (int v = lane; v < nvec; v += 64) gv[v] = lv[v];
        }
        for (int e = nvec * VN + lane; e < n_elem; e += 64) g[e] = wl[e];
    } else {
        for (int e = lane; e < n_elem; e += 64) g[e] = wl[e];
    }
}

static inline int chx_grid_for(int64_t work_items, int per_block, int cap) {
    int64_t g = (work_items + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

// chx_moments_dev.h — device pieces of the moment kernels that other translation units inline (chx_build.hip: the builders' VJP
// that forms the cotangent of a run's map from the gradient of ONE beam property itself, chx_run_vjp_entry).
#pragma once
#include <hip/hip_runtime.h>

#include "chx.h"
#include "chx_common.h"

// dR (49 values) of one batch row from the gradient g[29] of its outgoing moments, its map Rb (T) and the incoming moments
// m[29], by ONE WAVE: lane 6 i + j (i, j < 6) owns entry (i, j) of the 6x6 products, which pass through LDS (a single thread
// doing the two dense products took 14 us). g, m: anywhere readable by every lane; lds: 3 * 36 doubles of this wave.
template <typename T, typename TO>
__device__ __forceinline__ void mapped_bwd_row_wave(const double* g, const T* __restrict__ Rb, const double* __restrict__ m,
                                                    double* lds, TO* __restrict__ o) {
    const int lane = threadIdx.x & 63;
    double* G = lds;
    double* AC = lds + 36;
    double* C = lds + 72;
    const int i = lane / 6, j = lane - 6 * i;
    if (lane < 36) {
        // index of (min, max) in the upper-triangle listing that starts at 8
        const int lo = i < j ? i : j, hi = i < j ? j : i;
        const int k = 8 + lo * 6 - (lo * (lo - 1)) / 2 + (hi - lo);
        G[lane] = (i == j) ? g[k] : 0.5 * g[k];
        C[lane] = m[k];
    }
    chx_wave_sync();
    if (lane < 36) {
        double acc = 0.0;
#pragma unroll
        for (int l = 0; l < 6; ++l) acc += (double)Rb[i * 7 + l] * C[l * 6 + j];
        AC[lane] = acc;
    }
    chx_wave_sync();
    if (lane < 36) {
        double acc = 0.0;
#pragma unroll
        for (int l = 0; l < 6; ++l) acc += G[i * 6 + l] * AC[l * 6 + j];
        o[i * 7 + j] = (TO)(2.0 * acc + g[2 + i] * m[2 + j]);
    } else if (lane < 42) {
        o[(lane - 36) * 7 + 6] = (TO)g[2 + (lane - 36)];
    } else if (lane < 49) {
        o[42 + (lane - 42)] = (TO)0;
    }
}

// the gradient vector g[29] of chx_moments' outputs that ONE entry (or its square root) of them receives (chx_moments_entry): lanes
// 0..28 of the calling wave write it; infinite at a zero variance like torch.sqrt's own backward
__device__ __forceinline__ void moment_entry_gradient(double grad, const double* __restrict__ mom_y, int index, int take_sqrt, double* g) {
    const int lane = threadIdx.x & 63;
    if (lane < CHX_MOM_NOUT) {
        double gv = 0.0;
        if (lane == index) {
            gv = grad;
            if (take_sqrt) gv = gv * 0.5 / sqrt(mom_y[index]);
        }
        g[lane] = gv;
    }
    chx_wave_sync();
}

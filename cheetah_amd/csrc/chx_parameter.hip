// chx_parameter.hip — ParameterBeam path: a beam described by its mean vector and
// covariance matrix is tracked as  mu' = R mu,  cov' = R cov R^T  (cheetah/accelerator/element.py:167-179),
// the cavity applies the reference's moment updates (cavity.py:127-133, 202-218), and a Screen reads the beam
// as a bivariate normal density on the pixel grid (screen.py:255-291). All of it is O(49) work per batch row:
// these kernels exist so that a vectorised ParameterBeam scan (thousands of lattice settings, the usual RL
// workload) stays on the device without a chain of tiny torch ops per element.
#include "chx_common.h"

namespace {

// one wavefront per batch row; lane (i, j) owns one entry of the 7x7 result
template <typename T>
__global__ __launch_bounds__(64) void parameter_track_kernel(const T* __restrict__ mu, const T* __restrict__ cov,
                                                            const T* __restrict__ R,
                                                            const double* __restrict__ coeffs, int64_t Bmu,
                                                            int64_t Bcov, int64_t BR, T* __restrict__ mu_out,
                                                            T* __restrict__ cov_out) {
    __shared__ double r[49], c[49], m[7], tmp[49];
    const int64_t b = blockIdx.x;
    const int lane = threadIdx.x;
    const int i = lane / 7, j = lane - 7 * i;
    if (lane < 49) {
        r[lane] = (double)R[(BR == 1 ? 0 : b) * 49 + lane];
        c[lane] = (double)cov[(Bcov == 1 ? 0 : b) * 49 + lane];
    }
    if (lane < 7) m[lane] = (double)mu[(Bmu == 1 ? 0 : b) * 7 + lane];
    __syncthreads();
    if (lane < 49) {  // tmp = R cov
        double s = 0.0;
        for (int k = 0; k < 7; ++k) s = fma(r[i * 7 + k], c[k * 7 + j], s);
        tmp[lane] = s;
    }
    __syncthreads();
    double cov_ij = 0.0, mu_i = 0.0;
    if (lane < 49) {  // cov' = tmp R^T
        for (int k = 0; k < 7; ++k) cov_ij = fma(tmp[i * 7 + k], r[j * 7 + k], cov_ij);
    }
    if (lane < 7) {
        for (int k = 0; k < 7; ++k) mu_i = fma(r[lane * 7 + k], m[k], mu_i);
    }
    if (coeffs) {
        // cavity moment updates with the INCOMING moments (cavity.py:127-133, 202-218);
        // cf = [a, b, k*beta0, phi, cos phi, T566, T556, T555]
        const double* cf = coeffs + b * CHX_CAV_NCOEF;
        const double mu4 = m[4], mu5 = m[5], c44 = c[4 * 7 + 4], c45 = c[4 * 7 + 5], c55 = c[5 * 7 + 5];
        if (lane == 5) mu_i = mu5 * cf[0] + cf[1] * (cos(-mu4 * cf[2] + cf[3]) - cf[4]);
        if (lane == 4) mu_i = mu_i + (cf[5] * mu5 * mu5 + cf[6] * mu4 * mu5 + cf[7] * mu4 * mu4);
        const double q = cf[5] * c55 * c55 + cf[6] * c45 * c55 + cf[7] * c44 * c44;
        if (lane == 5 * 7 + 5) cov_ij = c55;
        if (lane == 4 * 7 + 4 || lane == 4 * 7 + 5 || lane == 5 * 7 + 4) cov_ij = q;
    }
    if (lane < 49) cov_out[b * 49 + lane] = (T)cov_ij;
    if (lane < 7) mu_out[b * 7 + lane] = (T)mu_i;
}

// backward of mu' = R mu, cov' = R cov R^T for one batch row per wavefront (lane (i, j) owns one entry): with g = dL/dmu',
// G = dL/dcov':  d_mu = R^T g,  d_cov = R^T G R,  d_R = g mu^T + G R cov^T + G^T R cov   (fp64 inside, like the forward)
template <typename T>
__global__ __launch_bounds__(64) void parameter_track_bwd_kernel(const T* __restrict__ g_mu, const T* __restrict__ g_cov,
                                                                const T* __restrict__ mu, const T* __restrict__ cov,
                                                                const T* __restrict__ R, int64_t Bmu, int64_t Bcov, int64_t BR,
                                                                T* __restrict__ d_mu, T* __restrict__ d_cov, T* __restrict__ d_R) {
    __shared__ double r[49], c[49], m[7], G[49], g[7], rc[49], rct[49], gr[49];
    const int64_t b = blockIdx.x;
    const int lane = threadIdx.x;
    const int i = lane / 7, j = lane - 7 * i;
    if (lane < 49) {
        r[lane] = (double)R[(BR == 1 ? 0 : b) * 49 + lane];
        c[lane] = (double)cov[(Bcov == 1 ? 0 : b) * 49 + lane];
        G[lane] = g_cov ? (double)g_cov[b * 49 + lane] : 0.0;
    }
    if (lane < 7) {
        m[lane] = (double)mu[(Bmu == 1 ? 0 : b) * 7 + lane];
        g[lane] = g_mu ? (double)g_mu[b * 7 + lane] : 0.0;
    }
    __syncthreads();
    if (lane < 49) {
        double s1 = 0.0, s2 = 0.0, s3 = 0.0;
        for (int k = 0; k < 7; ++k) {
            s1 = fma(r[i * 7 + k], c[k * 7 + j], s1);      // R cov
            s2 = fma(r[i * 7 + k], c[j * 7 + k], s2);      // R cov^T
            s3 = fma(G[i * 7 + k], r[k * 7 + j], s3);      // G R
        }
        rc[lane] = s1;
        rct[lane] = s2;
        gr[lane] = s3;
    }
    __syncthreads();
    if (lane < 49) {
        if (d_R) {
            double s = g[i] * m[j];
            for (int k = 0; k < 7; ++k) s = fma(G[i * 7 + k], rct[k * 7 + j], fma(G[k * 7 + i], rc[k * 7 + j], s));
            d_R[b * 49 + lane] = (T)s;
        }
        if (d_cov) {
            double s = 0.0;
            for (int k = 0; k < 7; ++k) s = fma(r[k * 7 + i], gr[k * 7 + j], s);
            d_cov[b * 49 + lane] = (T)s;
        }
    }
    if (lane < 7 && d_mu) {
        double s = 0.0;
        for (int k = 0; k < 7; ++k) s = fma(r[k * 7 + lane], g[k], s);
        d_mu[b * 7 + lane] = (T)s;
    }
}

// image[b][iy][ix] = N2(pos; mu_xy, cov_xy) at pos = (left + ix hstep, bottom + iy vstep)  (screen.py:277-291)
template <typename T>
__global__ __launch_bounds__(CHX_BLOCK) void screen_gaussian_kernel(const T* __restrict__ mu, const T* __restrict__ cov,
                                                                   const T* __restrict__ shift,
                                                                   const T* __restrict__ geom /*[left,hstep,bottom,vstep]*/,
                                                                   int64_t Bmu, int64_t Bcov, int64_t Bsh, int W,
                                                                   int H, int pos_f32, T* __restrict__ image) {
    const int64_t b = blockIdx.y;
    const T* m = mu + (Bmu == 1 ? 0 : b) * 7;
    const T* c = cov + (Bcov == 1 ? 0 : b) * 49;
    const double sx = shift ? (double)shift[(Bsh == 1 ? 0 : b) * 2] : 0.0;
    const double sy = shift ? (double)shift[(Bsh == 1 ? 0 : b) * 2 + 1] : 0.0;
    const double mx = (double)m[0] - sx, my = (double)m[2] - sy;
    const double cxx = (double)c[0], cxy = (double)c[2], cyy = (double)c[2 * 7 + 2];
    const double det = cxx * cyy - cxy * cxy;
    const double norm = 1.0 / (2.0 * 3.14159265358979323846 * sqrt(det));
    const double left = (double)geom[0], hstep = (double)geom[1], bottom = (double)geom[2], vstep = (double)geom[3];
    const int64_t npx = (int64_t)W * H;
    for (int64_t p = (int64_t)blockIdx.x * CHX_BLOCK + threadIdx.x; p < npx; p += (int64_t)gridDim.x * CHX_BLOCK) {
        const int iy = (int)(p / W), ix = (int)(p - (int64_t)iy * W);
        // NB: the reference builds the pixel grid with a dtype-less torch.arange, i.e. in torch's DEFAULT dtype
        // (screen.py:284-287): with a float32 default its sample positions carry ~1e-11 m of fp32 jitter even
        // for an fp64 screen (and ATen's vectorised arange is not round-to-nearest, so it cannot be reproduced
        // portably). With pos_f32 the positions are rounded to fp32 (nearest) to stay within that jitter class;
        // tests compare images of beams that are resolved by the pixel grid, where the jitter is harmless.
        double px = left + ix * hstep, py = bottom + iy * vstep;
        if (pos_f32) { px = (double)(float)px; py = (double)(float)py; }
        const double dx = px - mx, dy = py - my;
        const double quad = (cyy * dx * dx - 2.0 * cxy * dx * dy + cxx * dy * dy) / det;
        image[b * npx + p] = (T)(norm * exp(-0.5 * quad));
    }
}

// A ParameterBeam through a whole stretch of lattice — [run | active Cavity | active BPM]+ as prepared by lattice_prepare_kernel
// (chx_build.hip: one map per item, a coefficient row per cavity) — one wavefront per batch row walks the items: per item the
// arithmetic of parameter_track_kernel on moments rounded to T in between (what storing the beam behind every element and
// loading it again gives: bit-identical to the walk item by item); a monitor reads (mu_x, mu_y) - misalignment there.
struct ParameterScreens {
    void* mu[CHX_LATTICE_MAX_SCREENS];
    void* cov[CHX_LATTICE_MAX_SCREENS];
    const void* q[CHX_LATTICE_MAX_SCREENS];
    void* q_out[CHX_LATTICE_MAX_SCREENS];
    int n;
};

// STAGED (stretches of at most kParamStagedItems items): the wave first brings the item table, every item's map, the cavities' coefficient
// rows and the monitors' misalignments into LDS — one round of loads side by side — and then walks the items out of LDS. Read where they
// are used, the item's type and its map are two dependent memory round trips in front of every item of a walk that is sequential by
// nature (the moments are rounded to T behind every item). Measured over bench.py's diagnostics lattices (216 launches): 30.1 -> 26.9 us
// on average — what is left is the walk itself: four LDS exchanges and two 7-term fp64 FMA chains per item.
constexpr int kParamStagedItems = 96;

template <typename T, bool STAGED>
__global__ __launch_bounds__(64) void parameter_lattice_kernel(const T* __restrict__ mu, const T* __restrict__ cov, int64_t Bmu,
                                                              int64_t Bcov, int64_t Bm /*rows of lattice settings: 1 or B*/,
                                                              const int64_t* __restrict__ items_g, int n_items,
                                                              const double* __restrict__ Rs, const double* __restrict__ coeffs,
                                                              const int64_t* __restrict__ ptrs, T* __restrict__ mu_out,
                                                              T* __restrict__ cov_out, T* __restrict__ readings, ParameterScreens scr) {
    __shared__ double r[49], c[49], m[7], tmp[49];
    extern __shared__ __attribute__((aligned(8))) unsigned char stage_raw[];
    const int64_t b = blockIdx.x, B = gridDim.x;
    const int lane = threadIdx.x;
    const int i = lane / 7, j = lane - 7 * i;
    // staged copies: items[n_items][4] | coefficient rows [n_items][8] (double) | maps [n_items][49] (T) | misalignments [n_items][2] (T)
    int64_t* items_s = reinterpret_cast<int64_t*>(stage_raw);
    double* coeffs_s = reinterpret_cast<double*>(items_s + (STAGED ? n_items * 4 : 0));
    T* maps_s = reinterpret_cast<T*>(coeffs_s + (STAGED ? n_items * CHX_CAV_NCOEF : 0));
    T* mis_s = maps_s + (STAGED ? n_items * 49 : 0);
    const int64_t* items = STAGED ? items_s : items_g;
    if constexpr (STAGED) {
        for (int w = lane; w < n_items * 4; w += 64) items_s[w] = items_g[w];
        for (int w = lane; w < n_items * 49; w += 64) {
            const int it = w / 49, q = w - it * 49;
            maps_s[w] = reinterpret_cast<const T*>(Rs + ((int64_t)it * Bm + (Bm == 1 ? 0 : b)) * 49)[q];
        }
        for (int w = lane; w < n_items * CHX_CAV_NCOEF; w += 64) {
            const int it = w / CHX_CAV_NCOEF, q = w - it * CHX_CAV_NCOEF;
            coeffs_s[w] = coeffs[((int64_t)it * Bm + (Bm == 1 ? 0 : b)) * CHX_CAV_NCOEF + q];
        }
        for (int it = lane; it < n_items; it += 64) {
            if (items_g[it * 4] == 2) {
                const T* mis = (const T*)ptrs[items_g[it * 4 + 2]];
                mis_s[it * 2] = mis[0];
                mis_s[it * 2 + 1] = mis[1];
            }
        }
    }
    if (lane < 49) c[lane] = (double)cov[(Bcov == 1 ? 0 : b) * 49 + lane];
    if (lane < 7) m[lane] = (double)mu[(Bmu == 1 ? 0 : b) * 7 + lane];
    __syncthreads();
    for (int it = 0; it < n_items; ++it) {
        const int type = (int)items[it * 4];
        if (type == 2) {
            if (lane < 2) {
                T mis_v;
                if constexpr (STAGED) mis_v = mis_s[it * 2 + lane];
                else mis_v = ((const T*)ptrs[items[it * 4 + 2]])[lane];
                readings[((int64_t)items[it * 4 + 3] * B + b) * 2 + lane] = (T)m[lane == 0 ? 0 : 2] - mis_v;
            }
            continue;
        }
        if (type == 4) {
            // an active screen (screen.py:187-214): the moments that reach it are its record (unshifted; the image kernel and
            // the caller's get_read_beam subtract the misalignment)
            const int slot = (int)items[it * 4 + 3];
            if (slot < scr.n && b == 0) {
                if (scr.mu[slot] && lane < 7) ((T*)scr.mu[slot])[lane] = (T)m[lane];
                if (scr.cov[slot] && lane < 49) ((T*)scr.cov[slot])[lane] = (T)c[lane];
                if (scr.q[slot] && scr.q_out[slot] && lane == 63) *(T*)scr.q_out[slot] = *(const T*)scr.q[slot];
            }
            continue;
        }
        if (type != 0 && type != 1) continue;
        const int64_t mrow = (int64_t)it * Bm + (Bm == 1 ? 0 : b);         // this row's map of the item (vectorised settings)
        if (lane < 49) {
            if constexpr (STAGED) r[lane] = (double)maps_s[it * 49 + lane];
            else r[lane] = (double)reinterpret_cast<const T*>(Rs + mrow * 49)[lane];
        }
        __syncthreads();
        if (lane < 49) {  // tmp = R cov
            double s = 0.0;
            for (int k = 0; k < 7; ++k) s = fma(r[i * 7 + k], c[k * 7 + j], s);
            tmp[lane] = s;
        }
        __syncthreads();
        double cov_ij = 0.0, mu_i = 0.0;
        if (lane < 49) {  // cov' = tmp R^T
            for (int k = 0; k < 7; ++k) cov_ij = fma(tmp[i * 7 + k], r[j * 7 + k], cov_ij);
        }
        if (lane < 7) {
            for (int k = 0; k < 7; ++k) mu_i = fma(r[lane * 7 + k], m[k], mu_i);
        }
        if (type == 1) {      // cavity moment updates with the INCOMING moments (cavity.py:127-133, 202-218)
            const double* cf = STAGED ? coeffs_s + it * CHX_CAV_NCOEF : coeffs + mrow * CHX_CAV_NCOEF;
            const double mu4 = m[4], mu5 = m[5], c44 = c[4 * 7 + 4], c45 = c[4 * 7 + 5], c55 = c[5 * 7 + 5];
            if (lane == 5) mu_i = mu5 * cf[0] + cf[1] * (cos(-mu4 * cf[2] + cf[3]) - cf[4]);
            if (lane == 4) mu_i = mu_i + (cf[5] * mu5 * mu5 + cf[6] * mu4 * mu5 + cf[7] * mu4 * mu4);
            const double q = cf[5] * c55 * c55 + cf[6] * c45 * c55 + cf[7] * c44 * c44;
            if (lane == 5 * 7 + 5) cov_ij = c55;
            if (lane == 4 * 7 + 4 || lane == 4 * 7 + 5 || lane == 5 * 7 + 4) cov_ij = q;
        }
        __syncthreads();
        if (lane < 49) c[lane] = (double)(T)cov_ij;
        if (lane < 7) m[lane] = (double)(T)mu_i;
        __syncthreads();
    }
    if (lane < 49) cov_out[b * 49 + lane] = (T)c[lane];
    if (lane < 7) mu_out[b * 7 + lane] = (T)m[lane];
}

}  // namespace

extern "C" int chx_parameter_track(const void* mu, const void* cov, const void* R, const double* cavity_coeffs,
                                   int64_t B, int64_t Bmu, int64_t Bcov, int64_t BR, int dtype, void* mu_out,
                                   void* cov_out, void* stream) {
    if (!mu || !cov || !R || !mu_out || !cov_out || B < 1 || B > 0x7fffffffLL) return CHX_ERR_INVALID_ARG;
    if (!chx_bcast_ok(Bmu, B) || !chx_bcast_ok(Bcov, B) || !chx_bcast_ok(BR, B)) return CHX_ERR_INVALID_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == CHX_F32)
        hipLaunchKernelGGL(parameter_track_kernel<float>, dim3((unsigned)B), dim3(64), 0, s, (const float*)mu,
                           (const float*)cov, (const float*)R, cavity_coeffs, Bmu, Bcov, BR, (float*)mu_out,
                           (float*)cov_out);
    else if (dtype == CHX_F64)
        hipLaunchKernelGGL(parameter_track_kernel<double>, dim3((unsigned)B), dim3(64), 0, s, (const double*)mu,
                           (const double*)cov, (const double*)R, cavity_coeffs, Bmu, Bcov, BR, (double*)mu_out,
                           (double*)cov_out);
    else
        return CHX_ERR_DTYPE;
    CHX_CHECK_LAUNCH();
    return CHX_OK;
}

extern "C" int chx_parameter_lattice_track(const int64_t* table, int64_t n_items, int64_t n_elems, int64_t n_ptrs, const void* energy,
                                           double mass_eV, double n_charges, int dtype, void* state, size_t state_bytes,
                                           const void* mu, const void* cov, int64_t B, int64_t Bmu, int64_t Bcov, int64_t Bm,
                                           int small_runs, void* mu_out, void* cov_out, void* energy_out, const void* s_in,
                                           void* s_out, int64_t n_bpm, void* readings, void* stream) {
    return chx_parameter_lattice_track_screens(table, n_items, n_elems, n_ptrs, energy, mass_eV, n_charges, dtype, state, state_bytes, mu,
                                               cov, B, Bmu, Bcov, Bm, small_runs, mu_out, cov_out, energy_out, s_in, s_out, n_bpm,
                                               readings, nullptr, 0, stream);
}

extern "C" int chx_parameter_lattice_track_screens(const int64_t* table, int64_t n_items, int64_t n_elems, int64_t n_ptrs,
                                                   const void* energy, double mass_eV, double n_charges, int dtype, void* state,
                                                   size_t state_bytes, const void* mu, const void* cov, int64_t B, int64_t Bmu,
                                                   int64_t Bcov, int64_t Bm, int small_runs, void* mu_out, void* cov_out,
                                                   void* energy_out, const void* s_in, void* s_out, int64_t n_bpm, void* readings,
                                                   const chx_lattice_screen* screens, int64_t n_screens, void* stream) {
    if (!mu || !cov || !mu_out || !cov_out || B < 1 || B > 0x7fffffffLL || n_bpm < 0 || (n_bpm > 0 && !readings))
        return CHX_ERR_INVALID_ARG;
    if (!chx_bcast_ok(Bmu, B) || !chx_bcast_ok(Bcov, B) || !chx_bcast_ok(Bm, B)) return CHX_ERR_INVALID_ARG;
    if (n_screens < 0 || n_screens > CHX_LATTICE_MAX_SCREENS || (n_screens > 0 && (!screens || B != 1))) return CHX_ERR_INVALID_ARG;
    ParameterScreens scr;
    scr.n = (int)n_screens;
    chx_lattice_screen prep[CHX_LATTICE_MAX_SCREENS];
    for (int k = 0; k < CHX_LATTICE_MAX_SCREENS; ++k) {
        scr.mu[k] = k < n_screens ? screens[k].mu : nullptr;
        scr.cov[k] = k < n_screens ? screens[k].cov : nullptr;
        scr.q[k] = k < n_screens ? screens[k].total_charge : nullptr;
        scr.q_out[k] = k < n_screens ? screens[k].total_charge_out : nullptr;
        if (k < n_screens) {
            prep[k] = screens[k];
            prep[k].image = nullptr;           // (the gaussian image writes every pixel: nothing to zero)
            prep[k].image_bytes = 0;
            if (screens[k].image && (!screens[k].mu || !screens[k].cov || !screens[k].geom)) return CHX_ERR_INVALID_ARG;
        }
    }
    int st = chx_lattice_prepare_screens(table, n_items, n_elems, n_ptrs, Bm, small_runs, energy, mass_eV, n_charges, dtype, state,
                                         state_bytes, energy_out, s_in, s_out, n_screens > 0 ? prep : nullptr, n_screens, stream);
    if (st != CHX_OK) return st;
    const double* Rs = (const double*)state;
    const double* coeffs = Rs + n_items * Bm * 49;
    const int64_t* ptrs = table + n_items * 4 + 2 * n_elems;
    hipStream_t s = (hipStream_t)stream;
    // (CHX_TUNE_PARAMETER_STAGED=0: every item's table row and map read where they are used)
    static const bool stage_on = [] { const char* e = getenv("CHX_TUNE_PARAMETER_STAGED"); return !(e && e[0] == '0'); }();
    const size_t esz = dtype == CHX_F32 ? 4 : 8;
    const bool staged = stage_on && n_items <= kParamStagedItems;
    const size_t stage_bytes = staged ? (size_t)n_items * (4 * 8 + CHX_CAV_NCOEF * 8 + 49 * esz + 2 * esz) : 0;
#define CHX_PARAM_LAUNCH(T, ST)                                                                                                        \
    hipLaunchKernelGGL((parameter_lattice_kernel<T, ST>), dim3((unsigned)B), dim3(64), stage_bytes, s, (const T*)mu, (const T*)cov, Bmu, \
                       Bcov, Bm, table, (int)n_items, Rs, coeffs, ptrs, (T*)mu_out, (T*)cov_out, (T*)readings, scr)
    if (dtype == CHX_F32) {
        if (staged) CHX_PARAM_LAUNCH(float, true);
        else CHX_PARAM_LAUNCH(float, false);
    } else {
        if (staged) CHX_PARAM_LAUNCH(double, true);
        else CHX_PARAM_LAUNCH(double, false);
    }
#undef CHX_PARAM_LAUNCH
    CHX_CHECK_LAUNCH();
    for (int64_t k = 0; k < n_screens; ++k) {
        if (!screens[k].image) continue;
        st = chx_screen_gaussian(screens[k].mu, screens[k].cov, screens[k].shift, screens[k].geom, 1, 1, 1, 1, screens[k].width,
                                 screens[k].height, 0, dtype, screens[k].image, stream);
        if (st != CHX_OK) return st;
    }
    return CHX_OK;
}

extern "C" int chx_parameter_track_bwd(const void* g_mu, const void* g_cov, const void* mu, const void* cov, const void* R, int64_t B,
                                       int64_t Bmu, int64_t Bcov, int64_t BR, int dtype, void* d_mu, void* d_cov, void* d_R,
                                       void* stream) {
    if ((!g_mu && !g_cov) || !mu || !cov || !R || (!d_mu && !d_cov && !d_R) || B < 1 || B > 0x7fffffffLL) return CHX_ERR_INVALID_ARG;
    if (!chx_bcast_ok(Bmu, B) || !chx_bcast_ok(Bcov, B) || !chx_bcast_ok(BR, B)) return CHX_ERR_INVALID_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == CHX_F32)
        hipLaunchKernelGGL(parameter_track_bwd_kernel<float>, dim3((unsigned)B), dim3(64), 0, s, (const float*)g_mu, (const float*)g_cov,
                           (const float*)mu, (const float*)cov, (const float*)R, Bmu, Bcov, BR, (float*)d_mu, (float*)d_cov, (float*)d_R);
    else if (dtype == CHX_F64)
        hipLaunchKernelGGL(parameter_track_bwd_kernel<double>, dim3((unsigned)B), dim3(64), 0, s, (const double*)g_mu,
                           (const double*)g_cov, (const double*)mu, (const double*)cov, (const double*)R, Bmu, Bcov, BR,
                           (double*)d_mu, (double*)d_cov, (double*)d_R);
    else
        return CHX_ERR_DTYPE;
    CHX_CHECK_LAUNCH();
    return CHX_OK;
}

extern "C" int chx_screen_gaussian(const void* mu, const void* cov, const void* shift, const void* geom, int64_t B,
                                   int64_t Bmu, int64_t Bcov, int64_t Bsh, int32_t width, int32_t height,
                                   int positions_fp32, int dtype, void* image, void* stream) {
    if (!mu || !cov || !geom || !image || B < 1 || B > 65535 || width < 1 || height < 1) return CHX_ERR_INVALID_ARG;
    if (!chx_bcast_ok(Bmu, B) || !chx_bcast_ok(Bcov, B) || (shift && !chx_bcast_ok(Bsh, B))) return CHX_ERR_INVALID_ARG;
    hipStream_t s = (hipStream_t)stream;
    int64_t g = ((int64_t)width * height + CHX_BLOCK - 1) / CHX_BLOCK;
    if (g > 4096) g = 4096;
    dim3 grid((unsigned)g, (unsigned)B);
    if (dtype == CHX_F32)
        hipLaunchKernelGGL(screen_gaussian_kernel<float>, grid, dim3(CHX_BLOCK), 0, s, (const float*)mu, (const float*)cov,
                           (const float*)shift, (const float*)geom, Bmu, Bcov, Bsh, width, height, positions_fp32, (float*)image);
    else if (dtype == CHX_F64)
        hipLaunchKernelGGL(screen_gaussian_kernel<double>, grid, dim3(CHX_BLOCK), 0, s, (const double*)mu, (const double*)cov,
                           (const double*)shift, (const double*)geom, Bmu, Bcov, Bsh, width, height, positions_fp32, (double*)image);
    else
        return CHX_ERR_DTYPE;
    CHX_CHECK_LAUNCH();
    return CHX_OK;
}

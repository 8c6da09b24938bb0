// chx_spacecharge.hip — grid and particle kernels of SpaceChargeKick.
//
// Replaces cheetah/accelerator/space_charge_kick.py:103-586 and the SI coordinate conversions
// cheetah/particles/particle_beam.py:1262-1346. The 3-D FFTs stay in hipFFT (called through
// torch.fft by the host layer); everything else is here:
//   * chx_sc_igf        integrated Green function on the doubled (Hockney) grid, fp64 inside.
//                       The reference evaluates the 6-transcendental primitive F eight times per
//                       cell (space_charge_kick.py:195-236); neighbouring cells share corner points,
//                       so F is tabulated once on the (g+1)^3 corner lattice and G is the signed
//                       8-term difference of table entries (8x fewer asinh/atan), then mirrored into
//                       the other seven octants (space_charge_kick.py:249-289).
//   * chx_sc_spectral_mul   rho_hat *= G_hat * scale (space_charge_kick.py:313-316)
//   * chx_sc_gradient   central differences of the cropped potential, x(-1/gamma^2), packed as
//                       float4 (Fx,Fy,Fz,0) per node so the gather is one 16-byte load per corner
//   * chx_sc_gather_kick    to_xyz_pxpypz -> node-based trilinear gather -> p += F dt ->
//                       from_xyz_pxpypz fused per particle in registers (fp64 inside: the SI
//                       momenta ~1e-20 kg m/s square to denormals in fp32).
#include <hipfft/hipfft.h>

#include <new>

#include "chx_common.h"
#include "chx_sc_math.h"
#include "chx_sc_tiles.h"

namespace {

// table[b][i][j][k] = F((i-1/2) dx, (j-1/2) dy, (k-1/2) dt), i in [0, gx] etc.
template <typename T>
__global__ __launch_bounds__(CHX_BLOCK) void igf_table_kernel(const T* __restrict__ cell,
                                                             const T* __restrict__ gamma, int gx,
                                                             int gy, int gz,
                                                             double* __restrict__ table) {
    const int64_t b = blockIdx.y;
    const int64_t npts = (int64_t)(gx + 1) * (gy + 1) * (gz + 1);
    const double dx = (double)cell[b * 3 + 0], dy = (double)cell[b * 3 + 1];
    // longitudinal cell scaled by gamma (space_charge_kick.py:170-176); the product is formed in
    // the working dtype like the reference's `cell_size[..., 2] * beam.relativistic_gamma`
    const double dt = (double)(T)(cell[b * 3 + 2] * gamma[b]);
    // 32-bit index arithmetic ((g + 1)^3 < 2^31): 64-bit integer division costs ~100 instructions here
    for (unsigned idx = blockIdx.x * CHX_BLOCK + threadIdx.x; idx < (unsigned)npts; idx += gridDim.x * CHX_BLOCK) {
        const unsigned q1 = idx / (unsigned)(gz + 1), q2 = q1 / (unsigned)(gy + 1);
        const int k = (int)(idx - q1 * (unsigned)(gz + 1)), j = (int)(q1 - q2 * (unsigned)(gy + 1)), i = (int)q2;
        table[b * npts + idx] = igf_primitive<double>((i - 0.5) * dx, (j - 0.5) * dy, (k - 0.5) * dt);
    }
}

template <typename T>
__global__ __launch_bounds__(CHX_BLOCK) void igf_fill_kernel(const double* __restrict__ table, int gx,
                                                            int gy, int gz, int64_t ldz, T* __restrict__ G) {
    const int64_t b = blockIdx.y;
    const int64_t ncell = (int64_t)gx * gy * gz;
    const int64_t npts = (int64_t)(gx + 1) * (gy + 1) * (gz + 1);
    const double* tb = table + b * npts;
    const int64_t sy = gz + 1, sx = (int64_t)(gy + 1) * (gz + 1);
    // logical size (2gx, 2gy, 2gz); rows of the last axis are `ldz` apart (2gz, or 2gz + 2 for in-place R2C)
    const int64_t GX = 2 * gx, GY = 2 * gy, GZ = ldz;
    const int64_t GZlog = 2 * gz;
    T* Gb = G + b * GX * GY * GZ;
    for (unsigned idx = blockIdx.x * CHX_BLOCK + threadIdx.x; idx < (unsigned)ncell; idx += gridDim.x * CHX_BLOCK) {
        const unsigned q1 = idx / (unsigned)gz, q2 = q1 / (unsigned)gy;
        const int k = (int)(idx - q1 * (unsigned)gz), j = (int)(q1 - q2 * (unsigned)gy), i = (int)q2;
        const double* p = tb + i * sx + j * sy + k;
        // +F(+,+,+) -F(-,+,+) -F(+,-,+) -F(+,+,-) +F(+,-,-) +F(-,+,-) +F(-,-,+) -F(-,-,-)
        const double g = p[sx + sy + 1] - p[sy + 1] - p[sx + 1] - p[sx + sy] + p[sx] + p[sy] + p[1] - p[0];
        const T gv = (T)g;
        const int64_t i2 = GX - i, j2 = GY - j, k2 = GZlog - k;
        Gb[((int64_t)i * GY + j) * GZ + k] = gv;
        if (i > 0) Gb[(i2 * GY + j) * GZ + k] = gv;
        if (j > 0) Gb[((int64_t)i * GY + j2) * GZ + k] = gv;
        if (k > 0) Gb[((int64_t)i * GY + j) * GZ + k2] = gv;
        if (i > 0 && j > 0) Gb[(i2 * GY + j2) * GZ + k] = gv;
        if (j > 0 && k > 0) Gb[((int64_t)i * GY + j2) * GZ + k2] = gv;
        if (i > 0 && k > 0) Gb[(i2 * GY + j) * GZ + k2] = gv;
        if (i > 0 && j > 0 && k > 0) Gb[(i2 * GY + j2) * GZ + k2] = gv;
    }
}

template <typename T>
__global__ __launch_bounds__(CHX_BLOCK) void spectral_mul_kernel(T* __restrict__ a /*complex*/,
                                                                const T* __restrict__ g,
                                                                const double* __restrict__ scale,
                                                                int64_t n) {
    const int64_t b = blockIdx.y;
    const T sc = (T)scale[b];
    T* ab = a + b * n * 2;
    const T* gb = g + b * n * 2;
    for (int64_t i = (int64_t)blockIdx.x * CHX_BLOCK + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * CHX_BLOCK) {
        const T ar = ab[2 * i], ai = ab[2 * i + 1], gr = gb[2 * i], gi = gb[2 * i + 1];
        ab[2 * i] = (ar * gr - ai * gi) * sc;
        ab[2 * i + 1] = (ar * gi + ai * gr) * sc;
    }
}

// space_charge_kick.py:324-365
template <typename T>
__global__ __launch_bounds__(CHX_BLOCK) void gradient_kernel(const T* __restrict__ phi,
                                                            const T* __restrict__ cell,
                                                            const T* __restrict__ gamma, int gx,
                                                            int gy, int gz, int doubled, int64_t ldz,
                                                            T* __restrict__ F) {
    const int64_t b = blockIdx.y;
    // phi is either the doubled Hockney array [2gx][2gy][ldz] (cropped on the fly; ldz = 2gz, or 2gz + 2 when the
    // inverse FFT ran in place) or already compact [gx][gy][gz]
    const int64_t GY = doubled ? 2 * gy : gy, GZ = doubled ? ldz : gz;
    const T* pb = phi + b * (int64_t)(doubled ? 2 * gx : gx) * GY * GZ;
    const int64_t ncell = (int64_t)gx * gy * gz;
    const T gm = gamma[b];
    const T ig2 = (gm != (T)0) ? (T)1 / (gm * gm) : (T)0;
    const T hx = (T)0.5 * ((T)1 / cell[b * 3 + 0]);
    const T hy = (T)0.5 * ((T)1 / cell[b * 3 + 1]);
    const T hz = (T)0.5 * ((T)1 / cell[b * 3 + 2]);
    for (unsigned idx = blockIdx.x * CHX_BLOCK + threadIdx.x; idx < (unsigned)ncell; idx += gridDim.x * CHX_BLOCK) {
        const unsigned q1 = idx / (unsigned)gz, q2 = q1 / (unsigned)gy;
        const int k = (int)(idx - q1 * (unsigned)gz), j = (int)(q1 - q2 * (unsigned)gy), i = (int)q2;
        const int64_t c = ((int64_t)i * GY + j) * GZ + k;
        T fx = (T)0, fy = (T)0, fz = (T)0;
        if (i > 0 && i < gx - 1) fx = (pb[c + GY * GZ] - pb[c - GY * GZ]) * hx;
        if (j > 0 && j < gy - 1) fy = (pb[c + GZ] - pb[c - GZ]) * hy;
        if (k > 0 && k < gz - 1) fz = (pb[c + 1] - pb[c - 1]) * hz;
        T* o = F + (b * ncell + idx) * 4;
        o[0] = -ig2 * fx;
        o[1] = -ig2 * fy;
        o[2] = -ig2 * fz;
        o[3] = (T)0;
    }
}

// Force components at node (I, J, K) from the potential with a halo (chx_sc_convolve_halo): exactly gradient_kernel's
// arithmetic, evaluated where it is needed. `q` points at the node, `lo`/`hi` are the neighbours along the axis.
template <typename T>
__device__ __forceinline__ T node_force(T lo, T hi, T h, T neg_ig2, bool interior) {
    const T f = interior ? (hi - lo) * h : (T)0;
    return neg_ig2 * f;
}

template <typename T, int N>
struct alignas(sizeof(T)) Run { T v[N]; };   // N consecutive grid values at element (not vector) alignment

// ---- the particle step of a float32 beam from the potential, in float32 (round 6) ------------------------------------------
// space_charge_kick.py:387-475 (node-based trilinear gather of the force), :548-565 (p += F dt between to_xyz_pxpypz and
// from_xyz_pxpypz, particle_beam.py:1262-1346). Until round 5 a float32 row went through the float64 step below (SI momenta,
// 11 divisions and 5 square roots in fp64: 534 of the pass's ~1300 instructions per wave, the gather pass VALU-bound at 38 us per
// 1e6 rows). The SI detour is not needed for the accuracy: in units of m c, with g = gamma0 (1 + delta beta0) the particle's
// gamma and pn = (px, py) gamma0 beta0, pz = sqrt(g^2 - 1 - pn^2), the kick k = F e dt / (m c) gives
//     px' = px + kx / (gamma0 beta0)                     (one rounding: the sum itself)
//     g'^2 = g^2 + D,   D = 2 (pn . k) + k . k
//     delta' = (g' - gamma0) / (gamma0 beta0) = delta + D / ((g' + g) gamma0 beta0)
// — the same real-number map as the reference's, with the cancellation g' - gamma0 removed algebraically instead of carried in
// fp64: every term is a float32 product with a relative error of ~1e-7 OF THE KICK, and x, y, tau pass through untouched
// (the fp64 step returns them bit for bit as well). The potential is float32 to begin with. Measured against the reference's
// float64 run: profiles/r06_c4_gather.md. Float64 beams keep the fp64 step (sc_kick_locate / sc_kick_finish below).
struct ScKickCtx32 {
    float gamma0, p0n, rp0n, nbeta;      // gamma0, gamma0 beta0, its reciprocal, -beta0
    float r_hi[3], r_lo[3];              // 1 / cell as a sum of two floats (the cell coordinate keeps ~1e-7 of a cell, see below)
    float h_hi[3], h_lo[3];              // half / cell likewise
    float k[3];                          // (sum of weights x potential differences along d) -> kick in units of m c
    int g[3];
};

// No division in here: a wave holds 64 rows, so what a lane does once per kick is paid per row (seven float32 divisions were ~80 of
// the pass's instructions). v_rcp_f32 / v_sqrt_f32 are good to 1 ulp, which is what the terms they enter need.
__device__ __forceinline__ ScKickCtx32 sc_kick_ctx32(const float* __restrict__ half, const float* __restrict__ cell,
                                                     const float* __restrict__ energy, const float* __restrict__ dt,
                                                     const float* __restrict__ gamma, float inv_mass, float c_over_m, int gx, int gy,
                                                     int gz) {
    ScKickCtx32 c;
    const float g0 = energy[0] * inv_mass;                                // beam.py:323-326
    c.gamma0 = g0;
    c.p0n = __builtin_amdgcn_sqrtf((g0 - 1.0f) * (g0 + 1.0f));            // gamma0 beta0 = sqrt(gamma0^2 - 1), beam.py:328-336
    c.rp0n = __builtin_amdgcn_rcpf(c.p0n);
    c.nbeta = (fabsf(g0) > 0.0f) ? -(c.p0n * __builtin_amdgcn_rcpf(g0)) : -1.0f;
    const float gm = gamma[0];
    const float nig2 = -((gm != 0.0f) ? __builtin_amdgcn_rcpf(gm * gm) : 0.0f);   // space_charge_kick.py:367-372
    const float kmom = dt[0] * c_over_m;                                  // e dt / (m c) = dt c / (m c^2 / e)
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const float cl = cell[d], hf = half[d];
        const float r = __builtin_amdgcn_rcpf(cl);
        c.r_hi[d] = r;
        c.r_lo[d] = fmaf(-cl, r, 1.0f) * r;                               // 1 / cell - r to first order
        c.h_hi[d] = hf * r;
        c.h_lo[d] = fmaf(hf, r, -c.h_hi[d]) + hf * c.r_lo[d];
        c.k[d] = (nig2 * (0.5f * r)) * kmom;                              // -(1 / gamma^2) / (2 h) x e dt / (m c)
    }
    c.g[0] = gx; c.g[1] = gy; c.g[2] = gz;
    return c;
}

// The row step in three pieces (locate, the three sums, the kick); phi points at the first element of the potential's array (node (0, 0, 0) at
// [2][2][2] inside its halo of 2), py / pz are the array's x / y strides; the array has less than 2^30 elements (32-bit offsets from
// a wave-uniform base: scalar base + vector offset addressing)
// a row's cell: fraction f, floor i0 (clamped to +-2e9), node cn clamped into the halo array, `finite` false for a grid without extent
__device__ __forceinline__ void sc_row32_locate(const ScKickCtx32& c, const float (&v)[7], float (&f)[3], int (&i0)[3], int (&cn)[3],
                                                bool& finite) {
    const float pos[3] = {v[0], v[2], v[4] * c.nbeta};
    finite = true;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        // cell coordinate u = (pos + half) / cell (space_charge_kick.py:405-411). A float32 u near the far end of a 128-node axis
        // resolves 8e-6 of a cell; its fraction f = u - floor(u) is therefore formed again around the integer with the low parts
        // of 1 / cell and half / cell (two more FMAs per axis) and keeps ~1e-7 of a cell. floor(u) itself may land on the wrong
        // side of a node for a row within 1e-5 cells of it: f is then just below 0 or above 1 and the trilinear form, which is
        // continuous across nodes, extrapolates by that much.
        const float u = fmaf(pos[d], c.r_hi[d], c.h_hi[d]);
        finite = finite && (fabsf(u) <= 3.0e38f);                         // false for inf and NaN (a grid without extent)
        float fl = floorf(u);
        f[d] = fmaf(pos[d], c.r_hi[d], c.h_hi[d] - fl) + fmaf(pos[d], c.r_lo[d], c.h_lo[d]);
        fl = fl > 2.0e9f ? 2.0e9f : (fl < -2.0e9f ? -2.0e9f : fl);
        i0[d] = (int)fl;
        cn[d] = min(max(i0[d], -1), c.g[d] - 1);
    }
}

// sums over the cell's eight nodes of weight x central difference of the potential along x / y / z, the potential read from its
// global array (phi: first element; node (0, 0, 0) at [2][2][2] inside its halo of 2; less than 2^30 elements)
__device__ __forceinline__ void sc_row32_sums_global(const ScKickCtx32& c, const float (&f)[3], const int (&i0)[3], const int (&cn)[3],
                                                     const float* __restrict__ phi, int py, int pz, float& sx, float& sy, float& sz,
                                                     int diag) {
    // the 32 potential values around the cell (see phi_cell_forces): z runs of 4 on the four central rows, z pairs one step out
    unsigned o = 4u * (unsigned)((cn[0] + 2) * py + (cn[1] + 2) * pz + (cn[2] + 2));   // byte offset of the cell's first node
    if (diag & 1) o = 4u * (unsigned)(2 * py + 2 * pz + 2);   // (timing experiments only: every lane reads the same cell)
    const char* __restrict__ pb = reinterpret_cast<const char*>(phi);
    Run<float, 4> zr[2][2];
    Run<float, 2> xr[2][2], yr[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            zr[a][b] = *reinterpret_cast<const Run<float, 4>*>(pb + (o + 4u * (unsigned)(a * py + b * pz - 1)));
            xr[a][b] = *reinterpret_cast<const Run<float, 2>*>(pb + (o + 4u * (unsigned)((a ? 2 : -1) * py + b * pz)));
            yr[a][b] = *reinterpret_cast<const Run<float, 2>*>(pb + (o + 4u * (unsigned)(b * py + (a ? 2 : -1) * pz)));
        }
    }
    // per axis and corner: is the node inside the grid (its weight counts), is it an interior node (its central difference
    // along that axis exists; boundary nodes carry force 0, space_charge_kick.py:340-365)
    bool in[3][2], inner[3][2];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int n = i0[d] + a;
            in[d][a] = n >= 0 && n < c.g[d];
            inner[d][a] = n > 0 && n < c.g[d] - 1;
        }
    }
    const float wx[2] = {1.0f - f[0], f[0]}, wy[2] = {1.0f - f[1], f[1]}, wz[2] = {1.0f - f[2], f[2]};
    sx = sy = sz = 0.0f;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const float wab = wx[a] * wy[b];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                // differences along x / y / z at node (cn0 + a, cn1 + b, cn2 + e); a value outside the grid or a difference that
                // reaches into the halo (whose content is undefined, possibly NaN) is replaced by 0, not multiplied by 0
                const float xlo = a ? zr[0][b].v[1 + e] : xr[0][b].v[e];
                const float xhi = a ? xr[1][b].v[e] : zr[1][b].v[1 + e];
                const float ylo = b ? zr[a][0].v[1 + e] : yr[0][a].v[e];
                const float yhi = b ? yr[1][a].v[e] : zr[a][1].v[1 + e];
                const float zlo = zr[a][b].v[e], zhi = zr[a][b].v[e + 2];
                const bool node_in = in[0][a] && in[1][b] && in[2][e];
                const float w = wab * wz[e];
                sx = fmaf(w, (node_in && inner[0][a]) ? xhi - xlo : 0.0f, sx);
                sy = fmaf(w, (node_in && inner[1][b]) ? yhi - ylo : 0.0f, sy);
                sz = fmaf(w, (node_in && inner[2][e]) ? zhi - zlo : 0.0f, sz);
            }
        }
    }
}

// the kick from the three sums: see the header of this section
__device__ __forceinline__ void sc_row32_finish(const ScKickCtx32& c, const float (&v)[7], float sx, float sy, float sz, bool finite,
                                                float (&out)[7]) {
    float kx = sx * c.k[0], ky = sy * c.k[1], kz = sz * c.k[2];
    if (!finite) kx = ky = kz = __builtin_nanf("");
    const float g = fmaf(c.p0n, v[5], c.gamma0);
    const float pxn = v[1] * c.p0n, pyn = v[3] * c.p0n;
    const float pzn = __builtin_amdgcn_sqrtf(fmaf(g - 1.0f, g + 1.0f, -fmaf(pxn, pxn, pyn * pyn)));
    const float D = fmaf(2.0f, fmaf(pxn, kx, fmaf(pyn, ky, pzn * kz)), fmaf(kx, kx, fmaf(ky, ky, kz * kz)));
    const float g1 = __builtin_amdgcn_sqrtf(fmaf(g, g, D));
    out[0] = v[0];
    out[1] = fmaf(kx, c.rp0n, v[1]);
    out[2] = v[2];
    out[3] = fmaf(ky, c.rp0n, v[3]);
    out[4] = v[4];
    out[5] = fmaf(D, __builtin_amdgcn_rcpf((g1 + g) * c.p0n), v[5]);
    out[6] = v[6];
}

// one row: v (Cheetah coordinates) -> out (kicked, Cheetah coordinates), the potential from its global array
__device__ __forceinline__ void sc_kick_row32(const ScKickCtx32& c, const float (&v)[7], const float* __restrict__ phi, int py,
                                              int pz, float (&out)[7], int diag = 0) {
    float f[3], sx, sy, sz;
    int i0[3], cn[3];
    bool finite;
    sc_row32_locate(c, v, f, i0, cn, finite);
    sc_row32_sums_global(c, f, i0, cn, phi, py, pz, sx, sy, sz, diag);
    sc_row32_finish(c, v, sx, sy, sz, finite, out);
}

// the linear run behind a kick on the kicked row, the fma chain of chx_apply_affine7 (bit-identical to a second pass)
__device__ __forceinline__ void sc_post_map32(const float* __restrict__ R, float (&x)[7]) {
    float y[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        float acc = R[i * 7] * x[0];
#pragma unroll
        for (int j = 1; j < 7; ++j) acc = fmaf(R[i * 7 + j], x[j], acc);
        y[i] = acc;
    }
#pragma unroll
    for (int i = 0; i < 7; ++i) x[i] = y[i];
}

// CHX_SC_GATHER_FP64=1: float32 beams take the float64 particle step of rounds 1-5 again (A/B runs, bisecting)
static bool sc_gather_fp64() {
    static const bool on = [] {
        const char* e = getenv("CHX_SC_GATHER_FP64");
        return e && e[0] == '1';
    }();
    return on;
}

// CHX_TUNE_GATHER_DIAG (timing experiments, WRONG results): 1 every lane reads the potential around one cell, 2 no beam-size sums
static int sc_gather_diag() {
    static const int v = [] { const char* e = getenv("CHX_TUNE_GATHER_DIAG"); return e ? atoi(e) : 0; }();
    return v;
}

// sc_kick_row32 addresses the potential with 32-bit element offsets
static bool sc_phi_offsets_fit32(const int32_t* bins) {
    return bins && (int64_t)(bins[0] + 4) * (bins[1] + 4) * (bins[2] + 4) < (1LL << 30);
}

// chx_sc_gather_kick_phi on a float32 beam: the float32 particle step above, rows staged per wave like sc_particle_kernel below;
// inv_mass = (float)(1 / mass_eV), c_over_m = (float)(c / mass_eV) from the host
__global__ __launch_bounds__(CHX_BLOCK) void sc_particle32_kernel(
    const float* __restrict__ x_in, const float* __restrict__ phi, const float* __restrict__ half, const float* __restrict__ cell,
    const float* __restrict__ energy, const float* __restrict__ dt, float inv_mass, float c_over_m, int64_t Bx, int64_t Be, int64_t N,
    int gx, int gy, int gz, float* __restrict__ x_out, const float* __restrict__ post_map, int64_t BR,
    const float* __restrict__ gamma) {
    constexpr int TP = CHX_BLOCK;
    __shared__ __attribute__((aligned(16))) float lds[TP * 7];
    const int64_t b = blockIdx.y;
    const int64_t n0 = (int64_t)blockIdx.x * TP;
    const int np = (int)((N - n0 < TP) ? (N - n0) : TP);
    const int64_t xrow = (Bx == 1) ? 0 : b;
    const bool vin = chx_aligned16(x_in) && (((xrow * N * 7 * (int64_t)sizeof(float)) & 15) == 0);
    const bool vout = chx_aligned16(x_out) && (((b * N * 7 * (int64_t)sizeof(float)) & 15) == 0);
    const int wrow = (threadIdx.x >> 6) * 64;
    const int wvalid = (np - wrow < 0) ? 0 : ((np - wrow < 64) ? (np - wrow) : 64);
    wave_tile_load<float>(x_in + (xrow * N + n0 + wrow) * 7, lds + wrow * 7, wvalid * 7, vin, Bx != 1 || gridDim.y == 1);
    chx_wave_sync();
    const ScKickCtx32 c = sc_kick_ctx32(half + b * 3, cell + b * 3, energy + (Be == 1 ? 0 : b), dt + b, gamma + b, inv_mass, c_over_m,
                                        gx, gy, gz);
    const int pzs = gz + 4, pys = (gy + 4) * pzs;
    const int p = threadIdx.x;
    if (p < np) {
        float v[7], out[7];
#pragma unroll
        for (int j = 0; j < 7; ++j) v[j] = lds[p * 7 + j];
        sc_kick_row32(c, v, phi + b * (int64_t)(gx + 4) * pys, pys, pzs, out);
        if (post_map) sc_post_map32(post_map + ((BR == 1) ? 0 : b) * 49, out);
#pragma unroll
        for (int j = 0; j < 7; ++j) lds[p * 7 + j] = out[j];
    }
    chx_wave_sync();
    wave_tile_store<float>(x_out + (b * N + n0 + wrow) * 7, lds + wrow * 7, wvalid * 7, vout, true);
}

// MODE 0: gather + kick (full SpaceChargeKick particle step); 1: to_xyz only; 2: from_xyz only
// FROM_PHI (MODE 0): F is the potential with a halo instead of the force grid; the central differences are taken here
template <typename T, int MODE, bool FROM_PHI = false>
__global__ __launch_bounds__(CHX_BLOCK) void sc_particle_kernel(
    const T* __restrict__ x_in, const T* __restrict__ F, const T* __restrict__ half,
    const T* __restrict__ cell, const T* __restrict__ energy, const T* __restrict__ dt,
    double mass_eV, int64_t Bx, int64_t Be, int64_t N, int gx, int gy, int gz,
    T* __restrict__ x_out, const T* __restrict__ post_map, int64_t BR, const T* __restrict__ gamma) {
    constexpr int PPT = 1;
    constexpr int TP = PPT * CHX_BLOCK;
    __shared__ __attribute__((aligned(16))) T lds[TP * 7];
    const int64_t b = blockIdx.y;
    const int64_t n0 = (int64_t)blockIdx.x * TP;
    const int np = (int)((N - n0 < TP) ? (N - n0) : TP);
    const int64_t xrow = (Bx == 1) ? 0 : b;
    const bool vin = chx_aligned16(x_in) && (((xrow * N * 7 * (int64_t)sizeof(T)) & 15) == 0);
    const bool vout = chx_aligned16(x_out) && (((b * N * 7 * (int64_t)sizeof(T)) & 15) == 0);
    // staged per wave: no workgroup barrier, the four waves of a workgroup run independently through the long particle step
    const int wrow = (threadIdx.x >> 6) * 64;                       // first row of this wave inside the tile
    const int wvalid = (np - wrow < 0) ? 0 : ((np - wrow < 64) ? (np - wrow) : 64);
    wave_tile_load<T>(x_in + (xrow * N + n0 + wrow) * 7, lds + wrow * 7, wvalid * 7, vin, Bx != 1 || gridDim.y == 1);
    chx_wave_sync();
    const RefFrame<double> rf = ref_frame<double>((double)energy[Be == 1 ? 0 : b], mass_eV);
    const int p = threadIdx.x;
    if (p < np) {
        double v[7], s[7];
#pragma unroll
        for (int j = 0; j < 7; ++j) v[j] = (double)lds[p * 7 + j];
        if (MODE == 2) {
            from_si(rf, v, s);
#pragma unroll
            for (int j = 0; j < 7; ++j) lds[p * 7 + j] = (T)s[j];
        } else {
            to_si(rf, v, s);
            if (MODE == 0) {
                // node-based trilinear gather (space_charge_kick.py:387-473)
                const double pos[3] = {s[0], s[2], s[4]};
                const int g[3] = {gx, gy, gz};
                double u[3];
                int i0[3];
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    u[d] = (pos[d] + (double)half[b * 3 + d]) / (double)cell[b * 3 + d];
                    double fl = floor(u[d]);
                    fl = fl > 2.0e9 ? 2.0e9 : (fl < -2.0e9 ? -2.0e9 : fl);
                    i0[d] = (int)fl;
                }
                // The eight corners are fetched first, each as one 16/32-byte access at a clamped (always valid) node,
                // so all of them are in flight together; corners outside the grid get weight 0.
                struct alignas(4 * sizeof(T)) Node { T x, y, z, pad; };
                Node node[8];
                double w[8];
                if (FROM_PHI) {
                    // 32 potential values around the cell — z runs of 4 on the four central rows, z pairs on the rows one
                    // step out in x and in y — from a grid a quarter of the force grid's size (it stays in L2)
                    const int pz = gz + 4, py = (gy + 4) * pz;   // halo of 2: every index below is in bounds
                    const int ci = min(max(i0[0], -1), g[0] - 1), cj = min(max(i0[1], -1), g[1] - 1),
                              ck = min(max(i0[2], -1), g[2] - 1);
                    const T* q = F + b * (int64_t)(gx + 4) * py + ((int64_t)(ci + 2) * py + (cj + 2) * pz + (ck + 2));
                    Run<T, 4> zr[2][2];
                    Run<T, 2> xr[2][2], yr[2][2];          // [outer side: 0 = -1, 1 = +2][the other in-cell index]
#pragma unroll
                    for (int a = 0; a < 2; ++a) {
#pragma unroll
                        for (int c2 = 0; c2 < 2; ++c2) {
                            zr[a][c2] = *reinterpret_cast<const Run<T, 4>*>(q + a * py + c2 * pz - 1);
                            xr[a][c2] = *reinterpret_cast<const Run<T, 2>*>(q + (a ? 2 : -1) * py + c2 * pz);
                            yr[a][c2] = *reinterpret_cast<const Run<T, 2>*>(q + c2 * py + (a ? 2 : -1) * pz);
                        }
                    }
                    const T gm = gamma[b];
                    const T nig2 = -((gm != (T)0) ? (T)1 / (gm * gm) : (T)0);
                    const T hx = (T)0.5 * ((T)1 / cell[b * 3 + 0]);
                    const T hy = (T)0.5 * ((T)1 / cell[b * 3 + 1]);
                    const T hz = (T)0.5 * ((T)1 / cell[b * 3 + 2]);
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const int a = k >> 2, c2 = (k >> 1) & 1, e = k & 1;    // node (ci + a, cj + c2, ck + e)
                        const int I = ci + a, J = cj + c2, K = ck + e;
                        // neighbours along x: a = 0 -> (ci - 1, ci + 1), a = 1 -> (ci, ci + 2); value at (., cj + c2, ck + e)
                        const T xlo = a ? zr[0][c2].v[1 + e] : xr[0][c2].v[e];
                        const T xhi = a ? xr[1][c2].v[e] : zr[1][c2].v[1 + e];
                        const T ylo = c2 ? zr[a][0].v[1 + e] : yr[0][a].v[e];
                        const T yhi = c2 ? yr[1][a].v[e] : zr[a][1].v[1 + e];
                        const T zlo = zr[a][c2].v[e], zhi = zr[a][c2].v[e + 2];
                        node[k].x = node_force(xlo, xhi, hx, nig2, I > 0 && I < g[0] - 1);
                        node[k].y = node_force(ylo, yhi, hy, nig2, J > 0 && J < g[1] - 1);
                        node[k].z = node_force(zlo, zhi, hz, nig2, K > 0 && K < g[2] - 1);
                    }
                }
                const Node* Fb = reinterpret_cast<const Node*>(F) + b * (int64_t)gx * gy * gz;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int ix = i0[0] + (k >> 2), iy = i0[1] + ((k >> 1) & 1), iz = i0[2] + (k & 1);
                    const bool valid = ix >= 0 && ix < g[0] && iy >= 0 && iy < g[1] && iz >= 0 && iz < g[2];
                    if (!FROM_PHI) {
                        const int cx = min(max(ix, 0), g[0] - 1), cy = min(max(iy, 0), g[1] - 1), cz = min(max(iz, 0), g[2] - 1);
                        node[k] = Fb[((int64_t)cx * g[1] + cy) * g[2] + cz];
                    } else if (!valid) {
                        node[k].x = node[k].y = node[k].z = (T)0;   // built from halo values, which are undefined
                    }
                    w[k] = valid ? (1.0 - fabs(u[0] - ix)) * (1.0 - fabs(u[1] - iy)) * (1.0 - fabs(u[2] - iz)) * kElementaryCharge
                                 : 0.0;
                }
                double fx = 0.0, fy = 0.0, fz = 0.0;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    fx += w[k] * (double)node[k].x;
                    fy += w[k] * (double)node[k].y;
                    fz += w[k] * (double)node[k].z;
                }
                // a grid without extent along an axis (all particles in one plane or point: sigma = 0, cell = 0) gives the
                // reference 0 / 0 or x / 0 cell positions and NaN momenta for every particle; the clamped indices above would
                // hide that as "outside the grid, no kick"
                if (!(isfinite(u[0]) && isfinite(u[1]) && isfinite(u[2]))) fx = fy = fz = __builtin_nan("");
                const double dtb = (double)dt[b];
                s[1] += fx * dtb;
                s[3] += fy * dtb;
                s[5] += fz * dtb;
                from_si(rf, s, v);
                if (post_map) {
                    // the linear run that follows the kick, applied while the particle is still in registers: the same fma
                    // chain as chx_apply_affine7 on the kicked coordinates ROUNDED TO T first, i.e. bit-identical to writing
                    // the kicked beam and tracking it through the run in a second pass
                    const T* __restrict__ R = post_map + ((BR == 1) ? 0 : b) * 49;
                    T xk[7];
#pragma unroll
                    for (int j = 0; j < 7; ++j) xk[j] = (T)v[j];
#pragma unroll
                    for (int i = 0; i < 7; ++i) {
                        T acc = R[i * 7] * xk[0];
#pragma unroll
                        for (int j = 1; j < 7; ++j) acc = fma(R[i * 7 + j], xk[j], acc);
                        lds[p * 7 + i] = acc;
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 7; ++j) lds[p * 7 + j] = (T)v[j];
                }
            } else {
#pragma unroll
                for (int j = 0; j < 7; ++j) lds[p * 7 + j] = (T)s[j];
            }
        }
    }
    chx_wave_sync();
    wave_tile_store<T>(x_out + (b * N + n0 + wrow) * 7, lds + wrow * 7, wvalid * 7, vout, true);
}

inline dim3 cell_grid(int64_t n, int64_t B) {
    int64_t g = (n + CHX_BLOCK - 1) / CHX_BLOCK;
    int64_t cap = 16384 / B;
    if (cap < 1) cap = 1;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return dim3((unsigned)g, (unsigned)B);
}

// Grid geometry of one kick from the beam moments (space_charge_kick.py:531-550, 110-130): replaces ~25 tiny tensor
// ops (each a kernel launch) by one. Every step is rounded in T in the order the reference's tensor expressions
// round (sigma = sqrt(cov) cast to T, half = extent * sigma, cell = 2 half / g, gamma = E / m, beta, dt = L / (c beta)).
template <typename T>
__global__ void sc_geometry_kernel(const double* __restrict__ mom, const T* __restrict__ ext, const T* __restrict__ energy,
                                   const T* __restrict__ length, double mass, double pot_factor, int64_t B, int64_t Bm,
                                   int64_t Bext, int64_t Be, int64_t Bl, int gx, int gy, int gz, T* __restrict__ half,
                                   T* __restrict__ cell, T* __restrict__ gamma_out, T* __restrict__ dt,
                                   T* __restrict__ scale, T* __restrict__ extent, double* __restrict__ pot_scale,
                                   int* __restrict__ tile_hdr, int merge_rows) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    // a later kick of a chain of tile-ordered kicks (chx_sc_tiles.h): the header rolls over here, exactly as in
    // sc_geometry_partials_kernel (chx_moments.hip) — this kernel runs before the deposit / gather kernels of the kick
    if (tile_hdr && b == 0 && tile_hdr[1]) {
        tile_hdr[0] ^= 1;
        tile_hdr[1] = 0;
    }
    double var[3];
    if (merge_rows > 0) {
        // mom[merge_rows][29] are the moments of the SHARDS of one beam (what the ranks all-gathered): the three variances of
        // their union by the arithmetic of merge_moments_kernel (chx_moments.hip), entry by entry — bit-identical to
        // chx_merge_moments followed by this kernel with merge_rows = 0
        const int col[3] = {0, 2, 4}, diag[3] = {0, 11, 18};
        double W = 0.0, W2 = 0.0, mu[3] = {0.0, 0.0, 0.0};
        for (int r = 0; r < merge_rows; ++r) {
            const double* p = mom + (int64_t)r * CHX_MOM_NOUT;
            if (!(p[0] > 0.0)) continue;
            W += p[0];
            W2 += p[1];
            for (int d = 0; d < 3; ++d) mu[d] += p[0] * p[2 + col[d]];
        }
        for (int d = 0; d < 3; ++d) mu[d] /= W;
        double M[3] = {0.0, 0.0, 0.0};
        for (int r = 0; r < merge_rows; ++r) {
            const double* p = mom + (int64_t)r * CHX_MOM_NOUT;
            if (!(p[0] > 0.0)) continue;
            const double cf = p[0] - p[1] / p[0];
            for (int d = 0; d < 3; ++d) {
                const double dd = p[2 + col[d]] - mu[d];
                M[d] += p[8 + diag[d]] * cf + p[0] * dd * dd;
            }
        }
        const double cf = W - W2 / W;
        for (int d = 0; d < 3; ++d) var[d] = M[d] / cf;
    } else {
        const double* m = mom + ((Bm == 1) ? 0 : b) * CHX_MOM_NOUT;
        var[0] = m[8]; var[1] = m[8 + 11]; var[2] = m[8 + 18];  // cov_xx, cov_yy, cov_tautau
    }
    sc_geometry_row<T>(var, ext + ((Bext == 1) ? 0 : b) * 3, energy[(Be == 1) ? 0 : b], length[(Bl == 1) ? 0 : b], mass,
                       pot_factor, gx, gy, gz, half + b * 3, cell + b * 3, gamma_out + b, dt + b, scale + b * 3,
                       extent + b * 6, pot_scale + b);
}

bool bins_ok(const int32_t* bins) {
    return bins && bins[0] >= 2 && bins[1] >= 2 && bins[2] >= 2 && bins[0] <= 1024 && bins[1] <= 1024 &&
           bins[2] <= 1024;
}

}  // namespace

extern "C" int chx_sc_geometry(const double* moments, const void* grid_extent, const void* energy, const void* length,
                               double mass_eV, double pot_factor, int64_t B, int64_t Bm, int64_t Bext, int64_t Be,
                               int64_t Bl, const int32_t* bins, int dtype, void* half, void* cell, void* gamma,
                               void* dt, void* scale, void* extent, double* pot_scale, void* stream) {
    return chx_sc_geometry_tiles(moments, grid_extent, energy, length, mass_eV, pot_factor, B, Bm, Bext, Be, Bl, bins, dtype, half,
                                 cell, gamma, dt, scale, extent, pot_scale, nullptr, 0, stream);
}

extern "C" int chx_sc_geometry_tiles(const double* moments, const void* grid_extent, const void* energy, const void* length,
                                     double mass_eV, double pot_factor, int64_t B, int64_t Bm, int64_t Bext, int64_t Be,
                                     int64_t Bl, const int32_t* bins, int dtype, void* half, void* cell, void* gamma,
                                     void* dt, void* scale, void* extent, double* pot_scale, void* tile_header, int32_t merge_rows,
                                     void* stream) {
    if (merge_rows < 0 || (merge_rows > 0 && B != 1)) return CHX_ERR_INVALID_ARG;
    if (!moments || !grid_extent || !energy || !length || !half || !cell || !gamma || !dt || !scale || !extent ||
        !pot_scale || B < 1 || !bins_ok(bins))
        return CHX_ERR_INVALID_ARG;
    if (!chx_bcast_ok(Bm, B) || !chx_bcast_ok(Bext, B) || !chx_bcast_ok(Be, B) || !chx_bcast_ok(Bl, B))
        return CHX_ERR_INVALID_ARG;
    hipStream_t s = (hipStream_t)stream;
    const unsigned nb = (unsigned)((B + 63) / 64);
    if (dtype == CHX_F32)
        hipLaunchKernelGGL(sc_geometry_kernel<float>, dim3(nb), dim3(64), 0, s, moments, (const float*)grid_extent,
                           (const float*)energy, (const float*)length, mass_eV, pot_factor, B, Bm, Bext, Be, Bl, bins[0],
                           bins[1], bins[2], (float*)half, (float*)cell, (float*)gamma, (float*)dt, (float*)scale,
                           (float*)extent, pot_scale, (int*)tile_header, (int)merge_rows);
    else if (dtype == CHX_F64)
        hipLaunchKernelGGL(sc_geometry_kernel<double>, dim3(nb), dim3(64), 0, s, moments, (const double*)grid_extent,
                           (const double*)energy, (const double*)length, mass_eV, pot_factor, B, Bm, Bext, Be, Bl, bins[0],
                           bins[1], bins[2], (double*)half, (double*)cell, (double*)gamma, (double*)dt, (double*)scale,
                           (double*)extent, pot_scale, (int*)tile_header, (int)merge_rows);
    else
        return CHX_ERR_DTYPE;
    CHX_CHECK_LAUNCH();
    return CHX_OK;
}

extern "C" size_t chx_sc_igf_workspace_bytes(int64_t B, const int32_t* bins) {
    if (B < 1 || !bins_ok(bins)) return 0;
    return (size_t)B * (size_t)(bins[0] + 1) * (bins[1] + 1) * (bins[2] + 1) * sizeof(double);
}

extern "C" int chx_sc_igf(const void* cell, const void* gamma, int64_t B, const int32_t* bins, int dtype,
                          void* G_out, int64_t ldz, void* workspace, size_t workspace_bytes, void* stream) {
    if (!cell || !gamma || !G_out || B < 1 || B > 65535 || !bins_ok(bins)) return CHX_ERR_INVALID_ARG;
    if (dtype != CHX_F32 && dtype != CHX_F64) return CHX_ERR_DTYPE;
    if (!workspace || workspace_bytes < chx_sc_igf_workspace_bytes(B, bins)) return CHX_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const int gx = bins[0], gy = bins[1], gz = bins[2];
    if (ldz == 0) ldz = 2 * gz;
    if (ldz < 2 * gz) return CHX_ERR_INVALID_ARG;
    const size_t esz = dtype == CHX_F32 ? 4 : 8;
    const size_t gbytes = (size_t)B * 4 * (size_t)gx * gy * (size_t)ldz * esz;
    if (hipMemsetAsync(G_out, 0, gbytes, s) != hipSuccess) return CHX_ERR_LAUNCH;  // index-g planes stay 0
    const int64_t npts = (int64_t)(gx + 1) * (gy + 1) * (gz + 1);
    double* table = (double*)workspace;
    if (dtype == CHX_F32) {
        hipLaunchKernelGGL(igf_table_kernel<float>, cell_grid(npts, B), dim3(CHX_BLOCK), 0, s,
                           (const float*)cell, (const float*)gamma, gx, gy, gz, table);
        CHX_CHECK_LAUNCH();
        hipLaunchKernelGGL(igf_fill_kernel<float>, cell_grid((int64_t)gx * gy * gz, B), dim3(CHX_BLOCK), 0, s,
                           table, gx, gy, gz, ldz, (float*)G_out);
    } else {
        hipLaunchKernelGGL(igf_table_kernel<double>, cell_grid(npts, B), dim3(CHX_BLOCK), 0, s,
                           (const double*)cell, (const double*)gamma, gx, gy, gz, table);
        CHX_CHECK_LAUNCH();
        hipLaunchKernelGGL(igf_fill_kernel<double>, cell_grid((int64_t)gx * gy * gz, B), dim3(CHX_BLOCK), 0, s,
                           table, gx, gy, gz, ldz, (double*)G_out);
    }
    CHX_CHECK_LAUNCH();
    return CHX_OK;
}

// The doubled, mirrored Green array of chx_sc_igf from a corner table that is already there (e.g. one of the derivative tables
// of chx_sc_igf_table_grad: the backward pass of the Poisson stage on grids the pruned transforms do not cover).
extern "C" int chx_sc_igf_from_table(const double* table, int64_t B, const int32_t* bins, int dtype, void* G_out, int64_t ldz,
                                     void* stream) {
    if (!table || !G_out || B < 1 || B > 65535 || !bins_ok(bins)) return CHX_ERR_INVALID_ARG;
    if (dtype != CHX_F32 && dtype != CHX_F64) return CHX_ERR_DTYPE;
    hipStream_t s = (hipStream_t)stream;
    const int gx = bins[0], gy = bins[1], gz = bins[2];
    if (ldz == 0) ldz = 2 * gz;
    if (ldz < 2 * gz) return CHX_ERR_INVALID_ARG;
    const size_t esz = dtype == CHX_F32 ? 4 : 8;
    if (hipMemsetAsync(G_out, 0, (size_t)B * 4 * (size_t)gx * gy * (size_t)ldz * esz, s) != hipSuccess) return CHX_ERR_LAUNCH;
    if (dtype == CHX_F32)
        hipLaunchKernelGGL(igf_fill_kernel<float>, cell_grid((int64_t)gx * gy * gz, B), dim3(CHX_BLOCK), 0, s, table, gx, gy, gz,
                           ldz, (float*)G_out);
    else
        hipLaunchKernelGGL(igf_fill_kernel<double>, cell_grid((int64_t)gx * gy * gz, B), dim3(CHX_BLOCK), 0, s, table, gx, gy, gz,
                           ldz, (double*)G_out);
    CHX_CHECK_LAUNCH();
    return CHX_OK;
}

// Only the corner table of the primitive (the input of chx_sc_green_spectrum, chx_fft.hip).
extern "C" int chx_sc_igf_table(const void* cell, const void* gamma, int64_t B, const int32_t* bins, int dtype,
                                double* table, void* stream) {
    if (!cell || !gamma || !table || B < 1 || B > 65535 || !bins_ok(bins)) return CHX_ERR_INVALID_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int gx = bins[0], gy = bins[1], gz = bins[2];
    const int64_t npts = (int64_t)(gx + 1) * (gy + 1) * (gz + 1);
    if (dtype == CHX_F32)
        hipLaunchKernelGGL(igf_table_kernel<float>, cell_grid(npts, B), dim3(CHX_BLOCK), 0, s, (const float*)cell,
                           (const float*)gamma, gx, gy, gz, table);
    else if (dtype == CHX_F64)
        hipLaunchKernelGGL(igf_table_kernel<double>, cell_grid(npts, B), dim3(CHX_BLOCK), 0, s, (const double*)cell,
                           (const double*)gamma, gx, gy, gz, table);
    else
        return CHX_ERR_DTYPE;
    CHX_CHECK_LAUNCH();
    return CHX_OK;
}

extern "C" int chx_sc_spectral_mul(void* rho_hat, const void* G_hat, const double* scale, int64_t B,
                                   int64_t n_complex, int dtype, void* stream) {
    if (!rho_hat || !G_hat || !scale || B < 1 || B > 65535 || n_complex < 1) return CHX_ERR_INVALID_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == CHX_F32)
        hipLaunchKernelGGL(spectral_mul_kernel<float>, cell_grid(n_complex, B), dim3(CHX_BLOCK), 0, s,
                           (float*)rho_hat, (const float*)G_hat, scale, n_complex);
    else if (dtype == CHX_F64)
        hipLaunchKernelGGL(spectral_mul_kernel<double>, cell_grid(n_complex, B), dim3(CHX_BLOCK), 0, s,
                           (double*)rho_hat, (const double*)G_hat, scale, n_complex);
    else
        return CHX_ERR_DTYPE;
    CHX_CHECK_LAUNCH();
    return CHX_OK;
}

extern "C" int chx_sc_gradient(const void* phi, const void* cell, const void* gamma, int64_t B,
                               const int32_t* bins, int phi_doubled, int64_t ldz, int dtype, void* F_out,
                               void* stream) {
    if (!phi || !cell || !gamma || !F_out || B < 1 || B > 65535 || !bins_ok(bins)) return CHX_ERR_INVALID_ARG;
    if (ldz == 0) ldz = 2 * bins[2];
    if (phi_doubled && ldz < 2 * bins[2]) return CHX_ERR_INVALID_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int64_t ncell = (int64_t)bins[0] * bins[1] * bins[2];
    if (dtype == CHX_F32)
        hipLaunchKernelGGL(gradient_kernel<float>, cell_grid(ncell, B), dim3(CHX_BLOCK), 0, s,
                           (const float*)phi, (const float*)cell, (const float*)gamma, bins[0], bins[1],
                           bins[2], phi_doubled, ldz, (float*)F_out);
    else if (dtype == CHX_F64)
        hipLaunchKernelGGL(gradient_kernel<double>, cell_grid(ncell, B), dim3(CHX_BLOCK), 0, s,
                           (const double*)phi, (const double*)cell, (const double*)gamma, bins[0], bins[1],
                           bins[2], phi_doubled, ldz, (double*)F_out);
    else
        return CHX_ERR_DTYPE;
    CHX_CHECK_LAUNCH();
    return CHX_OK;
}

// ---- hipFFT plans for the Hockney convolution (space_charge_kick.py:306-314) ------------------------------------
// Real-to-complex / complex-to-real 3-D transforms of the (2gx, 2gy, 2gz) arrays, IN PLACE in the padded layout
// [B][2gx][2gy][2gz + 2] (complex view [B][2gx][2gy][gz + 1]), unnormalised. Going through hipFFT directly instead of
// torch.fft removes the defensive input copies around every transform (~20 us each at 256^3), the separate complex
// output arrays and the 1/N scaling pass. Two forward plans so that rho and the Green function can be transformed on
// different streams at the same time.
struct chx_sc_fft_plan {
    hipfftHandle fwd[2];
    hipfftHandle inv;
    int dtype;
    int64_t B;
    int32_t bins[3];
};

extern "C" int chx_sc_fft_plan_create(int64_t B, const int32_t* bins, int dtype, void** plan_out) {
    if (!plan_out || B < 1 || B > 65535 || !bins_ok(bins)) return CHX_ERR_INVALID_ARG;
    if (dtype != CHX_F32 && dtype != CHX_F64) return CHX_ERR_DTYPE;
    if (8.0 * bins[0] * bins[1] * (bins[2] + 1.0) > 2147483647.0) return CHX_ERR_INVALID_ARG;  // hipFFT int distances
    chx_sc_fft_plan* p = new (std::nothrow) chx_sc_fft_plan();
    if (!p) return CHX_ERR_WORKSPACE;
    p->dtype = dtype;
    p->B = B;
    for (int d = 0; d < 3; ++d) p->bins[d] = bins[d];
    int n[3] = {2 * bins[0], 2 * bins[1], 2 * bins[2]};
    int rembed[3] = {n[0], n[1], n[2] + 2};        // padded real layout
    int cembed[3] = {n[0], n[1], n[2] / 2 + 1};    // complex layout in the same bytes
    const int rdist = n[0] * n[1] * (n[2] + 2), cdist = n[0] * n[1] * (n[2] / 2 + 1);
    const hipfftType tf = dtype == CHX_F32 ? HIPFFT_R2C : HIPFFT_D2Z, ti = dtype == CHX_F32 ? HIPFFT_C2R : HIPFFT_Z2D;
    bool ok = true;
    int made = 0;
    for (int k = 0; k < 2 && ok; ++k, ++made)
        ok = hipfftPlanMany(&p->fwd[k], 3, n, rembed, 1, rdist, cembed, 1, cdist, tf, (int)B) == HIPFFT_SUCCESS;
    if (ok) {
        ok = hipfftPlanMany(&p->inv, 3, n, cembed, 1, cdist, rembed, 1, rdist, ti, (int)B) == HIPFFT_SUCCESS;
        if (ok) ++made;
    }
    if (!ok) {
        for (int k = 0; k < made && k < 2; ++k) hipfftDestroy(p->fwd[k]);
        delete p;
        return CHX_ERR_LAUNCH;
    }
    *plan_out = p;
    return CHX_OK;
}

extern "C" int chx_sc_fft_plan_destroy(void* plan) {
    if (!plan) return CHX_ERR_INVALID_ARG;
    chx_sc_fft_plan* p = (chx_sc_fft_plan*)plan;
    hipfftDestroy(p->fwd[0]);
    hipfftDestroy(p->fwd[1]);
    hipfftDestroy(p->inv);
    delete p;
    return CHX_OK;
}

// direction: 0 / 1 = forward with plan 0 / 1 (real padded -> complex), 2 = inverse (complex -> real padded)
extern "C" int chx_sc_fft_exec(void* plan, int direction, void* data, void* stream) {
    if (!plan || !data || direction < 0 || direction > 2) return CHX_ERR_INVALID_ARG;
    chx_sc_fft_plan* p = (chx_sc_fft_plan*)plan;
    hipfftHandle h = direction == 2 ? p->inv : p->fwd[direction];
    if (hipfftSetStream(h, (hipStream_t)stream) != HIPFFT_SUCCESS) return CHX_ERR_LAUNCH;
    hipfftResult r;
    if (p->dtype == CHX_F32)
        r = direction == 2 ? hipfftExecC2R(h, (hipfftComplex*)data, (hipfftReal*)data)
                           : hipfftExecR2C(h, (hipfftReal*)data, (hipfftComplex*)data);
    else
        r = direction == 2 ? hipfftExecZ2D(h, (hipfftDoubleComplex*)data, (hipfftDoubleReal*)data)
                           : hipfftExecD2Z(h, (hipfftDoubleReal*)data, (hipfftDoubleComplex*)data);
    return r == HIPFFT_SUCCESS ? CHX_OK : CHX_ERR_LAUNCH;
}

template <int MODE, bool FROM_PHI = false>
static int launch_particle(const void* x_in, const void* F, const void* half, const void* cell,
                           const void* energy, const void* dt, double mass_eV, int64_t B, int64_t Bx,
                           int64_t Be, int64_t N, const int32_t* bins, int dtype, void* x_out,
                           void* stream, const void* post_map = nullptr, int64_t BR = 1, const void* gamma = nullptr) {
    if (!x_in || !x_out || !energy || B < 1 || N < 1 || B > 65535) return CHX_ERR_INVALID_ARG;
    if (!chx_bcast_ok(Bx, B) || !chx_bcast_ok(Be, B)) return CHX_ERR_INVALID_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int64_t tiles = (N + CHX_BLOCK - 1) / CHX_BLOCK;
    if (tiles > 0x7fffffffLL) return CHX_ERR_INVALID_ARG;
    dim3 grid((unsigned)tiles, (unsigned)B);
    const int gx = bins ? bins[0] : 0, gy = bins ? bins[1] : 0, gz = bins ? bins[2] : 0;
    if (dtype == CHX_F32 && MODE == 0 && FROM_PHI && !sc_gather_fp64() && sc_phi_offsets_fit32(bins))
        hipLaunchKernelGGL(sc_particle32_kernel, grid, dim3(CHX_BLOCK), 0, s, (const float*)x_in, (const float*)F, (const float*)half,
                           (const float*)cell, (const float*)energy, (const float*)dt, (float)(1.0 / mass_eV), (float)(kC / mass_eV), Bx, Be, N,
                           gx, gy, gz, (float*)x_out, (const float*)post_map, BR, (const float*)gamma);
    else if (dtype == CHX_F32)
        hipLaunchKernelGGL((sc_particle_kernel<float, MODE, FROM_PHI>), grid, dim3(CHX_BLOCK), 0, s, (const float*)x_in,
                           (const float*)F, (const float*)half, (const float*)cell, (const float*)energy,
                           (const float*)dt, mass_eV, Bx, Be, N, gx, gy, gz, (float*)x_out, (const float*)post_map, BR,
                           (const float*)gamma);
    else if (dtype == CHX_F64)
        hipLaunchKernelGGL((sc_particle_kernel<double, MODE, FROM_PHI>), grid, dim3(CHX_BLOCK), 0, s, (const double*)x_in,
                           (const double*)F, (const double*)half, (const double*)cell, (const double*)energy,
                           (const double*)dt, mass_eV, Bx, Be, N, gx, gy, gz, (double*)x_out, (const double*)post_map, BR,
                           (const double*)gamma);
    else
        return CHX_ERR_DTYPE;
    CHX_CHECK_LAUNCH();
    return CHX_OK;
}

extern "C" int chx_sc_gather_kick(const void* x_in, const void* F, const void* half, const void* cell,
                                  const void* energy, const void* dt, double mass_eV, int64_t B,
                                  int64_t Bx, int64_t Be, int64_t N, const int32_t* bins, int dtype,
                                  void* x_out, void* stream) {
    if (!F || !half || !cell || !dt || !bins_ok(bins)) return CHX_ERR_INVALID_ARG;
    return launch_particle<0>(x_in, F, half, cell, energy, dt, mass_eV, B, Bx, Be, N, bins, dtype, x_out, stream);
}

extern "C" int chx_sc_gather_kick_mapped(const void* x_in, const void* F, const void* half, const void* cell,
                                         const void* energy, const void* dt, double mass_eV, int64_t B, int64_t Bx,
                                         int64_t Be, int64_t N, const int32_t* bins, int dtype, const void* post_map,
                                         int64_t BR, void* x_out, void* stream) {
    if (!F || !half || !cell || !dt || !bins_ok(bins) || !post_map || !chx_bcast_ok(BR, B)) return CHX_ERR_INVALID_ARG;
    return launch_particle<0>(x_in, F, half, cell, energy, dt, mass_eV, B, Bx, Be, N, bins, dtype, x_out, stream, post_map, BR);
}

// Gather + kick straight from the potential with a halo (chx_sc_convolve_halo): the central differences of chx_sc_gradient
// are taken per particle, bit-identical to chx_sc_gradient + chx_sc_gather_kick(_mapped); post_map may be null.
extern "C" int chx_sc_gather_kick_phi(const void* x_in, const void* phi_halo, const void* half, const void* cell,
                                      const void* gamma, const void* energy, const void* dt, double mass_eV, int64_t B,
                                      int64_t Bx, int64_t Be, int64_t N, const int32_t* bins, int dtype, const void* post_map,
                                      int64_t BR, void* x_out, void* stream) {
    if (!phi_halo || !half || !cell || !gamma || !dt || !bins_ok(bins)) return CHX_ERR_INVALID_ARG;
    if (post_map && !chx_bcast_ok(BR, B)) return CHX_ERR_INVALID_ARG;
    return launch_particle<0, true>(x_in, phi_halo, half, cell, energy, dt, mass_eV, B, Bx, Be, N, bins, dtype, x_out, stream,
                                    post_map, post_map ? BR : 1, gamma);
}

// ---- gather + kick on the tile-ordered beam of a chain of kicks (chx_sc_tiles.h) ----------------------------------------
// The arithmetic per particle IS sc_particle_kernel's (MODE 0, FROM_PHI): same values from the same nodes, bit-identical
// result. What the tile order buys is locality: the lanes of a wave sit in one 8^3 tile, so the 12 line requests per particle for
// the 32 potential values around its cell hit the CU's vector cache instead of the L2 — measured at 1e6 particles on 128^3
// (benchmarks/tile_gather_bench.py): 63.1 us on rows in the caller's order, 33.1 us on the same rows in tile order. (Staging
// each tile's 12^3 potential block in LDS explicitly was slower in every variant tried: 134 us with one block per workgroup
// and a barrier per tile segment, 179-195 us with one block per wave — the binary search for the tile, the staging and the
// spills of the fp64 particle step under a 128-VGPR cap cost more than the cache already gives for free.)
// Destination of a row: its own slot; the caller's particle index (perm) for the last kick of a chain; or, when the deposit of
// this kick found the order stale (header.scatter_now), a slot of the particle's CURRENT home tile taken from the cursors the
// merge pass prepared — the rows, weights, charges and the permutation then move to the other copy of the state arrays.
namespace {

template <typename T>
struct alignas(4 * sizeof(T)) PhiNode { T x, y, z, pad; };

// force components at the eight nodes (ci + a, cj + c2, ck + e) of a cell from the potential around it; q points at node
// (ci, cj, ck), py / pz are the x / y strides of the array q lives in (global halo array or LDS block)
template <typename T>
__device__ __forceinline__ void phi_cell_forces(const T* q, int py, int pz, int ci, int cj, int ck, const int (&g)[3], T hx, T hy,
                                                T hz, T nig2, PhiNode<T> (&node)[8]) {
    Run<T, 4> zr[2][2];
    Run<T, 2> xr[2][2], yr[2][2];          // [outer side: 0 = -1, 1 = +2][the other in-cell index]
#pragma unroll
    for (int a = 0; a < 2; ++a) {
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) {
            zr[a][c2] = *reinterpret_cast<const Run<T, 4>*>(q + a * py + c2 * pz - 1);
            xr[a][c2] = *reinterpret_cast<const Run<T, 2>*>(q + (a ? 2 : -1) * py + c2 * pz);
            yr[a][c2] = *reinterpret_cast<const Run<T, 2>*>(q + c2 * py + (a ? 2 : -1) * pz);
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int a = k >> 2, c2 = (k >> 1) & 1, e = k & 1;    // node (ci + a, cj + c2, ck + e)
        const int I = ci + a, J = cj + c2, K = ck + e;
        const T xlo = a ? zr[0][c2].v[1 + e] : xr[0][c2].v[e];
        const T xhi = a ? xr[1][c2].v[e] : zr[1][c2].v[1 + e];
        const T ylo = c2 ? zr[a][0].v[1 + e] : yr[0][a].v[e];
        const T yhi = c2 ? yr[1][a].v[e] : zr[a][1].v[1 + e];
        const T zlo = zr[a][c2].v[e], zhi = zr[a][c2].v[e + 2];
        node[k].x = node_force(xlo, xhi, hx, nig2, I > 0 && I < g[0] - 1);
        node[k].y = node_force(ylo, yhi, hy, nig2, J > 0 && J < g[1] - 1);
        node[k].z = node_force(zlo, zhi, hz, nig2, K > 0 && K < g[2] - 1);
    }
}

template <typename T>
struct ScKickCtx {
    RefFrame<double> rf;
    T nig2, hx, hy, hz;
    double halfd[3], celld[3], dtb;
    int g[3];
    const T* post_map;
};

template <typename T>
__device__ __forceinline__ ScKickCtx<T> sc_kick_ctx(const T* half, const T* cell, const T* energy, const T* dt, const T* gamma,
                                                    double mass_eV, int gx, int gy, int gz, const T* post_map) {
    ScKickCtx<T> c;
    c.rf = ref_frame<double>((double)energy[0], mass_eV);
    const T gm = gamma[0];
    c.nig2 = -((gm != (T)0) ? (T)1 / (gm * gm) : (T)0);
    c.hx = (T)0.5 * ((T)1 / cell[0]);
    c.hy = (T)0.5 * ((T)1 / cell[1]);
    c.hz = (T)0.5 * ((T)1 / cell[2]);
    for (int d = 0; d < 3; ++d) { c.halfd[d] = (double)half[d]; c.celld[d] = (double)cell[d]; }
    c.dtb = (double)dt[0];
    c.g[0] = gx; c.g[1] = gy; c.g[2] = gz;
    c.post_map = post_map;
    return c;
}

// SI coordinates, cell position and the clamped base node of a row
template <typename T>
__device__ __forceinline__ void sc_kick_locate(const ScKickCtx<T>& c, const double (&v)[7], double (&s)[7], double (&u)[3],
                                               int (&i0)[3], int (&cn)[3]) {
    to_si(c.rf, v, s);
    const double posd[3] = {s[0], s[2], s[4]};
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        u[d] = (posd[d] + c.halfd[d]) / c.celld[d];
        double fl = floor(u[d]);
        fl = fl > 2.0e9 ? 2.0e9 : (fl < -2.0e9 ? -2.0e9 : fl);
        i0[d] = (int)fl;
        cn[d] = min(max(i0[d], -1), c.g[d] - 1);
    }
}

// interpolation, kick, back to Cheetah coordinates, optional linear map; out in T
template <typename T>
__device__ __forceinline__ void sc_kick_finish(const ScKickCtx<T>& c, double (&s)[7], const double (&u)[3], const int (&i0)[3],
                                               PhiNode<T> (&node)[8], T (&out)[7]) {
    double w[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int ix = i0[0] + (k >> 2), iy = i0[1] + ((k >> 1) & 1), iz = i0[2] + (k & 1);
        const bool valid = ix >= 0 && ix < c.g[0] && iy >= 0 && iy < c.g[1] && iz >= 0 && iz < c.g[2];
        if (!valid) node[k].x = node[k].y = node[k].z = (T)0;   // built from halo values, which are undefined
        w[k] = valid ? (1.0 - fabs(u[0] - ix)) * (1.0 - fabs(u[1] - iy)) * (1.0 - fabs(u[2] - iz)) * kElementaryCharge : 0.0;
    }
    double fx = 0.0, fy = 0.0, fz = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        fx += w[k] * (double)node[k].x;
        fy += w[k] * (double)node[k].y;
        fz += w[k] * (double)node[k].z;
    }
    if (!(isfinite(u[0]) && isfinite(u[1]) && isfinite(u[2]))) fx = fy = fz = __builtin_nan("");
    s[1] += fx * c.dtb;
    s[3] += fy * c.dtb;
    s[5] += fz * c.dtb;
    double v[7];
    from_si(c.rf, s, v);
    if (c.post_map) {
        T xk[7];
#pragma unroll
        for (int j = 0; j < 7; ++j) xk[j] = (T)v[j];
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            T acc = c.post_map[i * 7] * xk[0];
#pragma unroll
            for (int j = 1; j < 7; ++j) acc = fma(c.post_map[i * 7 + j], xk[j], acc);
            out[i] = acc;
        }
    } else {
#pragma unroll
        for (int j = 0; j < 7; ++j) out[j] = (T)v[j];
    }
}

// where a finished row goes (see the header of this section); `active` lanes have a row. Wave-uniform control flow.
template <typename T>
__device__ __forceinline__ int64_t sc_row_dest(bool active, int64_t r, int mode /*0 slot, 1 perm, 2 scatter*/, int par, int64_t N,
                                               const int* perm2, const uint16_t* __restrict__ home, int* __restrict__ cursor,
                                               T* ws2, T* cs2, int* perm2w) {
    if (mode == 0) return r;
    if (mode == 1) return active ? (int64_t)perm2[(int64_t)par * N + r] : 0;
    // scatter: lanes of the wave that go to the same tile share one atomic on its cursor
    const int lane = threadIdx.x & 63;
    const int h = active ? (int)home[r] : -1;
    int64_t dst = 0;
    unsigned long long todo = __ballot(active);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int h0 = __shfl(h, leader, 64);
        const unsigned long long same = __ballot(active && h == h0) & todo;
        int base = 0;
        if (lane == leader) base = atomicAdd(&cursor[h0], __popcll(same));
        base = __shfl(base, leader, 64);
        if ((same >> lane) & 1ull) dst = base + __popcll(same & ((1ull << lane) - 1ull));
        todo &= ~same;
    }
    if (active) {
        const int64_t from = (int64_t)par * N + r, to = (int64_t)(par ^ 1) * N + dst;
        perm2w[to] = perm2[from];
        ws2[to] = ws2[from];
        cs2[to] = cs2[from];
    }
    return dst;
}

// R segments of 64 rows per wave, one after the other. R = 1 is what runs. Measured with R = 4 and R = 2 (round 5; the idea: what a
// lane does once per launch — the kick's constants, ~90 instructions; the eight float64 block sums of the beam sizes, 144 DPP
// instructions and two barriers = 3.6 us of the 38 us pass, benchmarks/_gather_sigma_cost.sh — is shared by R rows): the constants
// and the running sums stay live across the loop, the scalar registers overflow into vector registers, 114 -> 178 (R = 2) / 200
// (R = 4) VGPRs, two waves per SIMD instead of four, 38 -> 48 us. Capped at 128 VGPRs the kernel spills 76 registers.
template <typename T, int R>
__global__ __launch_bounds__(CHX_BLOCK) void sc_tile_particle_kernel(
    const T* __restrict__ src, const ScTileHeader* __restrict__ hdr, int* perm2, T* ws2, T* cs2, const uint16_t* __restrict__ home,
    int* __restrict__ cursor, const T* __restrict__ phi, const T* __restrict__ half, const T* __restrict__ cell,
    const T* __restrict__ energy, const T* __restrict__ dt, const T* __restrict__ gamma, double mass_eV, int64_t N, int gx, int gy,
    int gz, T* __restrict__ x_out, const T* __restrict__ post_map, int unpermute, double* __restrict__ sigma_partials,
    int* __restrict__ newcount, int nt, int* __restrict__ mis, double* __restrict__ sums_add /*[8][256] or null*/,
    double* __restrict__ sums_clear) {
    __shared__ __attribute__((aligned(16))) T lds[CHX_BLOCK * 7];
    __shared__ double red[4 * 8];
    const int par = hdr->parity;
    const int mode = hdr->scatter_now ? 2 : (unpermute ? 1 : 0);
    // the deposit's counters (read by the schedule kernel in front of this pass) go back to zero for the next kick's deposit
    for (int k = (int)blockIdx.x * CHX_BLOCK + threadIdx.x; k < nt; k += (int)gridDim.x * CHX_BLOCK) newcount[k] = 0;
    if (blockIdx.x == 0 && threadIdx.x < kScMisSlots) mis[threadIdx.x] = 0;
    // ... and the set of beam-size sums the NEXT gather pass adds into (chx_sc_geom_dev.h; its last readers ran in front of this pass)
    if (sums_clear) {
        for (int k = (int)blockIdx.x * CHX_BLOCK + threadIdx.x; k < 8 * 256; k += (int)gridDim.x * CHX_BLOCK) sums_clear[k] = 0.0;
    }
    const bool vin = chx_aligned16(src), vout = chx_aligned16(x_out);
    const ScKickCtx<T> ctx = sc_kick_ctx<T>(half, cell, energy, dt, gamma, mass_eV, gx, gy, gz, post_map);
    const int pz = gz + 4, py = (gy + 4) * pz;            // halo of 2: every index below is in bounds
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    T* wlds = lds + wave * 64 * 7;                        // staged per wave like sc_particle_kernel: no workgroup barrier
    double a[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = 0.0;
#pragma unroll 1
    for (int r = 0; r < R; ++r) {
        const ScKickCtx<T>& c = ctx;
        const int64_t w0 = (((int64_t)blockIdx.x * 4 + wave) * R + r) * 64;      // first row of this wave's segment
        const int wvalid = (N - w0 <= 0) ? 0 : (int)((N - w0 < 64) ? (N - w0) : 64);
        if (wvalid == 0) break;                           // (wave-uniform)
        wave_tile_load<T>(src + w0 * 7, wlds, wvalid * 7, vin, true);
        chx_wave_sync();
        const bool active = lane < wvalid;
        T out[7];
        if (active) {
            double v[7], s[7], u[3];
            int i0[3], cn[3];
#pragma unroll
            for (int j = 0; j < 7; ++j) v[j] = (double)wlds[lane * 7 + j];
            sc_kick_locate<T>(c, v, s, u, i0, cn);
            PhiNode<T> node[8];
            phi_cell_forces<T>(phi + ((int64_t)(cn[0] + 2) * py + (cn[1] + 2) * pz + (cn[2] + 2)), py, pz, cn[0], cn[1], cn[2], c.g,
                               c.hx, c.hy, c.hz, c.nig2, node);
            sc_kick_finish<T>(c, s, u, i0, node, out);
        }
        if (sigma_partials && active) {
            // the rows written here are the beam the NEXT kick of the chain sees: its three variances (space_charge_kick.py:531-538)
            // are accumulated on the way — the sums of sc_sigma_kernel about the origin instead of about the first particle
            const double w = (double)ws2[(int64_t)par * N + w0 + lane];
            const double d0 = (double)out[0], d1 = (double)out[2], d2 = (double)out[4];
            a[0] += w;
            a[1] += w * w;
            const double w0d = w * d0, w1d = w * d1, w2d = w * d2;
            a[2] += w0d; a[3] += w1d; a[4] += w2d;
            a[5] += w0d * d0; a[6] += w1d * d1; a[7] += w2d * d2;
        }
        if (mode == 0) {
            if (active) {
#pragma unroll
                for (int j = 0; j < 7; ++j) wlds[lane * 7 + j] = out[j];
            }
            chx_wave_sync();
            wave_tile_store<T>(x_out + w0 * 7, wlds, wvalid * 7, vout, true);
            chx_wave_sync();
        } else {
            const int64_t dst = sc_row_dest<T>(active, w0 + lane, mode, par, N, perm2, home, cursor, ws2, cs2, perm2);
            if (active) {
#pragma unroll
                for (int j = 0; j < 7; ++j) x_out[dst * 7 + j] = out[j];
            }
        }
    }
    if (sigma_partials) {
        chx_block_sum<8>(a, red);
        if (threadIdx.x == 0) {
#pragma unroll
            for (int k = 0; k < 8; ++k) sigma_partials[(int64_t)k * gridDim.x + blockIdx.x] = a[k];
            // the same sums folded into 256 rows with fp64 atomics: few enough for every workgroup of the next kick's first
            // kernels to reduce them itself (chx_sc_geom_dev.h) — no launch between this pass and those kernels
            if (sums_add) {
#pragma unroll
                for (int k = 0; k < 8; ++k) unsafeAtomicAdd(&sums_add[k * 256 + (blockIdx.x & 255)], a[k]);
            }
        }
    }
}

// The float32 beam's pass: the float32 particle step (sc_kick_row32), everything else as above.
__global__ __launch_bounds__(CHX_BLOCK) void sc_tile_particle32_kernel(
    const float* __restrict__ src, const ScTileHeader* __restrict__ hdr, int* perm2, float* ws2, float* cs2,
    const uint16_t* __restrict__ home, int* __restrict__ cursor, const float* __restrict__ phi, const float* __restrict__ half,
    const float* __restrict__ cell, const float* __restrict__ energy, const float* __restrict__ dt, const float* __restrict__ gamma,
    float inv_mass, float c_over_m, int64_t N, int gx, int gy, int gz, float* __restrict__ x_out, const float* __restrict__ post_map,
    int unpermute, double* __restrict__ sigma_partials, int* __restrict__ newcount, int nt, int* __restrict__ mis,
    double* __restrict__ sums_add /*[8][256] or null*/, double* __restrict__ sums_clear, int diag) {
    __shared__ __attribute__((aligned(16))) float lds[CHX_BLOCK * 7];
    __shared__ double red[16 * 8];
    if (diag & 2) sigma_partials = nullptr;
    const int par = hdr->parity;
    const int mode = hdr->scatter_now ? 2 : (unpermute ? 1 : 0);
    for (int k = (int)blockIdx.x * CHX_BLOCK + threadIdx.x; k < nt; k += (int)gridDim.x * CHX_BLOCK) newcount[k] = 0;
    if (blockIdx.x == 0 && threadIdx.x < kScMisSlots) mis[threadIdx.x] = 0;
    if (sums_clear) {
        for (int k = (int)blockIdx.x * CHX_BLOCK + threadIdx.x; k < 8 * 256; k += (int)gridDim.x * CHX_BLOCK) sums_clear[k] = 0.0;
    }
    const bool vin = chx_aligned16(src), vout = chx_aligned16(x_out);
    const int pz = gz + 4, py = (gy + 4) * pz;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float* wlds = lds + wave * 64 * 7;
    const int64_t w0 = ((int64_t)blockIdx.x * 4 + wave) * 64;             // first row of this wave's segment
    const int wvalid = (N - w0 <= 0) ? 0 : (int)((N - w0 < 64) ? (N - w0) : 64);
    double a[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = 0.0;
    if (wvalid > 0) {                                                     // (wave-uniform)
        wave_tile_load<float>(src + w0 * 7, wlds, wvalid * 7, vin, true);
        const float wgt = (sigma_partials && lane < wvalid) ? ws2[(int64_t)par * N + w0 + lane] : 0.0f;
        const ScKickCtx32 c = sc_kick_ctx32(half, cell, energy, dt, gamma, inv_mass, c_over_m, gx, gy, gz);
        chx_wave_sync();
        const bool active = lane < wvalid;
        float out[7];
        if (active) {
            float v[7];
#pragma unroll
            for (int j = 0; j < 7; ++j) v[j] = wlds[lane * 7 + j];
            sc_kick_row32(c, v, phi, py, pz, out, diag);
            if (post_map) sc_post_map32(post_map, out);
        }
        if (sigma_partials && active) {
            // the rows written here are the beam the NEXT kick of the chain sees (space_charge_kick.py:531-538): its sums about the origin
            const double w = (double)wgt;
            const double d0 = (double)out[0], d1 = (double)out[2], d2 = (double)out[4];
            const double w0d = w * d0, w1d = w * d1, w2d = w * d2;
            a[0] = w; a[1] = w * w;
            a[2] = w0d; a[3] = w1d; a[4] = w2d;
            a[5] = w0d * d0; a[6] = w1d * d1; a[7] = w2d * d2;
        }
        if (mode == 0) {
            if (active) {
#pragma unroll
                for (int j = 0; j < 7; ++j) wlds[lane * 7 + j] = out[j];
            }
            chx_wave_sync();
            wave_tile_store<float>(x_out + w0 * 7, wlds, wvalid * 7, vout, true);
        } else {
            const int64_t dst = sc_row_dest<float>(active, w0 + lane, mode, par, N, perm2, home, cursor, ws2, cs2, perm2);
            if (active) {
#pragma unroll
                for (int j = 0; j < 7; ++j) x_out[dst * 7 + j] = out[j];
            }
        }
    }
    if (sigma_partials) {
        chx_block_sum8_folded(a, red);
        if (threadIdx.x < 8) {                            // thread k holds the workgroup's total of sum k
            sigma_partials[(int64_t)threadIdx.x * gridDim.x + blockIdx.x] = a[0];
            if (sums_add) unsafeAtomicAdd(&sums_add[threadIdx.x * 256 + (blockIdx.x & 255)], a[0]);
        }
    }
}

// (Round 6, measured and dropped — profiles/r06_c4_gather.md: one workgroup per (tile, share) with the central differences of the
// potential at the tile's 10^3 nodes staged once in LDS, the rows of the share read from it with eight 16-byte reads each, bit-identical
// results: 74.7 us for 1e6 rows on 128^3 against 27.4 for the pass above. A share is 3-4 dependent load -> table -> store rounds per
// wave at 127 VGPRs; one wave per 64 rows at 70 VGPRs keeps seven such chains in flight per SIMD and lets the vector cache do the rest.)
}  // namespace

extern "C" int chx_sc_tile_gather_kick(const void* rows, const void* phi_halo, const void* half, const void* cell, const void* gamma,
                                       const void* energy, const void* dt, double mass_eV, int64_t N, const int32_t* bins, int dtype,
                                       const void* post_map, void* state, size_t state_bytes, int unpermute, void* x_out,
                                       void* stream) {
    return chx_sc_tile_gather_kick_chain(rows, phi_halo, half, cell, gamma, energy, dt, mass_eV, N, bins, dtype, post_map, state,
                                         state_bytes, unpermute, x_out, -1, stream);
}

int chx_sc_tile_gather_kick_chain(const void* rows, const void* phi_halo, const void* half, const void* cell, const void* gamma,
                                  const void* energy, const void* dt, double mass_eV, int64_t N, const int32_t* bins, int dtype,
                                  const void* post_map, void* state, size_t state_bytes, int unpermute, void* x_out, int sums_set,
                                  void* stream) {
    if (!phi_halo || !half || !cell || !gamma || !energy || !dt || !state || !x_out || !bins_ok(bins) || N < 1)
        return CHX_ERR_INVALID_ARG;
    if (dtype != CHX_F32 && dtype != CHX_F64) return CHX_ERR_DTYPE;
    const ScTileGeom tg = sc_tile_geom(bins);
    for (int d = 0; d < 3; ++d)
        if (tg.tdim[d] != kScTdim || (bins[d] & (tg.tdim[d] - 1))) return CHX_ERR_INVALID_ARG;
    const ScTileLayout L = sc_tile_layout(N, bins, dtype);
    if (state_bytes < L.total) return CHX_ERR_WORKSPACE;
    char* st = (char*)state;
    if (!rows) rows = st + L.rows_tmp;                    // the rows the sort of the first kick wrote
    const unsigned nwg = (unsigned)L.sigma_blocks;        // kScGatherRows rows per workgroup
    hipStream_t s = (hipStream_t)stream;
    const ScTileHeader* hdr = (const ScTileHeader*)(st + L.hdr);
    const bool with_sums = sums_set >= 0 && !unpermute;
    double* sums_add = with_sums ? (double*)(st + L.sums[sums_set & 1]) : nullptr;
    double* sums_clear = with_sums ? (double*)(st + L.sums[(sums_set & 1) ^ 1]) : nullptr;
    constexpr int R = kScGatherRows / CHX_BLOCK;
    static_assert(R == 1, "sc_tile_particle32_kernel walks one segment of 64 rows per wave");
    if (dtype == CHX_F32 && !sc_gather_fp64() && sc_phi_offsets_fit32(bins))
        hipLaunchKernelGGL(sc_tile_particle32_kernel, dim3(nwg), dim3(CHX_BLOCK), 0, s, (const float*)rows, hdr, (int*)(st + L.perm[0]),
                           (float*)(st + L.ws[0]), (float*)(st + L.cs[0]), (const uint16_t*)(st + L.home), (int*)(st + L.cursor),
                           (const float*)phi_halo, (const float*)half, (const float*)cell, (const float*)energy, (const float*)dt,
                           (const float*)gamma, (float)(1.0 / mass_eV), (float)(kC / mass_eV), N, bins[0], bins[1], bins[2], (float*)x_out,
                           (const float*)post_map, unpermute, unpermute ? nullptr : (double*)(st + L.sigma), (int*)(st + L.newcount),
                           tg.nt, (int*)(st + L.mis), sums_add, sums_clear, sc_gather_diag());
    else if (dtype == CHX_F32)
        hipLaunchKernelGGL((sc_tile_particle_kernel<float, R>), dim3(nwg), dim3(CHX_BLOCK), 0, s, (const float*)rows, hdr, (int*)(st + L.perm[0]),
                           (float*)(st + L.ws[0]), (float*)(st + L.cs[0]), (const uint16_t*)(st + L.home), (int*)(st + L.cursor),
                           (const float*)phi_halo, (const float*)half, (const float*)cell, (const float*)energy, (const float*)dt,
                           (const float*)gamma, mass_eV, N, bins[0], bins[1], bins[2], (float*)x_out, (const float*)post_map, unpermute,
                           unpermute ? nullptr : (double*)(st + L.sigma), (int*)(st + L.newcount), tg.nt, (int*)(st + L.mis), sums_add,
                           sums_clear);
    else
        hipLaunchKernelGGL((sc_tile_particle_kernel<double, R>), dim3(nwg), dim3(CHX_BLOCK), 0, s, (const double*)rows, hdr,
                           (int*)(st + L.perm[0]), (double*)(st + L.ws[0]), (double*)(st + L.cs[0]), (const uint16_t*)(st + L.home),
                           (int*)(st + L.cursor), (const double*)phi_halo, (const double*)half, (const double*)cell,
                           (const double*)energy, (const double*)dt, (const double*)gamma, mass_eV, N, bins[0], bins[1], bins[2],
                           (double*)x_out, (const double*)post_map, unpermute, unpermute ? nullptr : (double*)(st + L.sigma),
                           (int*)(st + L.newcount), tg.nt, (int*)(st + L.mis), sums_add, sums_clear);
    CHX_CHECK_LAUNCH();
    return CHX_OK;
}

extern "C" int chx_to_xyz_pxpypz(const void* x_in, const void* energy, double mass_eV, int64_t B,
                                 int64_t Bx, int64_t Be, int64_t N, int dtype, void* xp_out, void* stream) {
    return launch_particle<1>(x_in, nullptr, nullptr, nullptr, energy, nullptr, mass_eV, B, Bx, Be, N,
                              nullptr, dtype, xp_out, stream);
}

extern "C" int chx_from_xyz_pxpypz(const void* xp_in, const void* energy, double mass_eV, int64_t B,
                                   int64_t Bx, int64_t Be, int64_t N, int dtype, void* x_out, void* stream) {
    return launch_particle<2>(xp_in, nullptr, nullptr, nullptr, energy, nullptr, mass_eV, B, Bx, Be, N,
                              nullptr, dtype, x_out, stream);
}

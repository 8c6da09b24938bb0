// fused_mfma.hip — does the matrix core help the compute-bound fused run (E 7x7 maps applied to a particle kept in
// registers, chx_track_fused)? north_star reserves MFMA "for the batched small-matrix x particle-block contraction";
// round 1 argued it away on paper, this measures it.
//
//   valu     one particle per lane, 6 rows x 7 FMAs per element, map entries through the scalar cache (what the product does,
//            minus its v_pk_fma pairing)
//   mfma     v_mfma_f32_4x4x1_16b_f32: block = 4 particles, A = a column of 4 map rows, B = coordinate j of the lane's
//            particle, D = 4 output rows of the lane's particle; 2 row groups x 7 columns = 14 MFMAs per element and wave;
//            the accumulators of one element are the B operands of the next (no shuffles)
//   hybrid   even waves run `mfma`, odd waves `valu`, so that the matrix pipe and the vector pipe of a SIMD work at the same
//            time (MI355X_MICROARCH.md: separate pipes, "both ~max, not sum" across waves)
// All three produce the same bits (f32 MFMA = k-ordered fmaf chain). flop accounting of bench.py: 98 per particle-element.
// Build: hipcc --offload-arch=gfx950 -O3 -o fused_mfma benchmarks/fused_mfma.hip
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));
constexpr int E = 100;

__device__ __forceinline__ void valu_element(const float* __restrict__ R, float (&x)[7]) {
    float y[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        float acc = R[i * 7] * x[0];
#pragma unroll
        for (int j = 1; j < 7; ++j) acc = fmaf(R[i * 7 + j], x[j], acc);
        y[i] = acc;
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) x[i] = y[i];
}

// A table: At[e][g][j][r] = R_e[4 g + r][j] (row 7 = 0)
__device__ __forceinline__ void mfma_element(const float* __restrict__ At /* LDS, 56 floats of this element */, int r4,
                                             float (&x)[7]) {
    v4f d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        const float a0 = At[j * 4 + r4], a1 = At[28 + j * 4 + r4];
        d0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a0, x[j], d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a1, x[j], d1, 0, 0, 0);
    }
    x[0] = d0.x; x[1] = d0.y; x[2] = d0.z; x[3] = d0.w;
    x[4] = d1.x; x[5] = d1.y; x[6] = d1.z;
}

// dual: every lane carries TWO particles — one advanced on the matrix pipe, one on the vector pipe — in the same instruction
// stream, so that a single wave keeps both pipes of its SIMD busy (the compiler interleaves the two independent chains)
template <int NV>   // NV particles per lane on the vector pipe (1 or 2) next to one on the matrix pipe
__global__ __launch_bounds__(256) void dual_kernel(const float* __restrict__ xin, const float* __restrict__ R,
                                                   const float* __restrict__ At, float* __restrict__ xout, long N) {
    __shared__ float at[E * 56];
    for (int i = threadIdx.x; i < E * 56; i += 256) at[i] = At[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    constexpr int PP = 1 + NV;
    const long per_block = 256L * PP;
    for (long base = (long)blockIdx.x * per_block; base < N; base += (long)gridDim.x * per_block) {
        float xm[7], xv[NV][7];
        long nm = base + threadIdx.x;
        const long nmc = nm < N ? nm : N - 1;
#pragma unroll
        for (int j = 0; j < 7; ++j) xm[j] = xin[nmc * 7 + j];
        long nv[NV];
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            nv[u] = base + 256L * (1 + u) + threadIdx.x;
            const long c = nv[u] < N ? nv[u] : N - 1;
#pragma unroll
            for (int j = 0; j < 7; ++j) xv[u][j] = xin[c * 7 + j];
        }
        for (int e = 0; e < E; ++e) {
            mfma_element(at + e * 56, lane & 3, xm);
#pragma unroll
            for (int u = 0; u < NV; ++u) valu_element(R + e * 49, xv[u]);
        }
        if (nm < N) {
#pragma unroll
            for (int j = 0; j < 7; ++j) xout[nm * 7 + j] = xm[j];
        }
#pragma unroll
        for (int u = 0; u < NV; ++u)
            if (nv[u] < N) {
#pragma unroll
                for (int j = 0; j < 7; ++j) xout[nv[u] * 7 + j] = xv[u][j];
            }
    }
}

template <int MODE>  // 0 valu, 1 mfma, 2 hybrid
__global__ __launch_bounds__(256) void fused_kernel(const float* __restrict__ xin, const float* __restrict__ R /*[E][49]*/,
                                                    const float* __restrict__ At /*[E][56]*/, float* __restrict__ xout, long N) {
    __shared__ float at[E * 56];
    if (MODE != 0) {
        for (int i = threadIdx.x; i < E * 56; i += 256) at[i] = At[i];
        __syncthreads();
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool use_mfma = MODE == 1 || (MODE == 2 && (wave & 1) == 0);
    for (long n = (long)blockIdx.x * 256 + threadIdx.x; n < N + 255; n += (long)gridDim.x * 256) {
        const long nn = n < N ? n : N - 1;     // whole waves stay converged (MFMA needs EXEC all ones)
        float x[7];
#pragma unroll
        for (int j = 0; j < 7; ++j) x[j] = xin[nn * 7 + j];
        if (use_mfma) {
            for (int e = 0; e < E; ++e) mfma_element(at + e * 56, lane & 3, x);
        } else {
            for (int e = 0; e < E; ++e) valu_element(R + e * 49, x);
        }
        if (n < N) {
#pragma unroll
            for (int j = 0; j < 7; ++j) xout[n * 7 + j] = x[j];
        }
        if (n >= N) break;
    }
}

int main() {
    const long N = 1000000;
    std::vector<float> hx(N * 7), hR(E * 49), hA(E * 56, 0.f);
    srand(1);
    for (long n = 0; n < N; ++n) { for (int j = 0; j < 6; ++j) hx[n * 7 + j] = (rand() / (float)RAND_MAX - 0.5f) * 1e-3f; hx[n * 7 + 6] = 1.f; }
    for (int e = 0; e < E; ++e) {
        // FODO-like: alternating focusing / defocusing thin-ish maps with an affine kick, determinant ~1
        float* R = &hR[e * 49];
        for (int i = 0; i < 7; ++i) for (int j = 0; j < 7; ++j) R[i * 7 + j] = (i == j) ? 1.f : 0.f;
        const float k = (e % 2 ? -0.3f : 0.3f), L = 0.4f;
        R[0 * 7 + 1] = L; R[1 * 7 + 0] = -k; R[1 * 7 + 1] = 1.f - k * L; R[2 * 7 + 3] = L; R[3 * 7 + 2] = k; R[3 * 7 + 3] = 1.f + k * L;
        R[4 * 7 + 5] = -1e-4f; R[0 * 7 + 6] = 1e-6f * (e % 3); R[2 * 7 + 6] = -1e-6f;
        for (int g = 0; g < 2; ++g) for (int j = 0; j < 7; ++j) for (int r = 0; r < 4; ++r)
            if (4 * g + r < 7) hA[e * 56 + g * 28 + j * 4 + r] = R[(4 * g + r) * 7 + j];
    }
    float *x, *R, *At, *y0, *y1;
    CK(hipMalloc(&x, N * 28)); CK(hipMalloc(&y0, N * 28)); CK(hipMalloc(&y1, N * 28));
    CK(hipMalloc(&R, hR.size() * 4)); CK(hipMalloc(&At, hA.size() * 4));
    CK(hipMemcpy(x, hx.data(), N * 28, hipMemcpyHostToDevice));
    CK(hipMemcpy(R, hR.data(), hR.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(At, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> ref(N * 7), got(N * 7);
    const char* names[5] = {"valu", "mfma 4x4x1_16b", "hybrid (even waves mfma)", "dual 1 mfma + 1 valu / lane", "dual 1 mfma + 2 valu / lane"};
    for (int grid : {1024, 2048, 3907}) {
        for (int mode = 0; mode < 5; ++mode) {
            float* y = mode == 0 ? y0 : y1;
            auto launch = [&] {
                if (mode == 0) fused_kernel<0><<<grid, 256>>>(x, R, At, y, N);
                else if (mode == 1) fused_kernel<1><<<grid, 256>>>(x, R, At, y, N);
                else if (mode == 2) fused_kernel<2><<<grid, 256>>>(x, R, At, y, N);
                else if (mode == 3) dual_kernel<1><<<grid / 2 + 1, 256>>>(x, R, At, y, N);
                else dual_kernel<2><<<grid / 3 + 1, 256>>>(x, R, At, y, N);
            };
            launch(); launch();
            CK(hipDeviceSynchronize());
            float best = 1e30f;
            for (int it = 0; it < 10; ++it) {
                CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                best = ms < best ? ms : best;
            }
            CK(hipGetLastError());
            long bad = 0;
            if (mode == 0) CK(hipMemcpy(ref.data(), y, N * 28, hipMemcpyDeviceToHost));
            else {
                CK(hipMemcpy(got.data(), y, N * 28, hipMemcpyDeviceToHost));
                for (long i = 0; i < N * 7; ++i) bad += memcmp(&got[i], &ref[i], 4) != 0 && !(got[i] == 0.f && ref[i] == 0.f);
            }
            printf("grid %4d  %-26s %8.1f us  %6.1f TFLOP/s (98 flop / particle-element)  differing values vs valu: %ld\n", grid,
                   names[mode], best * 1e3, 98.0 * N * E / (best * 1e-3) / 1e12, bad);
        }
    }
    return 0;
}

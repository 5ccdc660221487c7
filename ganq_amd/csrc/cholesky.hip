// Lower Cholesky factorisation A = L L^T in fp32, in place, for the prologue inherited from GPTQ (reference
// gptq.py:280-309: torch.linalg.cholesky of the damped Hessian; the S-solve consumes L, the k-means weights and the
// per-weight losses consume the diagonal of the upper factor of H^-1).
//
// Blocked right-looking, block 128, three kernels per block column j (everything stays on the stream, no host sync):
//   chol_diag_kernel   one workgroup: the 128x128 diagonal block is factored in LDS (column by column, IEEE sqrt and
//                      divide) and its inverse X = L11^-1 is formed by forward substitution, all columns at once;
//   chol_abt_kernel<0> panel:   L21 = A21 X^T          (one 128-row block per workgroup, in place)
//   chol_abt_kernel<1> update:  A22 -= L21 L21^T       (lower-triangular 128x128 tiles)
// Both products are C = A B^T with K = 128 held entirely in LDS, v_mfma_f32_32x32x2_f32, 4 waves x (64x64).
// A non-positive pivot sets *info (1-based column, like LAPACK) and poisons the factor with NaN; the host wrapper
// reads info once at the end.
#include "common.h"

namespace ganq {

constexpr int CB = 128;
constexpr int CP = CB + 1;  // LDS row pitch: lanes walking down a column hit distinct banks

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int CDT = 1024;          // threads of the diagonal-block kernel: its cost is barriers, not arithmetic
constexpr int CDP = CDT / CB;      // k-range parts per column in the inverse
__global__ __launch_bounds__(CDT) void chol_diag_kernel(float* __restrict__ A, int64_t lda, int j, int nb,
                                                         float* __restrict__ Xout, int* __restrict__ info) {
    extern __shared__ __align__(16) float sm[];
    float(*S)[CP] = reinterpret_cast<float(*)[CP]>(sm);            // the block, lower part
    float(*X)[CP] = reinterpret_cast<float(*)[CP]>(sm + CB * CP);  // its inverse
    __shared__ float red[CDP][CB];
    const int tid = threadIdx.x;
    for (int i = tid; i < CB * CB; i += CDT) {
        const int r = i / CB, c = i % CB;
        S[r][c] = (r < nb && c <= r) ? A[(int64_t)(j + r) * lda + j + c] : (r == c ? 1.0f : 0.0f);
        X[r][c] = 0.0f;
    }
    const int ty = tid >> 5, tx = tid & 31;
    for (int c = 0; c < nb; ++c) {
        __syncthreads();
        const float d = S[c][c];
        if (!(d > 0.0f) && tid == 0) atomicCAS(info, 0, j + c + 1);
        const float dd = sqrtf(d);  // NaN for a negative pivot: the factor is visibly unusable
        __syncthreads();
        if (tid == 0) S[c][c] = dd;
        for (int r = c + 1 + tid; r < nb; r += CDT) S[r][c] = S[r][c] / dd;
        __syncthreads();
        // trailing update of the lower triangle: S[r][c2] -= S[r][c] * S[c2][c], c < c2 <= r
        for (int r = c + 1 + ty; r < nb; r += CDT / 32) {
            const float lr = S[r][c];
            for (int c2 = c + 1 + tx; c2 <= r; c2 += 32) S[r][c2] = fmaf(-lr, S[c2][c], S[r][c2]);
        }
    }
    __syncthreads();
    // X = L^-1 row by row: X[r][r] = 1 / L[r][r], X[r][c] = -(sum_{k=c}^{r-1} L[r][k] X[k][c]) / L[r][r] for c < r.
    // thread -> (column c = tid % 128, part = tid / 128 of the k range), partial sums meet in LDS.
    const int xc = tid & (CB - 1), part = tid >> 7;
    for (int r = 0; r < nb; ++r) {
        float s = 0.0f;
        if (xc < r)
            for (int k = xc + part; k < r; k += CDP) s = fmaf(S[r][k], X[k][xc], s);
        red[part][xc] = s;
        __syncthreads();
        if (part == 0) {
            const float lrr = S[r][r];
            if (xc < r) {
                float t = red[0][xc];
#pragma unroll
                for (int p = 1; p < CDP; ++p) t += red[p][xc];
                X[r][xc] = -t / lrr;
            } else if (xc == r) {
                X[r][r] = 1.0f / lrr;
            }
        }
        __syncthreads();
    }
    for (int i = tid; i < CB * CB; i += CDT) {
        const int r = i / CB, c = i % CB;
        if (r < nb && c < nb) A[(int64_t)(j + r) * lda + j + c] = (c <= r) ? S[r][c] : 0.0f;
        Xout[i] = (r < nb && c < nb) ? X[r][c] : 0.0f;
    }
}

// MODE 0 (panel):  rows R0 + 128*blockIdx.x .. : A21[rows][j:j+nb] <- A21 X^T            (X = Xin, 128x128 row-major)
// MODE 1 (update): tile (bi >= bj) of the trailing matrix at R0: A22[bi][bj] -= L21[bi] L21[bj]^T
template <int MODE>
__global__ __launch_bounds__(256) void chol_abt_kernel(float* __restrict__ A, int64_t lda, int n, int j, int nb,
                                                       const float* __restrict__ Xin) {
    extern __shared__ __align__(16) float sm[];
    float(*As)[CP] = reinterpret_cast<float(*)[CP]>(sm);
    float(*Bs)[CP] = reinterpret_cast<float(*)[CP]>(sm + CB * CP);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int R0 = j + nb;  // first row / column of the trailing part
    int bi, bj;
    if (MODE == 0) {
        bi = blockIdx.x;
        bj = 0;
    } else {
        // linear index -> (bi, bj) with bj <= bi
        const int t = blockIdx.x;
        int b = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
        while ((b + 1) * (b + 2) / 2 <= t) ++b;
        while (b * (b + 1) / 2 > t) --b;
        bi = b;
        bj = t - b * (b + 1) / 2;
    }
    const int ra0 = R0 + CB * bi, rb0 = R0 + CB * bj;
    // stage the operands: 128 rows x 128 k each (k beyond nb and rows beyond n are zero)
    for (int i = tid; i < CB * (CB / 4); i += 256) {
        const int r = i / (CB / 4), k4 = (i % (CB / 4)) * 4;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (ra0 + r < n) {
            const float* p = A + (int64_t)(ra0 + r) * lda + j + k4;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (k4 + e < nb) v[e] = p[e];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) As[r][k4 + e] = v[e];
        float w[4] = {0.f, 0.f, 0.f, 0.f};
        if (MODE == 0) {
            const float* p = Xin + (int64_t)r * CB + k4;
#pragma unroll
            for (int e = 0; e < 4; ++e) w[e] = p[e];
        } else if (rb0 + r < n) {
            const float* p = A + (int64_t)(rb0 + r) * lda + j + k4;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (k4 + e < nb) w[e] = p[e];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) Bs[r][k4 + e] = w[e];
    }
    __syncthreads();
    const int wm = (wv >> 1) * 64, wn = (wv & 1) * 64;
    const int i32 = lane & 31, kk = lane >> 5;
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.0f;
    for (int k = 0; k < CB; k += 2) {
        float av[2], bv[2];
#pragma unroll
        for (int a = 0; a < 2; ++a) av[a] = As[wm + 32 * a + i32][k + kk];
#pragma unroll
        for (int b = 0; b < 2; ++b) bv[b] = Bs[wn + 32 * b + i32][k + kk];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a], bv[b], acc[a][b], 0, 0, 0);
    }
    // C layout of the 32x32 MFMA: column = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int rr = wm + 32 * a + (e & 3) + 8 * (e >> 2) + 4 * kk;
                const int cc = wn + 32 * b + i32;
                const int row = ra0 + rr;
                if (row >= n) continue;
                if (MODE == 0) {
                    if (cc < nb) A[(int64_t)row * lda + j + cc] = acc[a][b][e];
                } else {
                    const int col = rb0 + cc;
                    if (col < n && col <= row) {
                        float* p = A + (int64_t)row * lda + col;
                        *p = *p - acc[a][b][e];
                    }
                }
            }
}

__global__ __launch_bounds__(256) void chol_zero_upper_kernel(float* __restrict__ A, int64_t lda, int n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)n * n) return;
    const int r = (int)(i / n), c = (int)(i % n);
    if (c > r) A[(int64_t)r * lda + c] = 0.0f;
}

}  // namespace ganq

using namespace ganq;

extern "C" size_t ganq_cholesky_workspace_bytes(int64_t n) {
    if (n <= 0) return 0;
    return align_up((size_t)CB * CB * sizeof(float), 256) + 256;
}

extern "C" int ganq_cholesky(float* A, int64_t n, int64_t lda, int32_t* info_out, void* workspace, size_t workspace_bytes,
                             void* stream_) {
    if (n < 0) return fail(-1, "ganq_cholesky: negative n");
    if (n == 0) return 0;
    if (lda < n) return fail(-1, "ganq_cholesky: lda=%lld < n=%lld", (long long)lda, (long long)n);
    if (n > INT32_MAX / 2) return fail(-1, "ganq_cholesky: n too large");
    if (!A || !info_out) return fail(-3, "ganq_cholesky: null pointer");
    const size_t need = ganq_cholesky_workspace_bytes(n);
    if (!workspace || workspace_bytes < need)
        return fail(-4, "ganq_cholesky: workspace %zu B < required %zu B", workspace_bytes, need);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    float* X = static_cast<float*>(workspace);
    const size_t smem = 2 * (size_t)CB * CP * sizeof(float);
    static bool attr = false;
    if (!attr) {
        GANQ_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(chol_diag_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        GANQ_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(chol_abt_kernel<0>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        GANQ_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(chol_abt_kernel<1>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr = true;
    }
    ProfScope prof(KID_CHOLESKY, stream);
    GANQ_HIP_CHECK(hipMemsetAsync(info_out, 0, sizeof(int32_t), stream));
    for (int64_t j = 0; j < n; j += CB) {
        const int nb = (int)std::min<int64_t>(CB, n - j);
        hipLaunchKernelGGL(chol_diag_kernel, dim3(1), dim3(CDT), smem, stream, A, lda, (int)j, nb, X, info_out);
        const int64_t rem = n - j - nb;
        if (rem > 0) {
            const int nblk = (int)((rem + CB - 1) / CB);
            hipLaunchKernelGGL(chol_abt_kernel<0>, dim3(nblk), dim3(256), smem, stream, A, lda, (int)n, (int)j, nb, X);
            hipLaunchKernelGGL(chol_abt_kernel<1>, dim3(nblk * (nblk + 1) / 2), dim3(256), smem, stream, A, lda, (int)n,
                               (int)j, nb, X);
        }
    }
    hipLaunchKernelGGL(chol_zero_upper_kernel, dim3((unsigned)((n * n + 255) / 256)), dim3(256), 0, stream, A, lda, (int)n);
    GANQ_LAUNCH_CHECK();
    return 0;
}

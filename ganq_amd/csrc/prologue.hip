// Prologue of GPTQ.quantize() (reference gptq.py:259-319) as three passes instead of ~10 torch launches and three 64 MiB
// clones: dead columns, act_sort permutation, ganq-style diagonal offset, damping, and the inputs of the two factorisations.
//   ganq_prologue_rowstats  one read of H:   diag[i] = H[i][i],  rowabs[i] = sum_j |H[i][j]|          (gptq.py:267-269, :289-291)
//   (host, on n-vectors:    dead = diag == 0, diag' = dead ? 1 : diag, perm = argsort(diag'), offset, damp -- torch one-liners)
//   ganq_prologue_gather    one read of H, up to three outputs written directly (gptq.py:281-300, :302-308):
//                               out_k[i][j] = H'[perm i][perm j] + (i == j ? add_k[i] : 0),   optionally index-reversed,
//                           H' = H with the dead columns' diagonal set to 1 (gptq.py:268): Xxt_damped (add = damp), the ganq-style
//                           Cholesky input H + diag(offset), and the index-reversed damped matrix whose factor yields diag(Hinv)
//   ganq_prologue_weights   W_out[r][c] = dead[perm c] ? fill_r : W[r][perm c],  fill = 0 or the row mean over live columns
//                           (gptq.py:270-276, :283)
#include "common.h"

namespace ganq {

__global__ __launch_bounds__(256) void prologue_rowstats_kernel(const float* __restrict__ H, int n, float* __restrict__ diag,
                                                                float* __restrict__ rowabs) {
    const int i = blockIdx.x;
    const float* row = H + (int64_t)i * n;
    double s = 0.0;  // fp64 partial sums: the result does not depend on how the row is dealt to the lanes beyond fp64 rounding
    for (int j = threadIdx.x * 4; j < n; j += 256 * 4) {
        if (j + 3 < n && ((reinterpret_cast<uintptr_t>(row + j) & 15) == 0)) {
            const float4 v = *reinterpret_cast<const float4*>(row + j);
            s += (double)fabsf(v.x) + (double)fabsf(v.y) + (double)fabsf(v.z) + (double)fabsf(v.w);
        } else {
            for (int e = j; e < min(j + 4, n); ++e) s += (double)fabsf(row[e]);
        }
    }
    __shared__ double sh[256];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        rowabs[i] = (float)sh[0];
        diag[i] = row[i];
    }
}

// 64 x 64 output tile per workgroup; lanes along j (coalesced stores; the loads gather inside one 16 KB row of H)
__global__ __launch_bounds__(256) void prologue_gather_kernel(const float* __restrict__ H, const int64_t* __restrict__ perm,
                                                              const float* __restrict__ diag_fixed, int n, float* __restrict__ out0,
                                                              const float* __restrict__ add0, int flip0, float* __restrict__ out1,
                                                              const float* __restrict__ add1, int flip1, float* __restrict__ out2,
                                                              const float* __restrict__ add2, int flip2) {
    const int j = blockIdx.x * 64 + (threadIdx.x & 63);
    const int i0 = blockIdx.y * 64 + (threadIdx.x >> 6) * 16;
    if (j >= n) return;
    const int64_t pj = perm ? perm[j] : j;
    for (int i = i0; i < min(i0 + 16, n); ++i) {
        const int64_t pi = perm ? perm[i] : i;
        float h = H[pi * n + pj];
        if (i == j) h = diag_fixed[pi];  // the dead columns' diagonal reads 1 (gptq.py:268)
        if (out0) {
            const float v = h + ((i == j && add0) ? add0[i] : 0.0f);
            out0[flip0 ? (int64_t)(n - 1 - i) * n + (n - 1 - j) : (int64_t)i * n + j] = v;
        }
        if (out1) {
            const float v = h + ((i == j && add1) ? add1[i] : 0.0f);
            out1[flip1 ? (int64_t)(n - 1 - i) * n + (n - 1 - j) : (int64_t)i * n + j] = v;
        }
        if (out2) {
            const float v = h + ((i == j && add2) ? add2[i] : 0.0f);
            out2[flip2 ? (int64_t)(n - 1 - i) * n + (n - 1 - j) : (int64_t)i * n + j] = v;
        }
    }
}

// one workgroup per row of W
__global__ __launch_bounds__(256) void prologue_weights_kernel(const float* __restrict__ W, const int64_t* __restrict__ perm,
                                                               const uint8_t* __restrict__ dead, int n, int mean_fill,
                                                               float* __restrict__ Wout) {
    const int r = blockIdx.x;
    const float* row = W + (int64_t)r * n;
    __shared__ float sh[256];
    __shared__ int shc[256];
    float fill = 0.0f;
    if (mean_fill) {  // torch.mean(W[:, ~dead], dim=1) (gptq.py:273): fp32 sum of the live columns / their count
        float s = 0.0f;
        int c = 0;
        for (int j = threadIdx.x; j < n; j += 256)
            if (!dead[j]) {
                s += row[j];
                ++c;
            }
        sh[threadIdx.x] = s;
        shc[threadIdx.x] = c;
        __syncthreads();
        for (int w = 128; w > 0; w >>= 1) {
            if ((int)threadIdx.x < w) {
                sh[threadIdx.x] += sh[threadIdx.x + w];
                shc[threadIdx.x] += shc[threadIdx.x + w];
            }
            __syncthreads();
        }
        fill = shc[0] > 0 ? sh[0] / (float)shc[0] : __builtin_nanf("");  // no live column: the reference's mean of nothing is NaN
    }
    float* orow = Wout + (int64_t)r * n;
    for (int j = threadIdx.x; j < n; j += 256) {
        const int64_t pj = perm ? perm[j] : j;
        orow[j] = dead[pj] ? fill : row[pj];
    }
}

}  // namespace ganq

using namespace ganq;

extern "C" int ganq_prologue_rowstats(const float* H, int64_t n, float* diag, float* rowabs, void* stream_) {
    if (n < 0) return fail(-1, "ganq_prologue_rowstats: negative n");
    if (n == 0) return 0;
    if (n > INT32_MAX / 2) return fail(-1, "ganq_prologue_rowstats: n too large");
    if (!H || !diag || !rowabs) return fail(-3, "ganq_prologue_rowstats: null pointer");
    hipLaunchKernelGGL(prologue_rowstats_kernel, dim3((unsigned)n), dim3(256), 0, static_cast<hipStream_t>(stream_), H, (int)n, diag,
                       rowabs);
    GANQ_LAUNCH_CHECK();
    return 0;
}

extern "C" int ganq_prologue_gather(const float* H, const int64_t* perm, const float* diag_fixed, int64_t n, float* out0,
                                    const float* add0, int flip0, float* out1, const float* add1, int flip1, float* out2,
                                    const float* add2, int flip2, void* stream_) {
    if (n < 0) return fail(-1, "ganq_prologue_gather: negative n");
    if (n == 0) return 0;
    if (n > INT32_MAX / 2) return fail(-1, "ganq_prologue_gather: n too large");
    if (!H || !diag_fixed || (!out0 && !out1 && !out2)) return fail(-3, "ganq_prologue_gather: null pointer");
    if (out0 == H || out1 == H || out2 == H) return fail(-2, "ganq_prologue_gather: the outputs must not alias H (a permuted gather)");
    const dim3 grid((unsigned)((n + 63) / 64), (unsigned)((n + 63) / 64));
    hipLaunchKernelGGL(prologue_gather_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream_), H, perm, diag_fixed, (int)n, out0,
                       add0, flip0, out1, add1, flip1, out2, add2, flip2);
    GANQ_LAUNCH_CHECK();
    return 0;
}

extern "C" int ganq_prologue_weights(const float* W, const int64_t* perm, const uint8_t* dead, int64_t m, int64_t n, int mean_fill,
                                     float* W_out, void* stream_) {
    if (m < 0 || n < 0) return fail(-1, "ganq_prologue_weights: negative shape");
    if (m == 0 || n == 0) return 0;
    if (m > INT32_MAX / 2 || n > INT32_MAX / 2) return fail(-1, "ganq_prologue_weights: shape too large");
    if (!W || !dead || !W_out) return fail(-3, "ganq_prologue_weights: null pointer");
    if (W == W_out) return fail(-2, "ganq_prologue_weights: W_out must not alias W (a permuted gather)");
    hipLaunchKernelGGL(prologue_weights_kernel, dim3((unsigned)m), dim3(256), 0, static_cast<hipStream_t>(stream_), W, perm, dead, (int)n,
                       mean_fill, W_out);
    GANQ_LAUNCH_CHECK();
    return 0;
}

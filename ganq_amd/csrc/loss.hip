// Loss and output kernels of the GANQ loop.
//   ganq_quad_loss        dist = sum( ((W - Wq) @ H) * (W - Wq) ),  Wq = T.gather(1,Q)   (ganq.py:392-395, :621-622)
//                         E = W - T[Q] (elementwise), EH = E @ H on the fp32 matrix cores (gemm_f32.hip),
//                         then a two-stage ordered reduction in fp64 (deterministic).
//   ganq_dequant_losses   Wq = T.gather(1,Q); Losses = (W - Wq)^2 / diag(Hinv)^2 / 2          (ganq.py:633-638)
#include "common.h"

namespace ganq {

__global__ __launch_bounds__(256) void err_kernel(const float* __restrict__ W, const float* __restrict__ T,
                                                  const uint8_t* __restrict__ Q, int64_t total, int n, int V,
                                                  float* __restrict__ E) {
    // 4 elements per thread; n % 4 == 0 is NOT assumed (row is recomputed per element when the quad straddles rows)
    const int64_t i4 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int64_t i = i4 + k;
        if (i < total) {
            const int64_t row = i / n;
            E[i] = W[i] - T[row * V + Q[i]];
        }
    }
}

constexpr int RED_BLOCKS = 1024;

__global__ __launch_bounds__(256) void dot_partial_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                          int64_t total, double* __restrict__ partial) {
    // block b sums a contiguous slice in a fixed order: thread-strided fp64 partials, then an LDS tree
    __shared__ double sh[256];
    const int64_t per = (total + RED_BLOCKS - 1) / RED_BLOCKS;
    const int64_t lo = (int64_t)blockIdx.x * per;
    const int64_t hi = lo + per < total ? lo + per : total;
    double s = 0.0;
    for (int64_t i = lo + threadIdx.x; i < hi; i += 256) s += (double)A[i] * (double)B[i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = sh[0];
}

__global__ __launch_bounds__(256) void dot_final_kernel(const double* __restrict__ partial, double* __restrict__ out) {
    __shared__ double sh[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < RED_BLOCKS; i += 256) s += partial[i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = sh[0];
}

__global__ __launch_bounds__(256) void dequant_losses_kernel(const float* __restrict__ W, const float* __restrict__ T,
                                                             const uint8_t* __restrict__ Q,
                                                             const float* __restrict__ hinv_diag, int64_t total, int n,
                                                             int V, float* __restrict__ Wq, float* __restrict__ Losses) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int64_t row = i / n;
    const int col = (int)(i - row * n);
    const float wq = T[row * V + Q[i]];
    if (Wq) Wq[i] = wq;
    if (Losses) {
        const float d = hinv_diag[col];
        const float e = W[i] - wq;
        Losses[i] = ((e * e) / (d * d)) / 2.0f;
    }
}

}  // namespace ganq

using namespace ganq;

extern "C" size_t ganq_quad_loss_workspace_bytes(int64_t m, int64_t n, int V) {
    (void)V;
    if (m <= 0 || n <= 0) return 0;
    return 2 * align_up((size_t)m * (size_t)n * sizeof(float), 256) + align_up(RED_BLOCKS * sizeof(double), 256);
}

extern "C" int ganq_quad_loss(const float* W, const float* H, const float* T, const uint8_t* Q, int64_t m, int64_t n,
                              int V, double* loss_out, void* workspace, size_t workspace_bytes, void* stream_) {
    if (m < 0 || n < 0) return fail(-1, "ganq_quad_loss: negative shape");
    if (!loss_out) return fail(-3, "ganq_quad_loss: null loss_out");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (m == 0 || n == 0) {
        GANQ_HIP_CHECK(hipMemsetAsync(loss_out, 0, sizeof(double), stream));
        return 0;
    }
    if (V < 2 || V > 256) return fail(-2, "ganq_quad_loss: bad V=%d", V);
    if (!W || !H || !T || !Q) return fail(-3, "ganq_quad_loss: null pointer");
    const size_t need = ganq_quad_loss_workspace_bytes(m, n, V);
    if (!workspace || workspace_bytes < need)
        return fail(-4, "ganq_quad_loss: workspace %zu B < required %zu B", workspace_bytes, need);
    char* ws = static_cast<char*>(workspace);
    const size_t mat = align_up((size_t)m * (size_t)n * sizeof(float), 256);
    float* E = reinterpret_cast<float*>(ws);
    float* EH = reinterpret_cast<float*>(ws + mat);
    double* partial = reinterpret_cast<double*>(ws + 2 * mat);
    const int64_t total = m * n;
    {
        ProfScope prof(KID_ERR, stream);
        hipLaunchKernelGGL(err_kernel, dim3((unsigned)((total + 1023) / 1024)), dim3(256), 0, stream, W, T, Q, total, (int)n,
                           V, E);
    }
    GANQ_LAUNCH_CHECK();
    int rc = ganq_matmul_f32(E, H, m, n, n, EH, stream_);
    if (rc) return rc;
    {
        ProfScope prof(KID_DOT, stream);
        hipLaunchKernelGGL(dot_partial_kernel, dim3(RED_BLOCKS), dim3(256), 0, stream, EH, E, total, partial);
        hipLaunchKernelGGL(dot_final_kernel, dim3(1), dim3(256), 0, stream, partial, loss_out);
    }
    GANQ_LAUNCH_CHECK();
    return 0;
}

extern "C" int ganq_dequant_losses(const float* W, const float* T, const uint8_t* Q, const float* hinv_diag, int64_t m,
                                   int64_t n, int V, float* Wq_out, float* Losses_out, void* stream_) {
    if (m < 0 || n < 0) return fail(-1, "ganq_dequant_losses: negative shape");
    if (m == 0 || n == 0) return 0;
    if (!T || !Q) return fail(-3, "ganq_dequant_losses: null pointer");
    if (Losses_out && (!W || !hinv_diag)) return fail(-3, "ganq_dequant_losses: Losses needs W and hinv_diag");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int64_t total = m * n;
    ProfScope prof(KID_DEQUANT, stream);
    hipLaunchKernelGGL(dequant_losses_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, W, T, Q,
                       hinv_diag, total, (int)n, V, Wq_out, Losses_out);
    GANQ_LAUNCH_CHECK();
    return 0;
}

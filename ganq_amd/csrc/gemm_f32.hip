// Dense fp32 GEMM on the fp32 matrix cores (v_mfma_f32_32x32x2_f32): C[M,N] = A[M,K] @ B[K,N],
// all row-major.  Used for W@H (reference ganq.py:590) and (W-Wq)@H (ganq.py:394).
// 128x128 workgroup tile, 4 waves (2x2), each wave 64x64 = 2x2 MFMA tiles; K in slabs of 16
// through double-buffered LDS (A slab stored k-major so both operand reads are conflict-free
// 128 B rows).
#include "common.h"

namespace ganq {

constexpr int GM = 128, GN = 128, GK = 16;

__device__ __forceinline__ int xcd_swizzle(int bid, int nblocks) {
    // consecutive logical tiles -> same XCD (blocks are dealt round-robin over the 8 XCDs), so the
    // tiles that share an A row-panel hit one L2.  Speed only.
    const int per = nblocks >> 3;
    if (per == 0 || (nblocks & 7)) return bid;
    return (bid & 7) * per + (bid >> 3);
}

__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                       float* __restrict__ C, int M, int N, int K) {
    __shared__ float As[2][GK][GM + 4];  // [k][m]
    __shared__ float Bs[2][GK][GN + 4];  // [k][n]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = tid >> 6;
    const int tiles_n = (N + GN - 1) / GN;
    const int tiles_m = (M + GM - 1) / GM;
    const int bid = xcd_swizzle(blockIdx.x, tiles_m * tiles_n);
    const int bm = (bid / tiles_n) * GM;
    const int bn = (bid % tiles_n) * GN;
    const int wm = (wv >> 1) * 64;
    const int wn = (wv & 1) * 64;

    // global -> register staging: A slab 128 rows x 16 k (2 float4 per thread), B slab 16 k x 128 n
    const int a_row = tid >> 1;            // 0..127
    const int a_k4 = (tid & 1) * 8;        // 0 or 8 (two float4: k .. k+7)
    const int b_k = tid >> 4;              // 0..15
    const int b_n8 = (tid & 15) * 8;       // 0..120
    float4 ra[2], rb[2];

    auto gload = [&](int k0) {
        const int row = bm + a_row;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int k = k0 + a_k4 + 4 * h;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < M) {
                const float* p = A + (int64_t)row * K + k;
                if (k + 3 < K && ((reinterpret_cast<uintptr_t>(p) & 15) == 0)) {
                    v = *reinterpret_cast<const float4*>(p);
                } else {
                    if (k + 0 < K) v.x = p[0];
                    if (k + 1 < K) v.y = p[1];
                    if (k + 2 < K) v.z = p[2];
                    if (k + 3 < K) v.w = p[3];
                }
            }
            ra[h] = v;
        }
        const int kb = k0 + b_k;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int col = bn + b_n8 + 4 * h;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (kb < K) {
                const float* p = B + (int64_t)kb * N + col;
                if (col + 3 < N && ((reinterpret_cast<uintptr_t>(p) & 15) == 0)) {
                    v = *reinterpret_cast<const float4*>(p);
                } else {
                    if (col + 0 < N) v.x = p[0];
                    if (col + 1 < N) v.y = p[1];
                    if (col + 2 < N) v.z = p[2];
                    if (col + 3 < N) v.w = p[3];
                }
            }
            rb[h] = v;
        }
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int k = a_k4 + 4 * h;
            As[buf][k + 0][a_row] = ra[h].x;
            As[buf][k + 1][a_row] = ra[h].y;
            As[buf][k + 2][a_row] = ra[h].z;
            As[buf][k + 3][a_row] = ra[h].w;
            *reinterpret_cast<float4*>(&Bs[buf][b_k][b_n8 + 4 * h]) = rb[h];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int nk = (K + GK - 1) / GK;
    gload(0);
    sstore(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gload((kt + 1) * GK);
#pragma unroll
        for (int kk = 0; kk < GK; kk += 2) {
            const int kq = kk + (lane >> 5);
            float a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = As[buf][kq][wm + 32 * i + (lane & 31)];
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = Bs[buf][kq][wn + 32 * j + (lane & 31)];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) sstore(buf ^ 1);
        __syncthreads();
    }

    // C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = bm + wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int col = bn + wn + 32 * j + (lane & 31);
                if (row < M && col < N) C[(int64_t)row * N + col] = acc[i][j][r];
            }
}

}  // namespace ganq

using namespace ganq;

extern "C" int ganq_matmul_f32(const float* A, const float* B, int64_t m, int64_t k, int64_t n, float* C,
                               void* stream_) {
    if (m < 0 || n < 0 || k < 0) return fail(-1, "ganq_matmul_f32: negative shape");
    if (m == 0 || n == 0) return 0;
    if (m > INT32_MAX / 2 || n > INT32_MAX / 2 || k > INT32_MAX / 2) return fail(-1, "ganq_matmul_f32: shape too large");
    if (!A || !B || !C) return fail(-3, "ganq_matmul_f32: null pointer");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int tiles = (int)(((m + GM - 1) / GM) * ((n + GN - 1) / GN));
    ProfScope prof(KID_GEMM_F32, stream);
    hipLaunchKernelGGL(gemm_f32_kernel, dim3(tiles), dim3(256), 0, stream, A, B, C, (int)m, (int)n, (int)k);
    GANQ_LAUNCH_CHECK();
    return 0;
}

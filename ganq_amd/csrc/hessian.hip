// Hessian accumulation (reference gptq.py:96-131 process_batch):
//     H <- H * N/(N+b) + (2/(N+b)) * X^T X        X [rows, n] fp16 or bf16, H [n,n] fp32
// (the reference scales X by sqrt(2/N') in fp32 and multiplies; fp16 x fp16 products are exact in fp32, so
// summing raw products and scaling once differs only by fp32 summation order.)
//
// v_mfma_f32_32x32x16_{f16,bf16}, 128x128 output tile per workgroup (4 waves x 64x64), tokens streamed in
// slabs of 32 through LDS in their natural [token][feature] layout; both MFMA operands need 8 consecutive
// TOKENS per lane, which ds_read_b64_tr_b16 delivers straight from that layout (hardware transpose).
// Only tiles on or below the diagonal are computed; the mirror image is written from the same registers,
// so H is exactly symmetric.
#include "common.h"

namespace ganq {

constexpr int HT = 128;       // output tile edge
constexpr int HK = 32;        // tokens per slab
constexpr int HP = HT + 8;    // LDS row pitch in 16-bit elements (272 B, multiple of 8 B)

typedef short s4v __attribute__((__vector_size__(4 * sizeof(short))));
typedef short s8v __attribute__((ext_vector_type(8)));
typedef _Float16 h8v __attribute__((ext_vector_type(8)));
typedef __bf16 b8v __attribute__((ext_vector_type(8)));

template <bool BF16>
__device__ __forceinline__ f32x16 mfma16(s8v a, s8v b, f32x16 c) {
    if constexpr (BF16) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(b8v, a), __builtin_bit_cast(b8v, b), c, 0, 0, 0);
    } else {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8v, a), __builtin_bit_cast(h8v, b), c, 0, 0, 0);
    }
}

template <bool BF16>
__global__ __launch_bounds__(256) void hessian_kernel(float* __restrict__ H, const uint16_t* __restrict__ X, int rows,
                                                      int n, float decay, float scale, int tiles_per_side) {
    __shared__ __align__(16) uint16_t Xa[2][HK][HP];
    __shared__ __align__(16) uint16_t Xb[2][HK][HP];

    // blockIdx.x enumerates the lower-triangular tile pairs (tu >= tv)
    int tu = 0, tv = 0;
    {
        int b = blockIdx.x;
        // row-major over the triangle: tile row tu has tu+1 entries
        int r = (int)((sqrtf(8.0f * (float)b + 1.0f) - 1.0f) * 0.5f);
        while ((r + 1) * (r + 2) / 2 <= b) ++r;
        while (r * (r + 1) / 2 > b) --r;
        tu = r;
        tv = b - r * (r + 1) / 2;
    }
    (void)tiles_per_side;
    const int u0 = tu * HT, v0 = tv * HT;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wm = (wv >> 1) * 64, wn = (wv & 1) * 64;

    // staging: slab = 32 tokens x 128 features = 512 x 16 B per operand, 2 per thread
    uint4 ra[2], rb[2];
    auto gload = [&](int t0) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int idx = h * 256 + tid;
            const int t = t0 + (idx >> 4), f8 = (idx & 15) * 8;
            uint4 va = make_uint4(0, 0, 0, 0), vb = make_uint4(0, 0, 0, 0);
            if (t < rows) {
                const uint16_t* pa = X + (int64_t)t * n + u0 + f8;
                const uint16_t* pb = X + (int64_t)t * n + v0 + f8;
                if (u0 + f8 + 7 < n && ((reinterpret_cast<uintptr_t>(pa) & 15) == 0)) {
                    va = *reinterpret_cast<const uint4*>(pa);
                } else {
                    uint16_t tmp[8];
                    for (int e = 0; e < 8; ++e) tmp[e] = (u0 + f8 + e < n) ? pa[e] : (uint16_t)0;
                    va = *reinterpret_cast<uint4*>(tmp);
                }
                if (v0 + f8 + 7 < n && ((reinterpret_cast<uintptr_t>(pb) & 15) == 0)) {
                    vb = *reinterpret_cast<const uint4*>(pb);
                } else {
                    uint16_t tmp[8];
                    for (int e = 0; e < 8; ++e) tmp[e] = (v0 + f8 + e < n) ? pb[e] : (uint16_t)0;
                    vb = *reinterpret_cast<uint4*>(tmp);
                }
            }
            ra[h] = va;
            rb[h] = vb;
        }
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int idx = h * 256 + tid;
            const int t = idx >> 4, f8 = (idx & 15) * 8;
            *reinterpret_cast<uint4*>(&Xa[buf][t][f8]) = ra[h];
            *reinterpret_cast<uint4*>(&Xb[buf][t][f8]) = rb[h];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // transposed-read addressing (cdna guide T10): per 16-lane group, lane 4q+p supplies row q, columns 4p..4p+3
    // and receives column (lane & 15), rows 0..3.  MFMA operand lane l: row/col = l & 31, k = 8*(l >> 5) + j.
    const int g16 = (lane >> 4) & 1, kh = lane >> 5, q = (lane & 15) >> 2, p = lane & 3;
    auto frag = [&](const uint16_t (*S)[HP], int kk, int base) -> s8v {
        const uint16_t* a0 = &S[kk + 8 * kh + q][base + 16 * g16 + 4 * p];
        const s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)a0);
        const s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)(a0 + 4 * HP));
        s8v out = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return out;
    };

    const int nslab = (rows + HK - 1) / HK;
    gload(0);
    sstore(0);
    __syncthreads();
    for (int s = 0; s < nslab; ++s) {
        const int buf = s & 1;
        if (s + 1 < nslab) gload((s + 1) * HK);
#pragma unroll
        for (int kk = 0; kk < HK; kk += 16) {
            s8v a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = frag(Xa[buf], kk, wm + 32 * i);
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = frag(Xb[buf], kk, wn + 32 * j);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = mfma16<BF16>(a[i], b[j], acc[i][j]);
        }
        if (s + 1 < nslab) sstore(buf ^ 1);
        __syncthreads();
    }

#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int u = u0 + wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int v = v0 + wn + 32 * j + (lane & 31);
                if (u < n && v < n && (tu != tv || u >= v)) {
                    const int64_t o = (int64_t)u * n + v;
                    const float old = (decay != 0.0f) ? H[o] * decay : 0.0f;
                    const float val = old + scale * acc[i][j][r];
                    H[o] = val;
                    if (u != v) H[(int64_t)v * n + u] = val;
                }
            }
}

}  // namespace ganq

using namespace ganq;

extern "C" int ganq_hessian_accum(float* H, const void* X, int dtype, int64_t rows, int64_t n, int64_t nsamples_before,
                                  int64_t batch, void* stream_) {
    if (rows < 0 || n < 0 || nsamples_before < 0 || batch <= 0) return fail(-1, "ganq_hessian_accum: bad sizes");
    if (n == 0) return 0;
    if (dtype != 0 && dtype != 1) return fail(-2, "ganq_hessian_accum: dtype %d (0 = fp16, 1 = bf16)", dtype);
    if (!H || (!X && rows > 0)) return fail(-3, "ganq_hessian_accum: null pointer");
    if (n > INT32_MAX / 2 || rows > INT32_MAX / 2) return fail(-1, "ganq_hessian_accum: shape too large");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const double total = (double)(nsamples_before + batch);
    const float decay = (float)((double)nsamples_before / total);
    const float scale = (float)(2.0 / total);
    const int tiles = (int)((n + HT - 1) / HT);
    const int blocks = tiles * (tiles + 1) / 2;
    ProfScope prof(KID_HESSIAN, stream);
    if (dtype == 1) {
        hipLaunchKernelGGL(hessian_kernel<true>, dim3(blocks), dim3(256), 0, stream, H, static_cast<const uint16_t*>(X),
                           (int)rows, (int)n, decay, scale, tiles);
    } else {
        hipLaunchKernelGGL(hessian_kernel<false>, dim3(blocks), dim3(256), 0, stream, H, static_cast<const uint16_t*>(X),
                           (int)rows, (int)n, decay, scale, tiles);
    }
    GANQ_LAUNCH_CHECK();
    return 0;
}

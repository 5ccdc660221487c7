// W @ H_fixed for the T-update (reference ganq.py:590, `W @ H` in fp32) on the fp16 matrix cores.
//
// Both operands are split into two fp16 pieces, x = hi + lo with hi = fp16(x), lo = fp16(x - hi): 22 significant bits
// per element (the reference multiplies fp32 values, 24 bits), after exact power-of-two scalings that keep everything
// inside fp16's 5 exponent bits whatever the dynamic range of the activations:
//   * H is scaled SYMMETRICALLY, H~[u][v] = H[u][v] 2^(-e_u - e_v) with e_u = floor(log2(H_uu) / 2): its diagonal lies in
//     [1, 4) and, H being positive semi-definite, no entry exceeds 4 -- a massive-activation feature (H_ii 10^6 x the
//     typical one) would otherwise push the typical row of H into fp16's subnormals;
//   * W takes the inverse column scaling, W~[i][v] = W[i][v] 2^(e_v), then a per-row scale puts the row maximum at
//     2^14..2^15;  W H = (W~ H~) with column u of the result scaled back by 2^(e_u).
//   * the DIAGONAL of H is taken out of the matrix product and added exactly in fp64, W H = W offdiag(H) + W diag(H): the
//     term w_u H_uu is the bulk of (W H)[i][u], and the closed-form loss of the T-update (dist = w^T H w - 2 t^T b + t^T A t,
//     a difference ~10^3 smaller than its terms) magnifies whatever rounding it carries.
// The product is
//     W H  ~=  Whi Hhi + Whi Hlo + Wlo Hhi          (Wlo Hlo is below 2^-22 of the result and dropped),
// three v_mfma_f32_32x32x16_f16 per fragment pair into one fp32 accumulator; products of fp16 values are exact in
// fp32, so the rounding left is that of the fp32 accumulation -- the same class of error as the reference's fp32 GEMM;
// every WFLUSH k tiles (256 columns) the fp32 accumulators are added into fp64 ones and cleared, which keeps the
// accumulation error 4x below that of one fp32 chain over n = 4096 (the closed-form loss of the T-update magnifies the
// error of W H by ~10^3: dist = w^T H w - 2 t^T b + t^T A t).  A module whose weights are fp16
// values has Wlo == 0: the split kernel records that per 128-row block and the third product and its loads are skipped.
// H is symmetric, so both operands are read "row x k" and k is the contiguous direction of both.
//
// Data layout: the split kernels write the pieces in the order the GEMM reads them -- per (128-row block, 32-deep
// k tile) one 8 KB image per piece, row r at byte 64 r, its four 16-byte chunks (8 k each) stored at chunk index
// c ^ ((r >> 2) & 3).  A workgroup copies 32 KB per k tile linearly (16 B per lane, fully coalesced) into LDS, and the
// swizzle makes the ds_read_b128 of an MFMA operand (32 rows x 16 B at one chunk index) hit all 64 banks once.
// 128x128 output tile per workgroup, 4 waves x 64x64, double-buffered LDS (64 KB), two workgroups per CU.
// The kernel is bound by the L2 -> LDS stream (4 GB at 4096^2), not by the matrix cores.
#include "common.h"
#include "wh_gemm.h"

namespace ganq {

typedef _Float16 h8v __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef uint32_t u4v __attribute__((ext_vector_type(4)));

constexpr int WT = 128;               // rows per operand block
constexpr int WK = 32;                // k per tile
constexpr int WIMG = WT * WK * 2;     // bytes of one piece of one tile (8 KB)
constexpr int WFLUSH = 8;             // k tiles between two flushes of the fp32 accumulators into fp64 (power of two)

__device__ __forceinline__ int64_t wh_chunk_offset(int r, int c) {  // inside one piece image
    return (int64_t)r * 64 + ((c ^ ((r >> 2) & 3)) << 4);
}

// one workgroup per (padded) row of W: column scaling 2^(dexp[u]), row maximum -> power-of-two scale, then the two
// pieces of every 8-k chunk
__global__ __launch_bounds__(256) void wh_split_w_kernel(const float* __restrict__ W, const int* __restrict__ dexp, int m, int n,
                                                        int KT, char* __restrict__ Wp, int* __restrict__ rexp,
                                                        int* __restrict__ wlo_any) {
    __shared__ float sh[4];
    const int r = blockIdx.x, rb = r >> 7, rr = r & 127;
    const float* w = W + (int64_t)min(r, m - 1) * n;
    // 16-byte loads where the row allows them (every layer shape in use: n a multiple of 8, 16-byte aligned base): with scalar
    // loads the two split kernels took 170 us of the 0.7 ms the T-update's preparation costs per layer
    const bool vec = (n & 7) == 0 && (reinterpret_cast<uintptr_t>(W) & 15) == 0 && (reinterpret_cast<uintptr_t>(dexp) & 15) == 0;
    float mx = 0.0f;
    if (r < m) {
        if (vec) {
            for (int u4 = threadIdx.x; u4 < n / 4; u4 += 256) {
                const float4 x = reinterpret_cast<const float4*>(w)[u4];
                const int4 e = reinterpret_cast<const int4*>(dexp)[u4];
                mx = fmaxf(fmaxf(mx, fmaxf(fabsf(ldexpf(x.x, e.x)), fabsf(ldexpf(x.y, e.y)))),
                           fmaxf(fabsf(ldexpf(x.z, e.z)), fabsf(ldexpf(x.w, e.w))));
            }
        } else {
            for (int u = threadIdx.x; u < n; u += 256) mx = fmaxf(mx, fabsf(ldexpf(w[u], dexp[u])));
        }
    }
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
    int sft = 0;
    if (mx > 0.0f && mx < __builtin_inff()) sft = 14 - max(ilogbf(mx), -100);  // mx * 2^sft in [2^14, 2^15)
    if (threadIdx.x == 0 && r < m) rexp[r] = sft;
    bool any = false;
    for (int ci = threadIdx.x; ci < KT * 4; ci += 256) {
        h8v hi, lo;
        float xs8[8];
        int es8[8];
        const bool full = vec && r < m && ci * 8 + 8 <= n;  // (padding pieces beyond n and rows beyond m are zeros)
        if (full) {
            *reinterpret_cast<float4*>(&xs8[0]) = reinterpret_cast<const float4*>(w)[ci * 2];
            *reinterpret_cast<float4*>(&xs8[4]) = reinterpret_cast<const float4*>(w)[ci * 2 + 1];
            *reinterpret_cast<int4*>(&es8[0]) = reinterpret_cast<const int4*>(dexp)[ci * 2];
            *reinterpret_cast<int4*>(&es8[4]) = reinterpret_cast<const int4*>(dexp)[ci * 2 + 1];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int u = ci * 8 + k;
            const float x = full ? ldexpf(xs8[k], es8[k] + sft) : ((r < m && u < n) ? ldexpf(w[u], dexp[u] + sft) : 0.0f);
            const _Float16 h = (_Float16)x;
            const _Float16 l = (_Float16)(x - (float)h);
            hi[k] = h;
            lo[k] = l;
            any |= (l != (_Float16)0.0f);
        }
        char* dst = Wp + ((int64_t)rb * KT + (ci >> 2)) * (2 * WIMG) + wh_chunk_offset(rr, ci & 3);
        *reinterpret_cast<h8v*>(dst) = hi;
        *reinterpret_cast<h8v*>(dst + WIMG) = lo;
    }
    if (__syncthreads_or(any) && threadIdx.x == 0) atomicOr(&wlo_any[rb], 1);
}

// one workgroup per (padded) row of the fixed-point H = hscale * (I + J / 65536) (J: the extension word, when on):
// x = H[r][u] 2^(12 - e_r - e_u), |x| < 2^14
__global__ __launch_bounds__(256) void wh_split_h_kernel(const int* __restrict__ Hint, const short* __restrict__ Jint,
                                                        const int* __restrict__ ext, const int* __restrict__ dexp,
                                                        const double* __restrict__ hscale, int n, int KT, char* __restrict__ Hp) {
    const int r = blockIdx.x, rb = r >> 7, rr = r & 127;
    const int rc = min(r, n - 1);
    const int* h = Hint + (int64_t)rc * n;
    const short* hj = Jint + (int64_t)rc * n;
    const bool use_j = *ext != 0;
    const double hs = *hscale;
    const int er = dexp[rc];
    const bool vec = (n & 7) == 0 && (reinterpret_cast<uintptr_t>(Hint) & 15) == 0 && (reinterpret_cast<uintptr_t>(Jint) & 15) == 0 &&
                     (reinterpret_cast<uintptr_t>(dexp) & 15) == 0;
    for (int ci = threadIdx.x; ci < KT * 4; ci += 256) {
        h8v hi, lo;
        int hv8[8], es8[8];
        short jv8[8];
        const bool full = vec && r < n && ci * 8 + 8 <= n;
        if (full) {  // 16-byte loads (see wh_split_w_kernel)
            *reinterpret_cast<int4*>(&hv8[0]) = reinterpret_cast<const int4*>(h)[ci * 2];
            *reinterpret_cast<int4*>(&hv8[4]) = reinterpret_cast<const int4*>(h)[ci * 2 + 1];
            *reinterpret_cast<int4*>(&es8[0]) = reinterpret_cast<const int4*>(dexp)[ci * 2];
            *reinterpret_cast<int4*>(&es8[4]) = reinterpret_cast<const int4*>(dexp)[ci * 2 + 1];
            if (use_j) *reinterpret_cast<int4*>(&jv8[0]) = reinterpret_cast<const int4*>(hj)[ci];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int u = ci * 8 + k;
            double x = 0.0;
            if (full) {
                if (u != r) {  // the diagonal is added exactly by the GEMM's epilogue
                    const double fx = (double)hv8[k] + (use_j ? (double)((int)jv8[k] - 128) * (1.0 / 65536.0) : 0.0);  // stored biased
                    x = ldexp(fx * hs, 12 - er - es8[k]);
                }
            } else if (r < n && u < n && u != r) {
                const double fx = (double)h[u] + (use_j ? (double)((int)hj[u] - 128) * (1.0 / 65536.0) : 0.0);
                x = ldexp(fx * hs, 12 - er - dexp[u]);
            }
            const _Float16 a = (_Float16)(float)x;
            hi[k] = a;
            lo[k] = (_Float16)(float)(x - (double)(float)a);
        }
        char* dst = Hp + ((int64_t)rb * KT + (ci >> 2)) * (2 * WIMG) + wh_chunk_offset(rr, ci & 3);
        *reinterpret_cast<h8v*>(dst) = hi;
        *reinterpret_cast<h8v*>(dst + WIMG) = lo;
    }
}

// WH[row][col] (fp64) = 2^(dexp[col] - 12 - rexp[row]) * sum_k (Whi + Wlo)[row][k] (Hhi + Hlo)[col][k]
// WLO: this instantiation serves the 128-row blocks of W whose low pieces are (not) all zero; both are launched and a
// workgroup of the other kind leaves at once (the flags are only known on the device).
template <bool WLO>
__global__ __launch_bounds__(256, 2) void wh_gemm_kernel(const char* __restrict__ Wp, const char* __restrict__ Hp,
                                                        const int* __restrict__ rexp, const int* __restrict__ wlo_any,
                                                        const int* __restrict__ dexp, const float* __restrict__ W,
                                                        const double* __restrict__ hdiag64, double* __restrict__ WH, int m,
                                                        int n, int KT, int tiles_m, int tiles_n) {
    __shared__ __align__(16) char lds[2][4 * WIMG];  // per buffer: Whi, Wlo, Hhi, Hlo images
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;

    // workgroups are dealt round-robin to the 8 XCDs: give every XCD a contiguous range of logical tiles, and walk
    // the tiles in groups of 8 tile rows, column by column, so that the workgroups resident together on an XCD form
    // an 8 x 8 block (8 W panels + 8 H panels in flight in its L2)
    const int total = tiles_m * tiles_n;
    int lt = blockIdx.x;
    if ((total & 7) == 0) lt = (lt & 7) * (total >> 3) + (lt >> 3);
    const int gsz = 8 * tiles_n;
    const int g = lt / gsz, gr = min(8, tiles_m - g * 8);
    const int tm = g * 8 + (lt % gsz) % gr, tn = (lt % gsz) / gr;

    const char* wsrc = Wp + (int64_t)tm * KT * (2 * WIMG);
    const char* hsrc = Hp + (int64_t)tn * KT * (2 * WIMG);
    if ((wlo_any[tm] != 0) != WLO) return;

    constexpr int NA = WLO ? 4 : 2;  // 4 KB slices of the W images to copy (hi only, or hi and lo)
    u4v sa[NA], sb[4];
    auto gload = [&](int kt) {
        const char* a = wsrc + (int64_t)kt * (2 * WIMG) + tid * 16;
        const char* b = hsrc + (int64_t)kt * (2 * WIMG) + tid * 16;
#pragma unroll
        for (int i = 0; i < NA; ++i) sa[i] = *reinterpret_cast<const u4v*>(a + i * 4096);
#pragma unroll
        for (int i = 0; i < 4; ++i) sb[i] = *reinterpret_cast<const u4v*>(b + i * 4096);
    };
    auto sstore = [&](int buf) {
        char* d = lds[buf] + tid * 16;
#pragma unroll
        for (int i = 0; i < NA; ++i) *reinterpret_cast<u4v*>(d + i * 4096) = sa[i];
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<u4v*>(d + 2 * WIMG + i * 4096) = sb[i];
    };

    const int wm = (wv >> 1) * 64, wn = (wv & 1) * 64;
    const int l31 = lane & 31, hf = lane >> 5;
    const int swz = (l31 >> 2) & 3;
    int aoff[2], boff[2], coff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        aoff[i] = (wm + 32 * i + l31) * 64;
        boff[i] = 2 * WIMG + (wn + 32 * i + l31) * 64;
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) coff[ks] = ((ks * 2 + hf) ^ swz) << 4;

    f16v acc[2][2];
    double accd[2][2][16];  // the fp32 accumulators are emptied into these every WFLUSH k tiles
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc[i][j][r] = 0.0f;
                accd[i][j][r] = 0.0;
            }
    auto flush = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    accd[i][j][r] += (double)acc[i][j][r];
                    acc[i][j][r] = 0.0f;
                }
    };

    gload(0);
    sstore(0);
    __syncthreads();
    for (int kt = 0; kt < KT; ++kt) {
        gload(min(kt + 1, KT - 1));  // the last iteration re-reads its own tile: no branch around the loads
        const char* base = lds[kt & 1];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            h8v ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                ah[i] = *reinterpret_cast<const h8v*>(base + aoff[i] + coff[ks]);
                bh[i] = *reinterpret_cast<const h8v*>(base + boff[i] + coff[ks]);
                bl[i] = *reinterpret_cast<const h8v*>(base + boff[i] + WIMG + coff[ks]);
            }
            if (WLO) {
#pragma unroll
                for (int i = 0; i < 2; ++i) al[i] = *reinterpret_cast<const h8v*>(base + aoff[i] + WIMG + coff[ks]);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                }
            if (WLO) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
            }
        }
        if ((kt & (WFLUSH - 1)) == WFLUSH - 1) flush();
        sstore((kt + 1) & 1);
        __syncthreads();
    }
    flush();

    // C layout of the 32x32 MFMA: column = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int row = tm * WT + wm + 32 * i + (reg & 3) + 8 * (reg >> 2) + 4 * hf;
            if (row >= m) continue;
            const int re = -12 - rexp[row];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int col = tn * WT + wn + 32 * j + l31;
                if (col < n)
                    WH[(int64_t)row * n + col] =
                        ldexp(accd[i][j][reg], re + dexp[col]) + (double)W[(int64_t)row * n + col] * hdiag64[col];
            }
        }
}

WhLayout wh_layout(int64_t m, int64_t n) {
    WhLayout lo;
    lo.KT = (n + WK - 1) / WK;
    lo.tiles_m = (m + WT - 1) / WT;
    lo.tiles_n = (n + WT - 1) / WT;
    lo.wp_bytes = (size_t)lo.tiles_m * lo.KT * 2 * WIMG;
    lo.hp_bytes = (size_t)lo.tiles_n * lo.KT * 2 * WIMG;
    lo.rexp_bytes = align_up((size_t)lo.tiles_m * WT * sizeof(int), 256);
    lo.wlo_bytes = align_up((size_t)lo.tiles_m * sizeof(int), 256);
    return lo;
}

int wh_gemm(const float* W, const int* Hint, const short* Jint, const int* ext, const int* dexp, const double* hdiag64,
            const double* hscale, int64_t m, int64_t n, const WhLayout& lo, char* wp, char* hp, int* rexp, int* wlo_any, double* WH,
            hipStream_t stream) {
    GANQ_HIP_CHECK(hipMemsetAsync(wlo_any, 0, lo.wlo_bytes, stream));
    hipLaunchKernelGGL(wh_split_w_kernel, dim3((unsigned)(lo.tiles_m * WT)), dim3(256), 0, stream, W, dexp, (int)m, (int)n,
                       (int)lo.KT, wp, rexp, wlo_any);
    hipLaunchKernelGGL(wh_split_h_kernel, dim3((unsigned)(lo.tiles_n * WT)), dim3(256), 0, stream, Hint, Jint, ext, dexp, hscale,
                       (int)n, (int)lo.KT, hp);
    hipLaunchKernelGGL(wh_gemm_kernel<false>, dim3((unsigned)(lo.tiles_m * lo.tiles_n)), dim3(256), 0, stream, wp, hp, rexp, wlo_any,
                       dexp, W, hdiag64, WH, (int)m, (int)n, (int)lo.KT, (int)lo.tiles_m, (int)lo.tiles_n);
    hipLaunchKernelGGL(wh_gemm_kernel<true>, dim3((unsigned)(lo.tiles_m * lo.tiles_n)), dim3(256), 0, stream, wp, hp, rexp, wlo_any,
                       dexp, W, hdiag64, WH, (int)m, (int)n, (int)lo.KT, (int)lo.tiles_m, (int)lo.tiles_n);
    GANQ_LAUNCH_CHECK();
    return 0;
}

}  // namespace ganq

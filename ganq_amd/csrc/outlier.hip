// Outlier split in front of GANQ (paper section 3.3 and Appendix A, Algorithm 2; reference paper.md:195-197,882-899 --
// the reference repository does not implement it):  W = W_sparse + W_dense, row by row the entries at or beyond the
// p = 1 - r/2 and 1 - p quantiles go to a sparse fp16 matrix, GANQ quantizes the rest, and the layer computes
//     y = LUT(x; Q, T) + x @ W_sparse^T.
//
//   outlier_cutoff_kernel  one workgroup per row: the row as order-preserving integer keys in LDS, the two order
//                          statistics by 4-pass radix select (256-bin LDS histograms), then the count of entries with
//                          w >= c_upper or w <= c_lower (ties count, as in Algorithm 2);
//   outlier_scan_kernel    exclusive scan of the counts -> CSR row pointers;
//   outlier_extract_kernel one wave per row: ballot compaction in ascending column order, zeroes W at those entries;
//   outlier_matmul_kernel  out[b][r] = sum_e vals[e] * x[b][cols[e]] in fp32 (16 lanes per output row; the products of
//                          16-bit floats are exact in fp32), handed to the LUT kernel as an fp32 addend so that the sum is
//                          rounded to the activation dtype once.
#include "common.h"

namespace ganq {

__device__ __forceinline__ uint32_t ordered_key(float x) {
    const uint32_t u = __builtin_bit_cast(uint32_t, x);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_value(uint32_t k) {
    return __builtin_bit_cast(float, (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// k-th smallest key (0-based) of keys[0..n): most significant byte first
__device__ uint32_t radix_select(const uint32_t* keys, int n, int k, uint32_t* hist, uint32_t* pick) {
    uint32_t prefix = 0, mask = 0;
    for (int shift = 24; shift >= 0; shift -= 8) {
        for (int i = threadIdx.x; i < 256; i += blockDim.x) hist[i] = 0;
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            const uint32_t key = keys[i];
            if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t below = 0;
            int b = 0;
            for (; b < 255; ++b) {
                if (below + hist[b] > (uint32_t)k) break;
                below += hist[b];
            }
            pick[0] = (uint32_t)b;
            pick[1] = below;
        }
        __syncthreads();
        prefix |= pick[0] << shift;
        mask |= 255u << shift;
        k -= (int)pick[1];
        __syncthreads();
    }
    return prefix;
}

__global__ __launch_bounds__(256) void outlier_cutoff_kernel(const float* __restrict__ W, int m, int n, int lower, int upper,
                                                            float* __restrict__ cut, int* __restrict__ counts) {
    extern __shared__ __align__(16) uint32_t okeys[];  // n keys, then 256 bins, then 2 words
    uint32_t* hist = okeys + n;
    uint32_t* pick = hist + 256;
    __shared__ int cnt_sh;
    const int row = blockIdx.x;
    const float* w = W + (int64_t)row * n;
    for (int i = threadIdx.x; i < n; i += 256) okeys[i] = ordered_key(w[i]);
    if (threadIdx.x == 0) cnt_sh = 0;
    __syncthreads();
    const float c_lower = key_value(radix_select(okeys, n, lower, hist, pick));
    const float c_upper = key_value(radix_select(okeys, n, upper, hist, pick));
    int c = 0;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float x = w[i];
        c += (x >= c_upper || x <= c_lower) ? 1 : 0;
    }
    for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off);
    if ((threadIdx.x & 63) == 0) atomicAdd(&cnt_sh, c);
    __syncthreads();
    if (threadIdx.x == 0) {
        cut[2 * row] = c_lower;
        cut[2 * row + 1] = c_upper;
        counts[row] = cnt_sh;
    }
}

// rowptr[0..m] = exclusive scan of counts[0..m); one workgroup, 1024 rows per pass
__global__ __launch_bounds__(1024) void outlier_scan_kernel(const int* __restrict__ counts, int m, int* __restrict__ rowptr) {
    __shared__ int wsum[16];
    __shared__ int carry_sh;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) carry_sh = 0;
    __syncthreads();
    for (int base = 0; base < m; base += 1024) {
        const int i = base + tid;
        const int v = i < m ? counts[i] : 0;
        int incl = v;
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(incl, off);
            if (lane >= off) incl += t;
        }
        if (lane == 63) wsum[wv] = incl;
        __syncthreads();
        int before = carry_sh;
        for (int k = 0; k < wv; ++k) before += wsum[k];
        if (i < m) rowptr[i] = before + incl - v;
        __syncthreads();
        if (tid == 1023) carry_sh = before + incl;
        __syncthreads();
    }
    if (tid == 0) rowptr[m] = carry_sh;
}

__global__ __launch_bounds__(256) void outlier_extract_kernel(float* __restrict__ W, int m, int n, const float* __restrict__ cut,
                                                             const int* __restrict__ rowptr, int* __restrict__ cols,
                                                             float* __restrict__ vals) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= m) return;
    float* w = W + (int64_t)row * n;
    const float c_lower = cut[2 * row], c_upper = cut[2 * row + 1];
    int pos = rowptr[row];
    for (int c0 = 0; c0 < n; c0 += 64) {
        const int c = c0 + lane;
        const float x = c < n ? w[c] : 0.0f;
        const bool o = c < n && (x >= c_upper || x <= c_lower);
        const unsigned long long mk = __ballot(o);
        if (o) {
            const int at = pos + __popcll(mk & ((1ull << lane) - 1ull));
            cols[at] = c;
            vals[at] = x;
            w[c] = 0.0f;
        }
        pos += __popcll(mk);
    }
}

template <bool BF16>
__device__ __forceinline__ float h2f(uint16_t v) {
    return BF16 ? __builtin_bit_cast(float, (uint32_t)v << 16) : (float)__builtin_bit_cast(_Float16, v);
}

template <int CTRL>
__device__ __forceinline__ float dpp_add(float x) {
    return x + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
}

// out[b][r] = sum_e vals[e] * x[b][cols[e]], e in [rowptr[r], rowptr[r+1]); 16 lanes per output row r
template <bool BF16>
__global__ __launch_bounds__(256) void outlier_matmul_kernel(const uint16_t* __restrict__ x, int M, int m, int n,
                                                            const int* __restrict__ rowptr, const int* __restrict__ cols,
                                                            const uint16_t* __restrict__ vals, float* __restrict__ out) {
    const int s = threadIdx.x & 15;
    const int r = blockIdx.x * 16 + (threadIdx.x >> 4);
    const int rc = min(r, m - 1);
    const int e0 = rowptr[rc], e1 = r < m ? rowptr[rc + 1] : e0;
    // the first two entries of this lane stay in registers (rows with up to 32 outliers never re-read the lists)
    const int ea = e0 + s, eb = e0 + 16 + s;
    const int ca = ea < e1 ? cols[ea] : 0, cb = eb < e1 ? cols[eb] : 0;
    const float va = ea < e1 ? h2f<BF16>(vals[ea]) : 0.0f, vb = eb < e1 ? h2f<BF16>(vals[eb]) : 0.0f;
    for (int b = 0; b < M; ++b) {
        const uint16_t* xb = x + (int64_t)b * n;
        float acc = va * h2f<BF16>(xb[ca]);
        acc = fmaf(vb, h2f<BF16>(xb[cb]), acc);
        for (int e = e0 + 32 + s; e < e1; e += 16) acc = fmaf(h2f<BF16>(vals[e]), h2f<BF16>(xb[cols[e]]), acc);
        acc = dpp_add<0xB1>(acc);   // quad_perm [1,0,3,2]
        acc = dpp_add<0x4E>(acc);   // quad_perm [2,3,0,1]
        acc = dpp_add<0x141>(acc);  // row_half_mirror
        acc = dpp_add<0x140>(acc);  // row_mirror
        if (s == 0 && r < m) out[(int64_t)b * m + r] = acc;
    }
}

}  // namespace ganq

using namespace ganq;

extern "C" int ganq_outlier_cutoffs(const float* W, int64_t m, int64_t n, double ratio, float* cut, int32_t* counts,
                                    int32_t* rowptr, void* stream_) {
    if (m < 0 || n < 0) return fail(-1, "ganq_outlier_cutoffs: negative shape");
    if (!(ratio > 0.0 && ratio < 1.0)) return fail(-2, "ganq_outlier_cutoffs: ratio %g must be in (0, 1)", ratio);
    if (n > 16384) return fail(-2, "ganq_outlier_cutoffs: in_features=%lld > 16384", (long long)n);
    if (m > INT32_MAX / 2) return fail(-1, "ganq_outlier_cutoffs: shape too large");
    if (!rowptr) return fail(-3, "ganq_outlier_cutoffs: null pointer");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (m == 0 || n == 0) {
        GANQ_HIP_CHECK(hipMemsetAsync(rowptr, 0, (size_t)(m + 1) * sizeof(int32_t), stream));
        if (counts && m) GANQ_HIP_CHECK(hipMemsetAsync(counts, 0, (size_t)m * sizeof(int32_t), stream));
        return 0;
    }
    if (!W || !cut || !counts) return fail(-3, "ganq_outlier_cutoffs: null pointer");
    // Algorithm 2: p = 1 - r/2, upper = floor(n p), lower = ceil(n (1 - p)), 0-based positions in the ascending row
    const double p = 1.0 - 0.5 * ratio;
    int upper = (int)__builtin_floor((double)n * p);
    int lower = (int)__builtin_ceil((double)n * (1.0 - p));
    upper = std::min(std::max(upper, 0), (int)n - 1);
    lower = std::min(std::max(lower, 0), (int)n - 1);
    const size_t smem = ((size_t)n + 256 + 2) * sizeof(uint32_t);
    {
        int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(outlier_cutoff_kernel), smem);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(outlier_cutoff_kernel, dim3((unsigned)m), dim3(256), smem, stream, W, (int)m, (int)n, lower, upper, cut,
                       counts);
    hipLaunchKernelGGL(outlier_scan_kernel, dim3(1), dim3(1024), 0, stream, counts, (int)m, rowptr);
    GANQ_LAUNCH_CHECK();
    return 0;
}

extern "C" int ganq_outlier_extract(float* W, int64_t m, int64_t n, const float* cut, const int32_t* rowptr, int32_t* cols,
                                    float* vals, void* stream_) {
    if (m < 0 || n < 0) return fail(-1, "ganq_outlier_extract: negative shape");
    if (m == 0 || n == 0) return 0;
    if (m > INT32_MAX / 2 || n > INT32_MAX / 2) return fail(-1, "ganq_outlier_extract: shape too large");
    if (!W || !cut || !rowptr || !cols || !vals) return fail(-3, "ganq_outlier_extract: null pointer");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    hipLaunchKernelGGL(outlier_extract_kernel, dim3((unsigned)((m + 3) / 4)), dim3(256), 0, stream, W, (int)m, (int)n, cut, rowptr,
                       cols, vals);
    GANQ_LAUNCH_CHECK();
    return 0;
}

extern "C" int ganq_outlier_matmul(const void* x, int dtype, int64_t M, int64_t m, int64_t n, const int32_t* rowptr,
                                   const int32_t* cols, const void* vals, float* out, void* stream_) {
    if (M < 0 || m < 0 || n < 0) return fail(-1, "ganq_outlier_matmul: negative shape");
    if (dtype != 0 && dtype != 1) return fail(-2, "ganq_outlier_matmul: dtype %d (0 = fp16, 1 = bf16)", dtype);
    if (M == 0 || m == 0) return 0;
    if (m > INT32_MAX / 2 || n > INT32_MAX / 2 || M > INT32_MAX / 2) return fail(-1, "ganq_outlier_matmul: shape too large");
    if (!x || !rowptr || !out) return fail(-3, "ganq_outlier_matmul: null pointer");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const dim3 grid((unsigned)((m + 15) / 16));
    const uint16_t* xp = static_cast<const uint16_t*>(x);
    const uint16_t* vp = static_cast<const uint16_t*>(vals);
    if (dtype == 1)
        hipLaunchKernelGGL(outlier_matmul_kernel<true>, grid, dim3(256), 0, stream, xp, (int)M, (int)m, (int)n, rowptr, cols, vp, out);
    else
        hipLaunchKernelGGL(outlier_matmul_kernel<false>, grid, dim3(256), 0, stream, xp, (int)M, (int)m, (int)n, rowptr, cols, vp, out);
    GANQ_LAUNCH_CHECK();
    return 0;
}

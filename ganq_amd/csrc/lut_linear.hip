// LUT-dequant linear forward and index packing.
//   y[M,m] = x[M,n] @ dequant(qweight, lut)^T + bias        (replaces FakeQuantLinear.forward, fake.py:88-89,
//   whose weight is T.gather(1,Q).half(); the reference keeps neither Q nor T, ganq.py:633-646)
//
// Storage: `bits`-wide indices packed as one little-endian bit stream per output feature along in_features,
// 32-bit words laid out qweight[n*bits/32][m] -- exactly the GPTQ int32 packing (qlinear/__init__.py:508-538,
// including its 3-bit 32-in-3-words scheme, which is the same bit stream); lut [m][V] in the activation dtype.
//
// ganq_lut_linear_fwd (M <= 16 rows, decode): one lane per output feature, the wave streams its features'
//   words (coalesced 256 B per word row), looks the codebook up in an LDS copy of lut and accumulates in fp32;
//   the in_features range is split over workgroups and waves (deterministic two-stage reduction).
// ganq_lut_dequant: materialises W_q [m,n] in the activation dtype for large-M products (prefill), which the
//   host side hands to a library GEMM.
#include "common.h"

namespace ganq {

__device__ __forceinline__ float load_act(const void* p, int64_t i, int dtype) {
    const uint16_t h = static_cast<const uint16_t*>(p)[i];
    if (dtype == 1) return __builtin_bit_cast(float, (uint32_t)h << 16);
    return (float)__builtin_bit_cast(_Float16, h);
}
__device__ __forceinline__ uint16_t store_act(float v, int dtype) {
    if (dtype == 1) return __builtin_bit_cast(uint16_t, (__bf16)v);
    return __builtin_bit_cast(uint16_t, (_Float16)v);
}

// extract element j (0..31) of a 32-element group from its BITS words
template <int BITS>
__device__ __forceinline__ uint32_t extract(const uint32_t (&w)[BITS], int j) {
    const int bitpos = BITS * j;
    const int wi = bitpos >> 5, sh = bitpos & 31;
    uint32_t v = w[wi] >> sh;
    if (sh + BITS > 32) v |= w[wi + 1 < BITS ? wi + 1 : wi] << (32 - sh);
    return v & ((1u << BITS) - 1u);
}

constexpr int LW = 4;  // waves per workgroup

template <int BITS, int MT>
__global__ __launch_bounds__(LW * 64) void lut_gemv_kernel(const void* __restrict__ x, const uint32_t* __restrict__ qw,
                                                          const void* __restrict__ lut, int dtype, int M, int m, int n,
                                                          int groups_per_wave, float* __restrict__ partial) {
    constexpr int V = 1 << BITS;
    __shared__ float tbl[64][V + 1];
    __shared__ float xs[LW][32][MT];
    __shared__ float red[LW][MT][64];

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int o0 = blockIdx.x * 64;
    const int ks = blockIdx.y;
    const int o = o0 + lane;
    const int oc = min(o, m - 1);
    for (int i = tid; i < 64 * V; i += LW * 64) {
        const int ol = i / V, e = i % V;
        tbl[ol][e] = load_act(lut, (int64_t)min(o0 + ol, m - 1) * V + e, dtype);
    }
    __syncthreads();

    const int ngroups = n >> 5;
    const int g_begin = (ks * LW + wv) * groups_per_wave;
    const int g_end = min(ngroups, g_begin + groups_per_wave);
    for (int r0 = 0; r0 < M; r0 += MT) {
        float acc[MT];
#pragma unroll
        for (int r = 0; r < MT; ++r) acc[r] = 0.f;
        for (int g = g_begin; g < g_end; ++g) {
            // activations of this group: 32 x MT values, wave-private LDS slab
            if (lane < 32) {
#pragma unroll
                for (int r = 0; r < MT; ++r)
                    xs[wv][lane][r] = (r0 + r < M) ? load_act(x, (int64_t)(r0 + r) * n + 32 * g + lane, dtype) : 0.f;
            }
            uint32_t w[BITS];
#pragma unroll
            for (int b = 0; b < BITS; ++b) w[b] = qw[(int64_t)(g * BITS + b) * m + oc];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                const float wq = tbl[lane][extract<BITS>(w, j)];
#pragma unroll
                for (int r = 0; r < MT; ++r) acc[r] = fmaf(xs[wv][j][r], wq, acc[r]);
            }
            __builtin_amdgcn_wave_barrier();
        }
#pragma unroll
        for (int r = 0; r < MT; ++r) red[wv][r][lane] = acc[r];
        __syncthreads();
        if (wv == 0) {
#pragma unroll
            for (int r = 0; r < MT; ++r) {
                float s = red[0][r][lane];
#pragma unroll
                for (int w2 = 1; w2 < LW; ++w2) s += red[w2][r][lane];
                if (o < m && r0 + r < M) partial[((int64_t)ks * M + r0 + r) * m + o] = s;
            }
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void lut_finish_kernel(const float* __restrict__ partial, const void* __restrict__ bias,
                                                         int dtype, int KS, int M, int m, void* __restrict__ y) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)M * m) return;
    const int o = (int)(i % m);
    float s = 0.f;
    for (int k = 0; k < KS; ++k) s += partial[(int64_t)k * M * m + i];
    if (bias) s += load_act(bias, o, dtype);
    static_cast<uint16_t*>(y)[i] = store_act(s, dtype);
}

template <int BITS>
__global__ __launch_bounds__(256) void lut_dequant_kernel(const uint32_t* __restrict__ qw, const void* __restrict__ lut,
                                                          int dtype, int m, int n, uint16_t* __restrict__ Wq) {
    // thread = (group g of 32 columns, output o); o fastest -> coalesced word reads; writes 64 B runs per thread
    constexpr int V = 1 << BITS;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int ngroups = n >> 5;
    if (i >= (int64_t)ngroups * m) return;
    const int o = (int)(i % m), g = (int)(i / m);
    uint32_t w[BITS];
#pragma unroll
    for (int b = 0; b < BITS; ++b) w[b] = qw[(int64_t)(g * BITS + b) * m + o];
    const uint16_t* lr = static_cast<const uint16_t*>(lut) + (int64_t)o * V;
    uint16_t* out = Wq + (int64_t)o * n + 32 * g;
    (void)dtype;
#pragma unroll
    for (int j = 0; j < 32; ++j) out[j] = lr[extract<BITS>(w, j)];
}

__global__ __launch_bounds__(256) void pack_kernel(const uint8_t* __restrict__ Q, int m, int n, int bits,
                                                   uint32_t* __restrict__ qw) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int nwords = n * bits / 32;
    if (i >= (int64_t)nwords * m) return;
    const int o = (int)(i % m), w = (int)(i / m);
    const uint8_t* q = Q + (int64_t)o * n;
    uint32_t word = 0;
    const int lo_bit = 32 * w, hi_bit = lo_bit + 32;
    for (int e = lo_bit / bits; e < n && e * bits < hi_bit; ++e) {
        const int pos = e * bits - lo_bit;
        const uint32_t v = q[e] & ((1u << bits) - 1u);
        word |= (pos >= 0) ? (v << pos) : (v >> (-pos));
    }
    qw[(int64_t)w * m + o] = word;
}

__global__ __launch_bounds__(256) void unpack_kernel(const uint32_t* __restrict__ qw, int m, int n, int bits,
                                                     uint8_t* __restrict__ Q) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)m * n) return;
    const int o = (int)(i / n), e = (int)(i % n);
    const int bitpos = e * bits, w = bitpos >> 5, sh = bitpos & 31;
    uint32_t v = qw[(int64_t)w * m + o] >> sh;
    if (sh + bits > 32) v |= qw[(int64_t)(w + 1) * m + o] << (32 - sh);
    Q[i] = (uint8_t)(v & ((1u << bits) - 1u));
}

static void lut_split(int64_t m, int64_t n, int* KS, int* groups_per_wave) {
    const int ngroups = (int)(n >> 5);
    const int ob = (int)((m + 63) / 64);
    int ks = std::max(1, std::min((ngroups + LW - 1) / LW, (1024 + ob - 1) / ob));
    int gpw = (ngroups + ks * LW - 1) / (ks * LW);
    ks = (ngroups + gpw * LW - 1) / (gpw * LW);
    *KS = ks;
    *groups_per_wave = gpw;
}

}  // namespace ganq

using namespace ganq;

extern "C" size_t ganq_lut_linear_workspace_bytes(int64_t M, int64_t m, int64_t n, int bits) {
    (void)bits;
    if (M <= 0 || m <= 0 || n <= 0) return 0;
    int KS, gpw;
    lut_split(m, n, &KS, &gpw);
    return align_up((size_t)KS * (size_t)M * (size_t)m * sizeof(float), 256);
}

template <int BITS>
static int launch_gemv(const void* x, const uint32_t* qw, const void* lut, int dtype, int M, int m, int n, int KS,
                       int gpw, float* partial, hipStream_t stream) {
    const dim3 grid((unsigned)((m + 63) / 64), (unsigned)KS);
    if (M == 1)
        hipLaunchKernelGGL((lut_gemv_kernel<BITS, 1>), grid, dim3(LW * 64), 0, stream, x, qw, lut, dtype, M, m, n, gpw, partial);
    else if (M == 2)
        hipLaunchKernelGGL((lut_gemv_kernel<BITS, 2>), grid, dim3(LW * 64), 0, stream, x, qw, lut, dtype, M, m, n, gpw, partial);
    else if (M <= 4)
        hipLaunchKernelGGL((lut_gemv_kernel<BITS, 4>), grid, dim3(LW * 64), 0, stream, x, qw, lut, dtype, M, m, n, gpw, partial);
    else
        hipLaunchKernelGGL((lut_gemv_kernel<BITS, 8>), grid, dim3(LW * 64), 0, stream, x, qw, lut, dtype, M, m, n, gpw, partial);
    GANQ_LAUNCH_CHECK();
    return 0;
}

static int check_lut_args(const char* who, int dtype, int64_t m, int64_t n, int bits) {
    if (dtype != 0 && dtype != 1) return fail(-2, "%s: dtype %d (0 = fp16, 1 = bf16)", who, dtype);
    if (bits != 2 && bits != 3 && bits != 4) return fail(-2, "%s: bits=%d not supported (2, 3, 4 are)", who, bits);
    if (n % 32 != 0) return fail(-2, "%s: in_features=%lld must be a multiple of 32", who, (long long)n);
    if (m > INT32_MAX / 2 || n > INT32_MAX / 2) return fail(-1, "%s: shape too large", who);
    return 0;
}

extern "C" int ganq_lut_linear_fwd(const void* x, const int32_t* qweight, const void* lut, const void* bias, int dtype,
                                   int64_t M, int64_t m, int64_t n, int bits, void* y, void* workspace,
                                   size_t workspace_bytes, void* stream_) {
    if (M < 0 || m < 0 || n < 0) return fail(-1, "ganq_lut_linear_fwd: negative shape");
    if (M == 0 || m == 0) return 0;
    int rc = check_lut_args("ganq_lut_linear_fwd", dtype, m, n, bits);
    if (rc) return rc;
    if (M > 16) return fail(-2, "ganq_lut_linear_fwd: M=%lld > 16; use ganq_lut_dequant + a GEMM for large batches", (long long)M);
    if (!x || !qweight || !lut || !y) return fail(-3, "ganq_lut_linear_fwd: null pointer");
    const size_t need = ganq_lut_linear_workspace_bytes(M, m, n, bits);
    if (!workspace || workspace_bytes < need)
        return fail(-4, "ganq_lut_linear_fwd: workspace %zu B < required %zu B", workspace_bytes, need);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    int KS, gpw;
    lut_split(m, n, &KS, &gpw);
    float* partial = static_cast<float*>(workspace);
    const uint32_t* qw = reinterpret_cast<const uint32_t*>(qweight);
    {
        ProfScope prof(KID_LUT_GEMV, stream);
        if (bits == 2) rc = launch_gemv<2>(x, qw, lut, dtype, (int)M, (int)m, (int)n, KS, gpw, partial, stream);
        else if (bits == 3) rc = launch_gemv<3>(x, qw, lut, dtype, (int)M, (int)m, (int)n, KS, gpw, partial, stream);
        else rc = launch_gemv<4>(x, qw, lut, dtype, (int)M, (int)m, (int)n, KS, gpw, partial, stream);
        if (rc) return rc;
        hipLaunchKernelGGL(lut_finish_kernel, dim3((unsigned)((M * m + 255) / 256)), dim3(256), 0, stream, partial, bias, dtype,
                           KS, (int)M, (int)m, y);
    }
    GANQ_LAUNCH_CHECK();
    return 0;
}

extern "C" int ganq_lut_dequant(const int32_t* qweight, const void* lut, int dtype, int64_t m, int64_t n, int bits,
                                void* Wq_out, void* stream_) {
    if (m < 0 || n < 0) return fail(-1, "ganq_lut_dequant: negative shape");
    if (m == 0 || n == 0) return 0;
    int rc = check_lut_args("ganq_lut_dequant", dtype, m, n, bits);
    if (rc) return rc;
    if (!qweight || !lut || !Wq_out) return fail(-3, "ganq_lut_dequant: null pointer");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int64_t total = (n >> 5) * m;
    const dim3 grid((unsigned)((total + 255) / 256));
    const uint32_t* qw = reinterpret_cast<const uint32_t*>(qweight);
    uint16_t* out = static_cast<uint16_t*>(Wq_out);
    ProfScope prof(KID_LUT_GEMM, stream);
    if (bits == 2) hipLaunchKernelGGL(lut_dequant_kernel<2>, grid, dim3(256), 0, stream, qw, lut, dtype, (int)m, (int)n, out);
    else if (bits == 3) hipLaunchKernelGGL(lut_dequant_kernel<3>, grid, dim3(256), 0, stream, qw, lut, dtype, (int)m, (int)n, out);
    else hipLaunchKernelGGL(lut_dequant_kernel<4>, grid, dim3(256), 0, stream, qw, lut, dtype, (int)m, (int)n, out);
    GANQ_LAUNCH_CHECK();
    return 0;
}

extern "C" int ganq_pack_indices(const uint8_t* Q, int64_t m, int64_t n, int bits, int32_t* qweight, void* stream_) {
    if (m < 0 || n < 0) return fail(-1, "ganq_pack_indices: negative shape");
    if (m == 0 || n == 0) return 0;
    int rc = check_lut_args("ganq_pack_indices", 0, m, n, bits);
    if (rc) return rc;
    if (!Q || !qweight) return fail(-3, "ganq_pack_indices: null pointer");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int64_t total = (n * bits / 32) * m;
    ProfScope prof(KID_PACK, stream);
    hipLaunchKernelGGL(pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, Q, (int)m, (int)n, bits,
                       reinterpret_cast<uint32_t*>(qweight));
    GANQ_LAUNCH_CHECK();
    return 0;
}

extern "C" int ganq_unpack_indices(const int32_t* qweight, int64_t m, int64_t n, int bits, uint8_t* Q, void* stream_) {
    if (m < 0 || n < 0) return fail(-1, "ganq_unpack_indices: negative shape");
    if (m == 0 || n == 0) return 0;
    int rc = check_lut_args("ganq_unpack_indices", 0, m, n, bits);
    if (rc) return rc;
    if (!Q || !qweight) return fail(-3, "ganq_unpack_indices: null pointer");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int64_t total = m * n;
    ProfScope prof(KID_PACK, stream);
    hipLaunchKernelGGL(unpack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream,
                       reinterpret_cast<const uint32_t*>(qweight), (int)m, (int)n, bits, Q);
    GANQ_LAUNCH_CHECK();
    return 0;
}

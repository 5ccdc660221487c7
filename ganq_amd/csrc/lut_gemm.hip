// LUT-dequant GEMM for M > 64 rows of x (prefill, perplexity evaluation at seqlen 2048):
//   y[M,m] = x[M,n] @ dequant(qweight, lut)^T (+ addend) (+ bias)        fake.py:88-89 with the weight never materialised
//
// One workgroup = a BM x BN tile of y (BM = 128 or 256 rows of x, BN = 128 output features), 4 waves as 2 x 2, every wave a
// (BM / 2) x 64 sub-tile = RM x 2 v_mfma_f32_32x32x16_{f16,bf16} tiles, fp32 accumulation over all of in_features (no split-K: one rounding).
//   A operand (activations): a BM x 128 slab of x per stage goes through LDS (row pitch 256 + 16 B: the 16-byte fragment
//     reads of 16 consecutive rows fall on 64 distinct banks); the global loads of a stage -- slab and weight words -- are
//     issued a whole stage (16 or 32 matrix instructions per wave) before they are needed.
//   B operand (weights): decoded straight into the matrix-core operand, never written anywhere.  In the 32x32x16 layout lane
//     l holds output feature l % 32 and the 8 consecutive in_features 8 * (l / 32) .. of a 16-column step -- exactly 8 * BITS
//     consecutive bits of that feature's GPTQ bit stream: one word load (two for 3-bit), 8 codebook lookups in LDS
//     (tbl[tile][entry][lane]: one dword slot per lane and entry, conflict-free; v_perm builds the addresses for 4-bit),
//     4 packs.  The packed weight is read once per BM rows: 1/4 of the fp16 bytes at 4 bits.
// The workgroups of one launch are dealt to the 8 XCDs round-robin by the hardware; the tile index is remapped so that the
// workgroups behind one L2 walk the SAME row block of x over consecutive feature blocks (x is the large operand).
#include "common.h"

#include <algorithm>
#include <type_traits>

namespace ganq {

typedef _Float16 g_f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 g_bf16x8 __attribute__((ext_vector_type(8)));
typedef float g_f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t g_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t g_lds_pair(uint32_t addr_lo, uint32_t addr_hi) {
    typedef const uint32_t __attribute__((address_space(3))) * lds_u32;
    return *reinterpret_cast<lds_u32>(addr_lo) | (*reinterpret_cast<lds_u32>(addr_hi) << 16);
}

constexpr int GBN = 128;      // output features per workgroup
constexpr int GBM = 256;      // rows of x per workgroup: 4 waves x 64 rows
constexpr int GST = 2;        // groups of 32 in_features per stage
constexpr int GPITCH = 64 * GST + 16;  // bytes per LDS row (activations and decoded weights alike): + 16 B so that the 16-byte
                                       // fragment reads of 16 consecutive rows hit 64 distinct banks

// One workgroup = a 256 x 128 tile of y.  Wave w owns rows 64 w .. 64 w + 63 and ALL 128 features: 2 x 4 tiles of
// v_mfma_f32_32x32x16, every activation fragment feeds four matrix instructions, every weight fragment two.
// The weights of a stage are decoded ONCE per workgroup into an fp16 / bf16 tile in LDS (thread = one output feature and two
// of the four 8-column octets of every 32-column group: one word load, 8 conflict-free codebook lookups, one 16-byte LDS
// store per octet) -- a quarter of the lookups a per-wave decode into registers costs, which had bound the first version of this
// kernel (371 vector + 144 LDS instructions per 32 matrix instructions).
// SPLIT: blockIdx.y = ks takes the stages [ks * st_per, (ks + 1) * st_per) of in_features and writes its fp32 partial tile;
// lut_gemm_reduce_kernel, the next launch on the stream, sums the splits in ks order (deterministic), adds addend / bias and
// rounds once.  Serves 64 < M <= ~1024, where whole-K tiles would leave CUs idle.
template <int BITS, bool BF16, bool SPLIT, bool RAGGED>
__global__ __launch_bounds__(256, 2) void lut_gemm_kernel(const uint16_t* __restrict__ x, const uint32_t* __restrict__ qw,
                                                        const uint16_t* __restrict__ lut, const uint16_t* __restrict__ bias,
                                                        const float* __restrict__ addend, int M, int m, int n, int tiles_m,
                                                        int tiles_n, uint16_t* __restrict__ y, int st_per,
                                                        float* __restrict__ partial) {
    constexpr int V = 1 << BITS;
    constexpr bool STRADDLE = (8 * BITS) % 16 != 0;  // 3-bit: an octet's 24 bits can span two words
    constexpr int TPR = 4 * GST;                     // threads per slab row (16 B each)
    constexpr int RPP = 256 / TPR;                   // rows per pass of the 256 threads
    constexpr int NLD = GBM / RPP;                   // 16-byte activation loads per thread and stage
    constexpr int NIT = 2 * GST;                     // (group, octet) items a thread decodes per stage
    __shared__ __attribute__((aligned(4096))) uint32_t tbl[2][V][64];  // [feature half][entry][lane]: waves w and w + 2 share
    __shared__ __attribute__((aligned(16))) unsigned char As[GBM * GPITCH];
    __shared__ __attribute__((aligned(16))) unsigned char Bs[GBN * GPITCH];

    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hf = lane >> 5;
    // XCD-aware tile order: workgroup b runs on XCD b % 8; give each XCD whole row blocks of x
    int bid = blockIdx.x;
    {
        const int total = tiles_m * tiles_n;
        const int per = total >> 3;
        if ((total & 7) == 0) bid = (bid & 7) * per + (bid >> 3);  // (otherwise the plain order: the remap would leave holes)
    }
    const int tm = bid / tiles_n, tn = bid % tiles_n;
    const int r0 = tm * GBM, o0 = tn * GBN;
    const int nkb = n >> 5, nst_all = (nkb + GST - 1) / GST;
    const int st_begin = SPLIT ? (int)blockIdx.y * st_per : 0;
    const int nst = SPLIT ? min(nst_all, st_begin + st_per) : nst_all;  // this workgroup's stages: [st_begin, nst)

    // ---- decode role: this thread's feature and octets
    const int dfeat = tid & 127, dq0 = __builtin_amdgcn_readfirstlane(tid >> 7);  // octets dq0 and dq0 + 2 of every group (wave-uniform)
    const int dcol = min(o0 + dfeat, m - 1);
    if (wv < 2) {  // the codebook of feature o0 + 64 * wv + lane, one dword slot per entry and lane
        const uint32_t* lp = reinterpret_cast<const uint32_t*>(lut + (int64_t)dcol * V);
        uint32_t h[V / 2];
#pragma unroll
        for (int e = 0; e < V / 2; ++e) h[e] = lp[e];
#pragma unroll
        for (int e = 0; e < V / 2; ++e) {
            tbl[wv][2 * e][lane] = h[e] & 0xffffu;
            tbl[wv][2 * e + 1][lane] = h[e] >> 16;
        }
    }
    int wi[2], sh[2], wi2[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int off = 8 * BITS * (dq0 + 2 * j);  // bit offset of the octet inside the 32-column group
        wi[j] = off >> 5;
        sh[j] = off & 31;
        wi2[j] = STRADDLE ? min(wi[j] + 1, BITS - 1) : wi[j];
    }

    // ---- global -> register staging of one stage
    const int achunk = tid % TPR, arow0 = tid / TPR;  // rows arow0 + RPP * i
    // buffer loads: this thread's constant byte offsets in VGPRs, the stage's offset in an SGPR (no per-load 64-bit address
    // arithmetic; the host takes the pointer-free path only while both operands stay below 2 GB); activation rows past M lie
    // past the resource's end and read as zeros
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(x), 0, (int)((int64_t)M * n * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(qw), 0, (int)((int64_t)(n >> 5) * BITS * m * 4), 0x00020000);
    int xoff[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) xoff[i] = ((r0 + arow0 + RPP * i) * n + 8 * achunk) * 2;
    const int woff = dcol * 4;
    g_u32x4 xa[NLD];
    uint32_t wl[NIT], wh[NIT];  // item = (group g, octet dq0 + 2 j): index 2 g + j
    // nothing may touch the loaded registers here: a select right behind the loads makes the compiler wait for them on the
    // spot (measured: a third of the kernel's time); the columns of a ragged last stage are zeroed where the slab is stored
    auto gload = [&](int st) {
        const int soff = 64 * GST * st;  // bytes along a row
#pragma unroll
        for (int i = 0; i < NLD; ++i)
            xa[i] = __builtin_bit_cast(g_u32x4, RAGGED ? __builtin_amdgcn_raw_buffer_load_b128(rsx, xoff[i] + soff, 0, 0)
                                                      : __builtin_amdgcn_raw_buffer_load_b128(rsx, xoff[i], soff, 0));
#pragma unroll
        for (int g = 0; g < GST; ++g) {
            const int kb = RAGGED ? min(GST * st + g, nkb - 1) : GST * st + g;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                wl[2 * g + j] = __builtin_amdgcn_raw_buffer_load_b32(rsw, woff, (kb * BITS + wi[j]) * m * 4, 0);
                if (STRADDLE) wh[2 * g + j] = __builtin_amdgcn_raw_buffer_load_b32(rsw, woff, (kb * BITS + wi2[j]) * m * 4, 0);
            }
        }
    };
    const uint32_t tb = (uint32_t)(uintptr_t)(&tbl[0][0][0]) + (uint32_t)(wv & 1) * (V * 256u);
    const uint32_t lane4 = 4u * lane;
    auto sstore = [&](int st) {  // registers -> LDS: the activation slab as it is, the weights decoded
        const bool col_ok = !RAGGED || 32 * GST * st + 8 * achunk < n;  // in_features a multiple of 64: no select at all
#pragma unroll
        for (int i = 0; i < NLD; ++i)
            *reinterpret_cast<g_u32x4*>(As + (arow0 + RPP * i) * GPITCH + 16 * achunk) = col_ok ? xa[i] : g_u32x4{0u, 0u, 0u, 0u};
#pragma unroll
        for (int g = 0; g < GST; ++g)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                uint32_t bits = wl[2 * g + j] >> sh[j];
                if (STRADDLE) bits = (uint32_t)((((uint64_t)wh[2 * g + j] << 32) | wl[2 * g + j]) >> sh[j]);
                g_u32x4 b;
                if (BITS == 4) {
                    const uint32_t hib = ((tb >> 8) & 0xffu) * 0x01010101u;
                    const uint32_t lo = (bits & 0x0f0f0f0fu) | hib, hi = ((bits >> 4) & 0x0f0f0f0fu) | hib;
#pragma unroll
                    for (int p = 0; p < 4; ++p)
                        b[p] = g_lds_pair(__builtin_amdgcn_perm(lo, lane4, 0x0c0c0000u | ((4u + p) << 8)),
                                          __builtin_amdgcn_perm(hi, lane4, 0x0c0c0000u | ((4u + p) << 8)));
                } else {
                    const uint32_t base = tb + lane4;
#pragma unroll
                    for (int p = 0; p < 4; ++p)
                        b[p] = g_lds_pair((((bits >> (BITS * (2 * p))) & (V - 1)) << 8) + base,
                                          (((bits >> (BITS * (2 * p + 1))) & (V - 1)) << 8) + base);
                }
                *reinterpret_cast<g_u32x4*>(Bs + dfeat * GPITCH + 64 * g + 16 * (dq0 + 2 * j)) = b;
            }
    };

    g_f32x16 acc[2][4];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[r][t][i] = 0.f;

    gload(st_begin);
    __syncthreads();  // codebook table
    sstore(st_begin);
    __syncthreads();  // first slab + weight tile
    if (st_begin + 1 < nst) gload(st_begin + 1);
    const unsigned char* abase = As + (64 * wv + l31) * GPITCH + 16 * hf;
    const unsigned char* bbase = Bs + l31 * GPITCH + 16 * hf;
    for (int st = st_begin; st < nst; ++st) {
        // ---- 2 * GST sub-steps of 16 in_features; the fragments of sub-step u + 1 are read before the matrix instructions of u
        g_u32x4 a_cur[2], b_cur[4];
#pragma unroll
        for (int r = 0; r < 2; ++r) a_cur[r] = *reinterpret_cast<const g_u32x4*>(abase + 32 * r * GPITCH);
#pragma unroll
        for (int t = 0; t < 4; ++t) b_cur[t] = *reinterpret_cast<const g_u32x4*>(bbase + 32 * t * GPITCH);
#pragma unroll
        for (int u = 0; u < 2 * GST; ++u) {
            g_u32x4 a_nxt[2], b_nxt[4];
            if (u + 1 < 2 * GST) {
#pragma unroll
                for (int r = 0; r < 2; ++r) a_nxt[r] = *reinterpret_cast<const g_u32x4*>(abase + 32 * r * GPITCH + 32 * (u + 1));
#pragma unroll
                for (int t = 0; t < 4; ++t) b_nxt[t] = *reinterpret_cast<const g_u32x4*>(bbase + 32 * t * GPITCH + 32 * (u + 1));
            }
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if (BF16)
                        acc[r][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(g_bf16x8, a_cur[r]),
                                                                            __builtin_bit_cast(g_bf16x8, b_cur[t]), acc[r][t], 0, 0, 0);
                    else
                        acc[r][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(g_f16x8, a_cur[r]),
                                                                           __builtin_bit_cast(g_f16x8, b_cur[t]), acc[r][t], 0, 0, 0);
                }
            if (u + 1 < 2 * GST) {
#pragma unroll
                for (int r = 0; r < 2; ++r) a_cur[r] = a_nxt[r];
#pragma unroll
                for (int t = 0; t < 4; ++t) b_cur[t] = b_nxt[t];
            }
        }
        if (st + 1 < nst) {
            __syncthreads();  // every wave has read this stage's tiles
            sstore(st + 1);   // the next stage (its loads were issued a stage ago)
            __syncthreads();
            if (st + 2 < nst) gload(st + 2);
        }
    }

    // C layout of the 32x32 tile: column (feature) = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    if constexpr (SPLIT) {  // this split's fp32 partial tile; lut_gemm_reduce_kernel (next launch on the stream) sums the splits
        const int ks = blockIdx.y;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int o = o0 + 32 * t + l31;
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int row = r0 + 64 * wv + 32 * r + (reg & 3) + 8 * (reg >> 2) + 4 * hf;
                    if (o < m && row < M) partial[((int64_t)ks * M + row) * m + o] = acc[r][t][reg];
                }
        }
        return;
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int o = o0 + 32 * t + l31;
        if (o >= m) continue;
        float bv = 0.f;
        if (bias) bv = BF16 ? __builtin_bit_cast(float, (uint32_t)bias[o] << 16) : (float)__builtin_bit_cast(_Float16, bias[o]);
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) acc[r][t][reg] += bv;
    }
    auto emit = [&](auto with_addend) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int o = o0 + 32 * t + l31;
            if (o >= m) continue;
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int row = r0 + 64 * wv + 32 * r + (reg & 3) + 8 * (reg >> 2) + 4 * hf;
                    if (row >= M) continue;
                    float v = acc[r][t][reg];
                    if (decltype(with_addend)::value) v += addend[(int64_t)row * m + o];
                    y[(int64_t)row * m + o] = BF16 ? __builtin_bit_cast(uint16_t, (__bf16)v) : __builtin_bit_cast(uint16_t, (_Float16)v);
                }
        }
    };
    if (addend) emit(std::true_type{});
    else emit(std::false_type{});
}

// ---------------------------------------------------------------------------------------------------------------------------
// Pipelined variant for launches with at least one whole-K tile per CU (prefill at seqlen >= 2048): ONE workgroup per CU, both
// tiles double-buffered in LDS (2 x 36 KB activations, 2 x 18 KB decoded weights), one barrier per stage.  While a wave issues
// the 32 matrix instructions of stage st (fragments of the next sub-step read one sub-step ahead), it files stage st + 1 --
// whose global loads were issued a whole stage earlier -- into the other buffers: two 16-byte activation stores and one decoded
// octet between the matrix instructions of each sub-step; the loads of stage st + 2 are issued first.  Registers: two sets of
// staging registers rotate (the stage loop is unrolled by two), 128 accumulators; the kernel runs at one wave per SIMD.
template <int BITS, bool BF16, bool RAGGED>
__global__ __launch_bounds__(256, 1) void lut_gemm_pipe_kernel(const uint16_t* __restrict__ x, const uint32_t* __restrict__ qw,
                                                             const uint16_t* __restrict__ lut, const uint16_t* __restrict__ bias,
                                                             const float* __restrict__ addend, int M, int m, int n, int tiles_m,
                                                             int tiles_n, uint16_t* __restrict__ y) {
    constexpr int V = 1 << BITS;
    constexpr bool STRADDLE = (8 * BITS) % 16 != 0;
    constexpr int TPR = 4 * GST, RPP = 256 / TPR, NLD = GBM / RPP, NIT = 2 * GST;
    constexpr int ABYTES = GBM * GPITCH, BBYTES = GBN * GPITCH;
    extern __shared__ __attribute__((aligned(4096))) unsigned char gsm[];
    uint32_t(*tbl)[V][64] = reinterpret_cast<uint32_t(*)[V][64]>(gsm);  // [2][V][64], 4 KB-aligned blocks (v_perm addressing)
    unsigned char* As = gsm + 2 * V * 64 * 4;   // [2][ABYTES]
    unsigned char* Bs = As + 2 * ABYTES;         // [2][BBYTES]

    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hf = lane >> 5;
    int bid = blockIdx.x;
    {
        const int total = tiles_m * tiles_n;
        const int per = total >> 3;
        if ((total & 7) == 0) bid = (bid & 7) * per + (bid >> 3);
    }
    const int tm = bid / tiles_n, tn = bid % tiles_n;
    const int r0 = tm * GBM, o0 = tn * GBN;
    const int nkb = n >> 5, nst = (nkb + GST - 1) / GST;

    const int dfeat = tid & 127, dq0 = __builtin_amdgcn_readfirstlane(tid >> 7);  // wave-uniform: waves 0, 1 / 2, 3
    const int dcol = min(o0 + dfeat, m - 1);
    if (wv < 2) {
        const uint32_t* lp = reinterpret_cast<const uint32_t*>(lut + (int64_t)dcol * V);
        uint32_t h[V / 2];
#pragma unroll
        for (int e = 0; e < V / 2; ++e) h[e] = lp[e];
#pragma unroll
        for (int e = 0; e < V / 2; ++e) {
            tbl[wv][2 * e][lane] = h[e] & 0xffffu;
            tbl[wv][2 * e + 1][lane] = h[e] >> 16;
        }
    }
    int wi[2], sh[2], wi2[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int off = 8 * BITS * (dq0 + 2 * j);
        wi[j] = off >> 5;
        sh[j] = off & 31;
        wi2[j] = STRADDLE ? min(wi[j] + 1, BITS - 1) : wi[j];
    }
    const int achunk = tid % TPR, arow0 = tid / TPR;
    // buffer loads: this thread's constant byte offsets in VGPRs, the stage's offset in an SGPR -- no per-load 64-bit address
    // arithmetic (the host takes this kernel only while both operands stay below 2 GB); activation rows past M lie past the
    // resource's end and read as zeros
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(x), 0, (int)((int64_t)M * n * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(qw), 0, (int)((int64_t)(n >> 5) * BITS * m * 4), 0x00020000);
    int xoff[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) xoff[i] = ((r0 + arow0 + RPP * i) * n + 8 * achunk) * 2;
    const int woff = dcol * 4;

    struct Stage {  // staging registers of one stage
        g_u32x4 xa[NLD];
        uint32_t wl[NIT], wh[NIT];
    };
    auto gload = [&](int st, Stage& R) {
        const int soff = 64 * GST * st;  // bytes along a row; a ragged last stage reads the next row's head, zeroed when filed
#pragma unroll
        for (int i = 0; i < NLD; ++i)
            // (ragged: the stage offset rides in the range-checked VGPR offset, so the last row's overshoot reads zeros whatever
            // the hardware does with the scalar offset in its bounds check)
            R.xa[i] = __builtin_bit_cast(g_u32x4, RAGGED ? __builtin_amdgcn_raw_buffer_load_b128(rsx, xoff[i] + soff, 0, 0)
                                                        : __builtin_amdgcn_raw_buffer_load_b128(rsx, xoff[i], soff, 0));
#pragma unroll
        for (int g = 0; g < GST; ++g) {
            const int kb = RAGGED ? min(GST * st + g, nkb - 1) : GST * st + g;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                R.wl[2 * g + j] = __builtin_amdgcn_raw_buffer_load_b32(rsw, woff, (kb * BITS + wi[j]) * m * 4, 0);
                if (STRADDLE) R.wh[2 * g + j] = __builtin_amdgcn_raw_buffer_load_b32(rsw, woff, (kb * BITS + wi2[j]) * m * 4, 0);
            }
        }
    };
    const uint32_t tb = (uint32_t)(uintptr_t)(&tbl[0][0][0]) + (uint32_t)(wv & 1) * (V * 256u);
    const uint32_t lane4 = 4u * lane;
    // one quarter of the hand-over of a stage, in two halves with the sub-step's matrix instructions between them: first the two
    // activation chunks are stored and the eight codebook lookups of one octet ISSUED (raw dwords), afterwards the lookups are
    // packed and the octet stored -- the LDS round trip of the lookups passes under the matrix instructions (a wave issues in
    // order: consumed right where they are issued, every lookup group exposed its latency)
    struct Raw {
        uint32_t lo[4], hi[4];
    };
    auto file_issue = [&](int part, const Stage& R, int st, int buf, Raw& raw) {
        const bool col_ok = !RAGGED || 32 * GST * st + 8 * achunk < n;  // in_features a multiple of 64: no select at all
        unsigned char* Ab = As + buf * ABYTES;
#pragma unroll
        for (int i = 2 * part; i < 2 * part + 2; ++i)
            *reinterpret_cast<g_u32x4*>(Ab + (arow0 + RPP * i) * GPITCH + 16 * achunk) = col_ok ? R.xa[i] : g_u32x4{0u, 0u, 0u, 0u};
        const int j = part & 1;
        uint32_t bits = R.wl[part] >> sh[j];
        if (STRADDLE) bits = (uint32_t)((((uint64_t)R.wh[part] << 32) | R.wl[part]) >> sh[j]);
        typedef const uint32_t __attribute__((address_space(3))) * lds_u32;
        if (BITS == 4) {
            const uint32_t hib = ((tb >> 8) & 0xffu) * 0x01010101u;
            const uint32_t lo = (bits & 0x0f0f0f0fu) | hib, hi = ((bits >> 4) & 0x0f0f0f0fu) | hib;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                raw.lo[p] = *reinterpret_cast<lds_u32>(__builtin_amdgcn_perm(lo, lane4, 0x0c0c0000u | ((4u + p) << 8)));
                raw.hi[p] = *reinterpret_cast<lds_u32>(__builtin_amdgcn_perm(hi, lane4, 0x0c0c0000u | ((4u + p) << 8)));
            }
        } else {
            const uint32_t base = tb + lane4;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                raw.lo[p] = *reinterpret_cast<lds_u32>((((bits >> (BITS * (2 * p))) & (V - 1)) << 8) + base);
                raw.hi[p] = *reinterpret_cast<lds_u32>((((bits >> (BITS * (2 * p + 1))) & (V - 1)) << 8) + base);
            }
        }
    };
    auto file_finish = [&](int part, int buf, const Raw& raw) {
        unsigned char* Bb = Bs + buf * BBYTES;
        const int g = part >> 1, j = part & 1;
        g_u32x4 b;
#pragma unroll
        for (int p = 0; p < 4; ++p) b[p] = raw.lo[p] | (raw.hi[p] << 16);
        *reinterpret_cast<g_u32x4*>(Bb + dfeat * GPITCH + 64 * g + 16 * (dq0 + 2 * j)) = b;
    };
    static_assert(NLD == 8 && NIT == 4, "the hand-over is split into four parts of two chunks and one octet");

    g_f32x16 acc[2][4];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[r][t][i] = 0.f;

    const int aoff = (64 * wv + l31) * GPITCH + 16 * hf, boff = l31 * GPITCH + 16 * hf;
    // one stage: loads of st + 2 first, then four sub-steps, each: fragments of the next sub-step, eight matrix
    // instructions, a quarter of the hand-over of stage st + 1
    auto stage = [&](int st, Stage& Rnext, Stage& Rfree) {
        const int buf = st & 1;
        // (no branches inside a stage: the instruction scheduler's interleaving hints below work on ONE basic block; past the
        // end the loads re-read the last stage and the hand-over files into a buffer nobody reads any more)
        gload(min(st + 2, nst - 1), Rfree);
        const unsigned char* ab = As + buf * ABYTES + aoff;
        const unsigned char* bb = Bs + buf * BBYTES + boff;
        g_u32x4 a_cur[2], b_cur[4];
#pragma unroll
        for (int r = 0; r < 2; ++r) a_cur[r] = *reinterpret_cast<const g_u32x4*>(ab + 32 * r * GPITCH);
#pragma unroll
        for (int t = 0; t < 4; ++t) b_cur[t] = *reinterpret_cast<const g_u32x4*>(bb + 32 * t * GPITCH);
#pragma unroll
        for (int u = 0; u < 2 * GST; ++u) {
            g_u32x4 a_nxt[2], b_nxt[4];
            if (u + 1 < 2 * GST) {
#pragma unroll
                for (int r = 0; r < 2; ++r) a_nxt[r] = *reinterpret_cast<const g_u32x4*>(ab + 32 * r * GPITCH + 32 * (u + 1));
#pragma unroll
                for (int t = 0; t < 4; ++t) b_nxt[t] = *reinterpret_cast<const g_u32x4*>(bb + 32 * t * GPITCH + 32 * (u + 1));
            }
            Raw raw;
            file_issue(u, Rnext, st + 1, buf ^ 1, raw);
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if (BF16)
                        acc[r][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(g_bf16x8, a_cur[r]),
                                                                            __builtin_bit_cast(g_bf16x8, b_cur[t]), acc[r][t], 0, 0, 0);
                    else
                        acc[r][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(g_f16x8, a_cur[r]),
                                                                           __builtin_bit_cast(g_f16x8, b_cur[t]), acc[r][t], 0, 0, 0);
                }
            file_finish(u, buf ^ 1, raw);
            if (u + 1 < 2 * GST) {
#pragma unroll
                for (int r = 0; r < 2; ++r) a_cur[r] = a_nxt[r];
#pragma unroll
                for (int t = 0; t < 4; ++t) b_cur[t] = b_nxt[t];
            }
        }
        __syncthreads();
    };

    Stage R0, R1;
    gload(0, R0);
    __syncthreads();  // codebook table
#pragma unroll
    for (int part = 0; part < 4; ++part) {
        Raw raw;
        file_issue(part, R0, 0, 0, raw);
        file_finish(part, 0, raw);
    }
    if (nst > 1) gload(1, R1);
    __syncthreads();  // stage 0 filed
    for (int st = 0; st < nst; st += 2) {
        stage(st, R1, R0);                      // files stage st + 1 from R1, loads st + 2 into R0
        if (st + 1 < nst) stage(st + 1, R0, R1);  // files stage st + 2 from R0, loads st + 3 into R1
    }

#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int o = o0 + 32 * t + l31;
        if (o >= m) continue;
        float bv = 0.f;
        if (bias) bv = BF16 ? __builtin_bit_cast(float, (uint32_t)bias[o] << 16) : (float)__builtin_bit_cast(_Float16, bias[o]);
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) acc[r][t][reg] += bv;
    }
    auto emit = [&](auto with_addend) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int o = o0 + 32 * t + l31;
            if (o >= m) continue;
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int row = r0 + 64 * wv + 32 * r + (reg & 3) + 8 * (reg >> 2) + 4 * hf;
                    if (row >= M) continue;
                    float v = acc[r][t][reg];
                    if (decltype(with_addend)::value) v += addend[(int64_t)row * m + o];
                    y[(int64_t)row * m + o] = BF16 ? __builtin_bit_cast(uint16_t, (__bf16)v) : __builtin_bit_cast(uint16_t, (_Float16)v);
                }
        }
    };
    if (addend) emit(std::true_type{});
    else emit(std::false_type{});
}
constexpr size_t GPIPE_LDS = 2 * 16 * 64 * 4 + 2 * (size_t)GBM * GPITCH + 2 * (size_t)GBN * GPITCH;  // table sized for V = 16

template <bool BF16>
__global__ __launch_bounds__(256) void lut_gemm_reduce_kernel(const float* __restrict__ partial, const uint16_t* __restrict__ bias,
                                                              const float* __restrict__ addend, int64_t total, int m, int KS,
                                                              uint16_t* __restrict__ y) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;  // element of y, row-major
    if (i >= total) return;
    float v = 0.f;
    for (int ks = 0; ks < KS; ++ks) v += partial[(int64_t)ks * total + i];  // split order
    if (addend) v += addend[i];
    if (bias) {
        const uint16_t b = bias[i % m];
        v += BF16 ? __builtin_bit_cast(float, (uint32_t)b << 16) : (float)__builtin_bit_cast(_Float16, b);
    }
    y[i] = BF16 ? __builtin_bit_cast(uint16_t, (__bf16)v) : __builtin_bit_cast(uint16_t, (_Float16)v);
}

struct GemmPlan {
    int tiles_m, tiles_n, KS, st_per;
    size_t partial_bytes;
};
GemmPlan lut_gemm_plan(int64_t M, int64_t m, int64_t n) {
    GemmPlan p;
    p.tiles_m = (int)((M + GBM - 1) / GBM);
    p.tiles_n = (int)((m + GBN - 1) / GBN);
    const int nst = (int)(((n >> 5) + GST - 1) / GST);
    const int64_t tiles = (int64_t)p.tiles_m * p.tiles_n;
    // split in_features while the launch has fewer workgroups than CUs, every split keeping at least 8 stages (512 columns)
    const int force = (int)opt_get(OPT_LUT_GEMM_RM);  // developer switch: forced split factor
    int ks = 1;
    const int ncu = std::max(1, current_device_cus());
    while (ks < 8 && tiles * ks < ncu && nst / (2 * ks) >= 8) ks *= 2;
    if (force > 0) ks = std::min(force, nst);
    p.st_per = (nst + ks - 1) / ks;
    p.KS = (nst + p.st_per - 1) / p.st_per;
    p.partial_bytes = p.KS > 1 ? align_up((size_t)p.KS * (size_t)M * (size_t)m * sizeof(float), 256) : 0;
    return p;
}

template <int BITS>
static int launch_gemm_bits(const void* x, const uint32_t* qw, const void* lut, const void* bias, const float* addend, int dtype, int M,
                            int m, int n, void* y, const GemmPlan& p, float* partial, hipStream_t stream) {
    const uint16_t* xp = static_cast<const uint16_t*>(x);
    const uint16_t* lp = static_cast<const uint16_t*>(lut);
    const uint16_t* bp = static_cast<const uint16_t*>(bias);
    uint16_t* yp = static_cast<uint16_t*>(y);
    const dim3 grid((unsigned)(p.tiles_m * p.tiles_n), (unsigned)p.KS);
#define GANQ_GEMM_LAUNCH(BF, SP)                                                                                                        \
    do {                                                                                                                                \
        if ((n & 63) != 0)                                                                                                              \
            hipLaunchKernelGGL((lut_gemm_kernel<BITS, BF, SP, true>), grid, dim3(256), 0, stream, xp, qw, lp, bp, addend, M, m, n,      \
                               p.tiles_m, p.tiles_n, yp, p.st_per, partial);                                                            \
        else                                                                                                                            \
            hipLaunchKernelGGL((lut_gemm_kernel<BITS, BF, SP, false>), grid, dim3(256), 0, stream, xp, qw, lp, bp, addend, M, m, n,     \
                               p.tiles_m, p.tiles_n, yp, p.st_per, partial);                                                            \
    } while (0)
    const int pipe_opt = (int)opt_get(OPT_LUT_GEMM_PIPE);
    // measured (4-bit, fp16, both kernels with buffer loads): about one tile per CU (4096 x 4096, M = 2048: 256 tiles) 87 us
    // pipelined vs 103 us with the two-workgroups-per-CU kernel; 512 tiles (M = 4096) 175 vs 164; 896 tiles (14336 x 4096,
    // M = 2048) 310 vs 287; 1792 tiles (M = 4096) 556 vs 520: the pipelined kernel where a CU gets one tile, the other above
    const int64_t ntile = (int64_t)p.tiles_m * p.tiles_n;
    const int64_t ncu = std::max(1, current_device_cus());
    const bool pipe_by_shape = 8 * ntile >= 7 * ncu && 4 * ntile < 7 * ncu;  // 224 <= tiles < 448 on the 256 CUs of an MI355X
    // the pipelined kernel addresses both operands through 32-bit buffer offsets (incl. one stage of look-ahead)
    const bool fits32 = ((int64_t)M + GBM) * n * 2 < (1ll << 31) && (int64_t)(n >> 5) * BITS * m * 4 < (1ll << 31);
    if (!fits32)
        return fail(-1, "ganq_lut_linear_fwd: x (%lld x %lld) or qweight beyond the 2 GB window of the GEMM's buffer loads: split the "
                        "rows of x over several calls", (long long)M, (long long)n);
    const bool pipe = p.KS == 1 && (pipe_opt >= 0 ? pipe_opt != 0 : pipe_by_shape);
    if (pipe) {
        const bool ragged = (n & 63) != 0;
#define GANQ_PIPE_LAUNCH(BF, RG)                                                                                                   \
    do {                                                                                                                           \
        int rc_ = ensure_dynamic_lds(reinterpret_cast<const void*>(lut_gemm_pipe_kernel<BITS, BF, RG>), GPIPE_LDS);                  \
        if (rc_) return rc_;                                                                                                       \
        hipLaunchKernelGGL((lut_gemm_pipe_kernel<BITS, BF, RG>), grid, dim3(256), GPIPE_LDS, stream, xp, qw, lp, bp, addend, M, m, \
                           n, p.tiles_m, p.tiles_n, yp);                                                                           \
    } while (0)
        if (dtype == 1) {
            if (ragged) GANQ_PIPE_LAUNCH(true, true);
            else GANQ_PIPE_LAUNCH(true, false);
        } else {
            if (ragged) GANQ_PIPE_LAUNCH(false, true);
            else GANQ_PIPE_LAUNCH(false, false);
        }
#undef GANQ_PIPE_LAUNCH
    } else if (p.KS > 1) {
        const int64_t total = (int64_t)M * m;
        const dim3 rgrid((unsigned)((total + 255) / 256));
        if (dtype == 1) {
            GANQ_GEMM_LAUNCH(true, true);
            hipLaunchKernelGGL(lut_gemm_reduce_kernel<true>, rgrid, dim3(256), 0, stream, partial, bp, addend, total, m, p.KS, yp);
        } else {
            GANQ_GEMM_LAUNCH(false, true);
            hipLaunchKernelGGL(lut_gemm_reduce_kernel<false>, rgrid, dim3(256), 0, stream, partial, bp, addend, total, m, p.KS, yp);
        }
    } else {
        if (dtype == 1) GANQ_GEMM_LAUNCH(true, false);
        else GANQ_GEMM_LAUNCH(false, false);
    }
#undef GANQ_GEMM_LAUNCH
    GANQ_LAUNCH_CHECK();
    return 0;
}

// called by lut_linear_fwd (lut_linear.hip) for M > 64; arguments are already validated there
size_t lut_gemm_workspace_bytes(int64_t M, int64_t m, int64_t n) { return lut_gemm_plan(M, m, n).partial_bytes; }

int lut_gemm(const void* x, const uint32_t* qw, const void* lut, const void* bias, const float* addend, int dtype, int64_t M,
             int64_t m, int64_t n, int bits, void* y, float* partial, size_t partial_bytes, hipStream_t stream) {
    if (M > INT32_MAX / 2) return fail(-1, "ganq_lut_linear_fwd: M too large");
    const GemmPlan p = lut_gemm_plan(M, m, n);
    if (p.KS > 1 && (!partial || partial_bytes < p.partial_bytes))
        return fail(-4, "ganq_lut_linear_fwd: workspace %zu B < required %zu B of partial tiles", partial_bytes, p.partial_bytes);
    if (bits == 2) return launch_gemm_bits<2>(x, qw, lut, bias, addend, dtype, (int)M, (int)m, (int)n, y, p, partial, stream);
    if (bits == 3) return launch_gemm_bits<3>(x, qw, lut, bias, addend, dtype, (int)M, (int)m, (int)n, y, p, partial, stream);
    return launch_gemm_bits<4>(x, qw, lut, bias, addend, dtype, (int)M, (int)m, (int)n, y, p, partial, stream);
}

}  // namespace ganq

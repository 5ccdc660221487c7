// LUT-dequant linear forward and index packing.
//   y[M,m] = x[M,n] @ dequant(qweight, lut)^T + bias        (replaces FakeQuantLinear.forward, fake.py:88-89,
//   whose weight is T.gather(1,Q).half(); the reference keeps neither Q nor T, ganq.py:633-646)
//
// Storage: `bits`-wide indices packed as one little-endian bit stream per output feature along in_features,
// 32-bit words laid out qweight[n*bits/32][m] -- exactly the GPTQ int32 packing (qlinear/__init__.py:508-538,
// including its 3-bit 32-in-3-words scheme, which is the same bit stream); lut [m][V] in the activation dtype.
//
// ganq_lut_linear_fwd (M <= 64 rows, decode / small batches): weights are decoded straight into the B operand of
//   v_mfma_f32_16x16x32_{f16,bf16} (codebook lookups in LDS), activations are the A operand, fp32 accumulation;
//   the in_features range is split over workgroups and waves (deterministic two-stage reduction).
//   M > 64: lut_gemm.hip (fused LUT-dequant GEMM).
// ganq_lut_dequant: materialises W_q [m,n] in the activation dtype (dequantize_weight(), tests, A/B runs against a library GEMM).
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

constexpr int LW = 4;          // waves per workgroup
constexpr int LUT_MAX_M = 64;  // rows of x one call takes (4 row tiles of 16)
constexpr int LUT_FB = 32 * LW;  // output features per workgroup (split-K-through-memory variant)
constexpr int LWK = 16;          // waves per workgroup of the in-workgroup split variant

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// two codebook entries (dword slots holding a 16-bit value) -> one packed register.  (The d16 / d16_hi LDS loads
// cannot be used for this: with SRAM-ECC on, as on gfx950, they overwrite the whole register.)
__device__ __forceinline__ uint32_t lds_pair(uint32_t addr_lo, uint32_t addr_hi) {
    typedef const uint32_t __attribute__((address_space(3))) * lds_u32;
    return *reinterpret_cast<lds_u32>(addr_lo) | (*reinterpret_cast<lds_u32>(addr_hi) << 16);
}

// decode kernel on the 16x16x32 matrix-core tile: D[row][feature] += x[row][k] * Wq[feature][k].
//   B operand (weights): lane l holds feature l%16 and the 8 consecutive in_features 8*(l/16).. of a 32-column
//   group, i.e. exactly 8*BITS consecutive bits of that feature's stream: one word load (two for 3-bit), 8 codebook
//   lookups, 4 packs.  A operand (activations): lane l holds row l%16, the same 8
//   in_features: one 16 B load.
//   A workgroup owns 128 features (each wave two B tiles sharing A: the workgroup reads 512 B of every word row)
//   and a range of 32-column groups (blockIdx.y); the waves share the activation loads through L1.
//   tbl[wave][t][e][lane]: codebook entry e of lane's feature in tile t, one dword slot per lane and entry, so the
//   per-lane dynamic lookup is bank-conflict free.
//   Split-K: each workgroup stores its partial tile, takes a ticket on the feature block's counter, and the last
//   one to arrive sums the partials in ks order (deterministic), adds the bias and writes y; it leaves the counter 0.
// NW waves per workgroup.  INWG = false: every wave owns its own 32 features, blockIdx.y splits in_features and the
// partial tiles meet in memory (ticket).  INWG = true: the NW waves share ONE set of 32 features and split
// in_features among themselves; the partial tiles meet in LDS -- no second memory round trip, which at decode sizes
// is what the kernel's time is made of.
template <int BITS, int RT, bool BF16, int NW, bool INWG>
__global__ __launch_bounds__(NW * 64) void lut_mfma_kernel(const uint16_t* __restrict__ x, const uint32_t* __restrict__ qw,
                                                          const uint16_t* __restrict__ lut, const uint16_t* __restrict__ bias,
                                                          const float* __restrict__ addend, int M, int m, int n, int kb_per_wg,
                                                          int KS,
                                                          float* __restrict__ partial, int* __restrict__ counters,
                                                          uint16_t* __restrict__ y) {
    constexpr int V = 1 << BITS;
    constexpr int KC = 4;  // groups in flight per wave (loads issued one chunk ahead)
    constexpr bool STRADDLE = (8 * BITS) % 16 != 0;  // 3-bit: a lane's 24 bits can span two words
    __shared__ __attribute__((aligned(4096))) uint32_t tbl[INWG ? 1 : NW][2][V][64];  // 4 KB alignment: see the v_perm addressing
    __shared__ int s_ticket;
    __shared__ float red[INWG ? NW - 1 : 1][2][RT][4][64];

    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 15, q = lane >> 4;
    const int o0 = INWG ? blockIdx.x * 32 : blockIdx.x * (32 * NW) + 32 * wv, ks = blockIdx.y;
    const int nkb = n >> 5;
    // INWG: kb_per_wg is the per-WAVE share of the groups; a wave past the end simply has an empty range
    const int kb_begin = INWG ? min(nkb, wv * kb_per_wg) : ks * kb_per_wg;
    const int kb_end = min(nkb, kb_begin + kb_per_wg);
    const int off = 8 * BITS * q, wi = off >> 5, sh = off & 31;
    const int wi2 = STRADDLE ? min(wi + 1, BITS - 1) : wi;
    const int oc0 = min(o0 + col, m - 1), oc1 = min(o0 + 16 + col, m - 1);

    uint32_t wl[2][KC][2], wh[2][KC][2];
    u32x4 xa[2][KC][RT];
    auto issue = [&](int buf, int kb0) {
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            const int kb = kb0 + c;
            const bool ok = kb < kb_end;
            const int kbc = ok ? kb : min(kb_begin, nkb - 1);
            const int64_t row = (int64_t)(kbc * BITS + wi) * m, row2 = (int64_t)(kbc * BITS + wi2) * m;
            wl[buf][c][0] = qw[row + oc0];
            wl[buf][c][1] = qw[row + oc1];
            if (STRADDLE) {
                wh[buf][c][0] = qw[row2 + oc0];
                wh[buf][c][1] = qw[row2 + oc1];
            }
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                const int xr = 16 * r + col;
                u32x4 v = {0u, 0u, 0u, 0u};
                if (ok && xr < M) v = *reinterpret_cast<const u32x4*>(x + (int64_t)xr * n + 32 * kbc + 8 * q);
                xa[buf][c][r] = v;
            }
        }
    };
    f32x4 acc[2][RT];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < RT; ++r) acc[t][r] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (kb_begin < kb_end) issue(0, kb_begin);  // weights and activations are in flight while the table is built
    {   // every lane fetches the V entries of its own feature (rows of lut are 2V bytes, 4-byte aligned)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const uint32_t* lp = reinterpret_cast<const uint32_t*>(lut + (int64_t)(t ? oc1 : oc0) * V);
            uint32_t h[V / 2];
#pragma unroll
            for (int e = 0; e < V / 2; ++e) h[e] = lp[e];
#pragma unroll
            for (int e = 0; e < V / 2; ++e) {
                tbl[INWG ? 0 : wv][t][2 * e][lane] = h[e] & 0xffffu;  // INWG: every wave writes the same values
                tbl[INWG ? 0 : wv][t][2 * e + 1][lane] = h[e] >> 16;
            }
        }
    }
    __syncthreads();  // tbl ready (each wave reads only its own part; the barrier orders the LDS writes)

    // LDS byte address of entry e: tb0 + 256*e + 4*lane (+ 256*V for tile 1)
    const uint32_t tb0 = (uint32_t)(uintptr_t)(&tbl[0][0][0][0]) + (INWG ? 0u : (uint32_t)wv * (2u * V * 256u));
    const uint32_t lane4 = 4u * lane;
    auto consume = [&](int buf) {
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            u32x4 b[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                uint32_t bits = wl[buf][c][t] >> sh;
                if (STRADDLE) bits = (uint32_t)((((uint64_t)wh[buf][c][t] << 32) | wl[buf][c][t]) >> sh);
                const uint32_t tb = tb0 + (uint32_t)t * (V * 256u);
                if (BITS == 4) {
                    // byte k of lo/hi = entry index of element 2k / 2k+1, OR-ed with byte 1 of the (4 KB aligned) tile base:
                    // one v_perm per address ({byte1, byte0} = {index | base, 4*lane})
                    const uint32_t hib = ((tb >> 8) & 0xffu) * 0x01010101u;
                    const uint32_t lo = (bits & 0x0f0f0f0fu) | hib, hi = ((bits >> 4) & 0x0f0f0f0fu) | hib;
#pragma unroll
                    for (int p = 0; p < 4; ++p)
                        b[t][p] = lds_pair(__builtin_amdgcn_perm(lo, lane4, 0x0c0c0000u | ((4u + p) << 8)),
                                               __builtin_amdgcn_perm(hi, lane4, 0x0c0c0000u | ((4u + p) << 8)));
                } else {
                    const uint32_t base = tb + lane4;
#pragma unroll
                    for (int p = 0; p < 4; ++p)
                        b[t][p] = lds_pair((((bits >> (BITS * (2 * p))) & (V - 1)) << 8) + base,
                                               (((bits >> (BITS * (2 * p + 1))) & (V - 1)) << 8) + base);
                }
            }
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < RT; ++r) {
                    if (BF16)
                        acc[t][r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, xa[buf][c][r]),
                                                                            __builtin_bit_cast(bf16x8, b[t]), acc[t][r], 0, 0, 0);
                    else
                        acc[t][r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, xa[buf][c][r]),
                                                                           __builtin_bit_cast(f16x8, b[t]), acc[t][r], 0, 0, 0);
                }
        }
    };
    // groups past kb_end carry zero activations, so a ragged last chunk needs no special case
    for (int kb0 = kb_begin; kb0 < kb_end; kb0 += 2 * KC) {
        if (kb0 + KC < kb_end) issue(1, kb0 + KC);
        consume(0);
        if (kb0 + KC < kb_end) {
            if (kb0 + 2 * KC < kb_end) issue(0, kb0 + 2 * KC);
            consume(1);
        }
    }

    auto finish = [&](float v, int row, int o) {
        if (addend) v += addend[(int64_t)row * m + o];  // fp32 [M, m]: the sparse-outlier product (outlier.hip)
        if (bias) v += BF16 ? __builtin_bit_cast(float, (uint32_t)bias[o] << 16) : (float)__builtin_bit_cast(_Float16, bias[o]);
        y[(int64_t)row * m + o] = BF16 ? __builtin_bit_cast(uint16_t, (__bf16)v) : __builtin_bit_cast(uint16_t, (_Float16)v);
    };
    if constexpr (INWG) {
        if (wv > 0) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < RT; ++r)
#pragma unroll
                    for (int i = 0; i < 4; ++i) red[wv - 1][t][r][i][lane] = acc[t][r][i];
        }
        __syncthreads();
        if (wv == 0) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < RT; ++r)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float v = acc[t][r][i];
#pragma unroll
                        for (int w2 = 0; w2 < NW - 1; ++w2) v += red[w2][t][r][i][lane];  // fixed order
                        const int row = 16 * r + 4 * q + i, o = o0 + 16 * t + col;
                        if (row < M && o < m) finish(v, row, o);
                    }
        }
        return;
    }
    if (KS == 1) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < RT; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = 16 * r + 4 * q + i, o = o0 + 16 * t + col;
                    if (row < M && o < m) finish(acc[t][r][i], row, o);
                }
        return;
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = 16 * r + 4 * q + i, o = o0 + 16 * t + col;
                if (row < M && o < m)
                    __hip_atomic_store(&partial[((int64_t)ks * M + row) * m + o], acc[t][r][i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
    // The partial tiles are written and read with device-scope (L2-bypassing, write-through) accesses, so no cache
    // write-back / invalidate is needed around the ticket (a full __threadfence() costs ~60 us here: it walks L2):
    // once vmcnt reaches 0 the stores are visible to every XCD, and the ticket is taken after that.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) s_ticket = __hip_atomic_fetch_add(&counters[blockIdx.x], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (s_ticket != KS - 1) return;
    // all of this lane's outputs at once, KBT splits per batch: 64 independent loads in flight, sums in ks order
    constexpr int KBT = 8 / RT;
    float s[RT][4][2];
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i) s[r][i][0] = s[r][i][1] = 0.f;
    for (int k0 = 0; k0 < KS; k0 += KBT) {
        float v[KBT][RT][4][2];
#pragma unroll
        for (int u = 0; u < KBT; ++u)
#pragma unroll
            for (int r = 0; r < RT; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const int row = 16 * r + 4 * q + i, o = o0 + 16 * t + col;
                        v[u][r][i][t] = (k0 + u < KS && row < M && o < m)
                                            ? __hip_atomic_load(&partial[((int64_t)(k0 + u) * M + row) * m + o], __ATOMIC_RELAXED,
                                                                __HIP_MEMORY_SCOPE_AGENT)
                                            : 0.f;
                    }
#pragma unroll
        for (int u = 0; u < KBT; ++u)
#pragma unroll
            for (int r = 0; r < RT; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int t = 0; t < 2; ++t) s[r][i][t] += v[u][r][i][t];
    }
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int row = 16 * r + 4 * q + i, o = o0 + 16 * t + col;
                if (row < M && o < m) finish(s[r][i][t], row, o);
            }
    if (tid == 0) __hip_atomic_store(&counters[blockIdx.x], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // clean for the next call
}

// ---------------------------------------------------------------------------------------------------------------
// Decode kernel (M <= 32, out_features >= 128), one memory round trip.  At decode sizes the packed weight of a layer (8 MB at 4096 x 4096 x
// 4 bit) is smaller than the chip's bandwidth-latency product (8 TB/s x ~2 us = 16 MB): the kernel is bound by how many
// loads it keeps in flight, not by bandwidth.  So: one workgroup of 16 waves per 16 (NT = 1) or 32 (NT = 2) output
// features -- 256 workgroups at m = 4096, one per CU --, the waves split in_features among themselves, and every wave
// issues ALL the weight words and activations of up to KC groups of 32 columns before it touches any of them (for
// n = 4096: its whole share); the codebook table is built while they fly; the 16 partial tiles meet in LDS (fixed order:
// deterministic) and wave 0 writes y.  The sparse outliers of a layer (CSR by output feature, ganq_outlier_ratio) are
// added by the same launch: wave w takes the entries j = w (mod 16) of each feature into its accumulators before the
// reduction, so y = round(LUT part + sparse part + bias) with one rounding, like the two-launch path.
//
// SPLIT (layers with too few output features to give every CU a workgroup, e.g. 2048 x 8192): blockIdx.y splits
// in_features once more, ACROSS workgroups.  Wave 0 of each workgroup stores the workgroup's tile as fp32 (device-scope
// stores, like lut_mfma_kernel), takes a ticket on the feature block's counter, and the last one to arrive sums the
// partial tiles in split order (deterministic), adds addend / bias and writes y; it leaves the counter 0.
template <int BITS, int RT, bool BF16, int NT, int KC, bool SPLIT>
__global__ __launch_bounds__(1024, (RT == 1 && BITS != 3) ? 8 : 4) void lut_decode_kernel(const uint16_t* __restrict__ x, const uint32_t* __restrict__ qw,
                                                          const uint16_t* __restrict__ lut, const uint16_t* __restrict__ bias,
                                                          const float* __restrict__ addend, const int32_t* __restrict__ rowptr,
                                                          const int32_t* __restrict__ ocols, const uint16_t* __restrict__ ovals,
                                                          int M, int m, int n, int kb_per_wave, int kb_per_wg, int KS,
                                                          float* __restrict__ partial, int* __restrict__ counters,
                                                          uint16_t* __restrict__ y) {
    constexpr int V = 1 << BITS;
    constexpr int NW = 16;
    constexpr bool STRADDLE = (8 * BITS) % 16 != 0;  // 3-bit: a lane's 24 bits can span two words
    // tbl[t][e][lane]: codebook entry e of lane's feature in tile t, one dword slot per lane and entry (conflict-free).
    // (A table of PAIRS -- one lookup per two weights, no packing arithmetic -- was measured and lost: building 64 KB of it
    // per workgroup costs more than the shorter decode saves: 4096 x 4096 6.1 vs 5.7 us, 14336 x 4096 14.9 vs 14.1 us.)
    __shared__ __attribute__((aligned(4096))) uint32_t tbl[NT][V][64];  // 4 KB alignment: see the v_perm addressing
    __shared__ float red[NW - 1][NT][RT][4][64];

    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 15, q = lane >> 4;
    const int o0 = blockIdx.x * (16 * NT);
    const int nkb = n >> 5;
    const int ks = SPLIT ? (int)blockIdx.y : 0;
    const int wg_begin = SPLIT ? min(nkb, ks * kb_per_wg) : 0, wg_end = SPLIT ? min(nkb, wg_begin + kb_per_wg) : nkb;
    const int kb_begin = min(wg_end, wg_begin + wv * kb_per_wave), kb_end = min(wg_end, kb_begin + kb_per_wave);
    const int off = 8 * BITS * q, wi = off >> 5, sh = off & 31;
    const int wi2 = STRADDLE ? min(wi + 1, BITS - 1) : wi;
    // features are dealt to (tile, lane) interleaved: tile t of lane `col` is feature o0 + NT col + t, so that the NT words
    // a lane needs from one word row are adjacent in memory (one 8-byte load for NT = 2, 128-byte runs per 16 lanes)
    int oc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) oc[t] = min(o0 + NT * col + t, m - 1);
    // NT = 2 (the plan only picks it for even m): the pair's column, clamped so that a ragged last workgroup still loads
    // inside the matrix (its surplus lanes compute features they never write)
    const int cb = min(o0 + NT * col, m - NT);

    // sparse outliers: this wave's share of each feature's entries (issued first: the longest dependent chain)
    int o_beg[NT], o_end[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        o_beg[t] = o_end[t] = 0;
        if (rowptr && ks == 0) {  // the sparse part goes with the first split
            o_beg[t] = rowptr[oc[t]] + wv;
            o_end[t] = rowptr[oc[t] + 1];
        }
    }

    uint32_t wl[KC][NT], wh[KC][NT];
    u32x4 xa[KC][RT];
    auto issue = [&](int kb0) {
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            const int kb = kb0 + c;
            const bool ok = kb < kb_end;
            const int kbc = ok ? kb : min(kb_begin, nkb - 1);
            const int64_t row = (int64_t)(kbc * BITS + wi) * m, row2 = (int64_t)(kbc * BITS + wi2) * m;
            if constexpr (NT == 2) {
                const uint2 a = *reinterpret_cast<const uint2*>(qw + row + cb);
                wl[c][0] = a.x;
                wl[c][1] = a.y;
                if (STRADDLE) {
                    const uint2 b2 = *reinterpret_cast<const uint2*>(qw + row2 + cb);
                    wh[c][0] = b2.x;
                    wh[c][1] = b2.y;
                }
            } else {
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    wl[c][t] = qw[row + oc[t]];
                    if (STRADDLE) wh[c][t] = qw[row2 + oc[t]];
                }
            }
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                const int xr = 16 * r + col;
                u32x4 v = {0u, 0u, 0u, 0u};
                if (ok && xr < M) v = *reinterpret_cast<const u32x4*>(x + (int64_t)xr * n + 32 * kbc + 8 * q);
                xa[c][r] = v;
            }
        }
    };
    f32x4 acc[NT][RT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < RT; ++r) acc[t][r] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (kb_begin < kb_end) issue(kb_begin);  // everything this wave needs (n <= 32 * 16 * KC) is in flight from here on
    {   // wave t fetches the V entries of every lane's feature of tile t and fills that tile's table
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (wv == t) {
                const uint32_t* lp = reinterpret_cast<const uint32_t*>(lut + (int64_t)oc[t] * V);
                uint32_t h[V / 2];
#pragma unroll
                for (int e = 0; e < V / 2; ++e) h[e] = lp[e];
#pragma unroll
                for (int e = 0; e < V / 2; ++e) {
                    tbl[t][2 * e][lane] = h[e] & 0xffffu;
                    tbl[t][2 * e + 1][lane] = h[e] >> 16;
                }
            }
        }
    }
    __syncthreads();

    const uint32_t tb0 = (uint32_t)(uintptr_t)(&tbl[0][0][0]);
    const uint32_t lane4 = 4u * lane;
    auto consume = [&]() {
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            u32x4 b[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                uint32_t bits = wl[c][t] >> sh;
                if (STRADDLE) bits = (uint32_t)((((uint64_t)wh[c][t] << 32) | wl[c][t]) >> sh);
                const uint32_t tb = tb0 + (uint32_t)t * (V * 256u);
                if (BITS == 4) {
                    // byte k of lo/hi = entry index of element 2k / 2k+1, OR-ed with byte 1 of the (4 KB aligned) tile base:
                    // one v_perm per address ({byte1, byte0} = {index | base, 4*lane})
                    const uint32_t hib = ((tb >> 8) & 0xffu) * 0x01010101u;
                    const uint32_t lo = (bits & 0x0f0f0f0fu) | hib, hi = ((bits >> 4) & 0x0f0f0f0fu) | hib;
#pragma unroll
                    for (int p = 0; p < 4; ++p)
                        b[t][p] = lds_pair(__builtin_amdgcn_perm(lo, lane4, 0x0c0c0000u | ((4u + p) << 8)),
                                           __builtin_amdgcn_perm(hi, lane4, 0x0c0c0000u | ((4u + p) << 8)));
                } else {
                    const uint32_t base = tb + lane4;
#pragma unroll
                    for (int p = 0; p < 4; ++p)
                        b[t][p] = lds_pair((((bits >> (BITS * (2 * p))) & (V - 1)) << 8) + base,
                                           (((bits >> (BITS * (2 * p + 1))) & (V - 1)) << 8) + base);
                }
            }
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < RT; ++r) {
                    if (BF16)
                        acc[t][r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, xa[c][r]),
                                                                            __builtin_bit_cast(bf16x8, b[t]), acc[t][r], 0, 0, 0);
                    else
                        acc[t][r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, xa[c][r]),
                                                                           __builtin_bit_cast(f16x8, b[t]), acc[t][r], 0, 0, 0);
                }
        }
    };
    // groups past kb_end carry zero activations, so a ragged last round needs no special case
    for (int kb0 = kb_begin; kb0 < kb_end; kb0 += KC) {
        if (kb0 != kb_begin) issue(kb0);
        consume();
    }

    // sparse part: D[row][feature] += x[row][c_j] * v_j over this wave's entries of the feature (fp32)
    if (rowptr && ks == 0) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
            for (int j = o_beg[t]; j < o_end[t]; j += NW) {
                const int c = ocols[j];
                const uint16_t vb = ovals[j];
                const float v = BF16 ? __builtin_bit_cast(float, (uint32_t)vb << 16) : (float)__builtin_bit_cast(_Float16, vb);
#pragma unroll
                for (int r = 0; r < RT; ++r)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int row = 16 * r + 4 * q + i;
                        if (row < M && o0 + NT * col + t < m) {
                            const uint16_t xb = x[(int64_t)row * n + c];
                            const float xv = BF16 ? __builtin_bit_cast(float, (uint32_t)xb << 16) : (float)__builtin_bit_cast(_Float16, xb);
                            acc[t][r][i] = fmaf(xv, v, acc[t][r][i]);
                        }
                    }
            }
    }

    if (wv > 0) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < RT; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) red[wv - 1][t][r][i][lane] = acc[t][r][i];
    }
    __syncthreads();
    if (wv != 0) return;
    auto finish = [&](float v, int row, int o) {
        if (addend) v += addend[(int64_t)row * m + o];
        if (bias) v += BF16 ? __builtin_bit_cast(float, (uint32_t)bias[o] << 16) : (float)__builtin_bit_cast(_Float16, bias[o]);
        y[(int64_t)row * m + o] = BF16 ? __builtin_bit_cast(uint16_t, (__bf16)v) : __builtin_bit_cast(uint16_t, (_Float16)v);
    };
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float v = acc[t][r][i];
#pragma unroll
                for (int w2 = 0; w2 < NW - 1; ++w2) v += red[w2][t][r][i][lane];  // fixed order
                const int row = 16 * r + 4 * q + i, o = o0 + NT * col + t;
                if (row < M && o < m) {
                    if constexpr (SPLIT)
                        __hip_atomic_store(&partial[((int64_t)ks * M + row) * m + o], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    else
                        finish(v, row, o);
                }
            }
    if constexpr (SPLIT) {
        // only wave 0 is left: its device-scope stores are visible to every XCD once vmcnt reaches 0 (no cache walk),
        // then the ticket; the last workgroup of the feature block sums the KS tiles in split order
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        int ticket = 0;
        if (lane == 0) ticket = __hip_atomic_fetch_add(&counters[blockIdx.x], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ticket = __builtin_amdgcn_readfirstlane(ticket);
        if (ticket != KS - 1) return;
        float s[NT][RT][4];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < RT; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) s[t][r][i] = 0.f;
        for (int k2 = 0; k2 < KS; ++k2) {
            float v[NT][RT][4];
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < RT; ++r)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int row = 16 * r + 4 * q + i, o = o0 + NT * col + t;
                        v[t][r][i] = (row < M && o < m) ? __hip_atomic_load(&partial[((int64_t)k2 * M + row) * m + o], __ATOMIC_RELAXED,
                                                                            __HIP_MEMORY_SCOPE_AGENT)
                                                        : 0.f;
                    }
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < RT; ++r)
#pragma unroll
                    for (int i = 0; i < 4; ++i) s[t][r][i] += v[t][r][i];
        }
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < RT; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = 16 * r + 4 * q + i, o = o0 + NT * col + t;
                    if (row < M && o < m) finish(s[t][r][i], row, o);
                }
        if (lane == 0) __hip_atomic_store(&counters[blockIdx.x], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // clean for the next call
    }
}

// The dense prefill path (gemm_h16.hip) dequantises the layer once per call.  64 output features per workgroup, their codebooks
// in LDS as tbl[entry][feature] (the 64 lanes of a wave = 64 features: conflict-free); a wave takes 64 columns (two 32-column
// groups) of its 64 features per step: coalesced word loads (lane = feature), 64 lookups, and the 128 bytes of every feature go
// through an LDS tile (16-byte slots XOR-swizzled with (feature >> 1) & 7: conflict-free both ways) so that the global stores are
// WHOLE 128-byte lines -- 8 lanes per feature, 8 features per instruction.  (Round 4's first version stored 16 bytes per lane
// at a stride of one row, 64 lines per instruction: 17 us per 4096 x 4096 against 8 now.)  An odd last group of 32 columns is
// stored the slow way.
template <int BITS>
__global__ __launch_bounds__(256) void lut_dequant_rows_kernel(const uint32_t* __restrict__ qw, const uint16_t* __restrict__ lut,
                                                               int m, int n, uint16_t* __restrict__ Wq) {
    constexpr int V = 1 << BITS;
    __shared__ uint16_t tbl[V][64];
    __shared__ __align__(16) char tile[4][64 * 128];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int o0 = blockIdx.x * 64;
    for (int i = tid; i < 64 * V; i += 256) {
        const int f = i / V, e = i % V;
        tbl[e][f] = (o0 + f < m) ? lut[(int64_t)(o0 + f) * V + e] : (uint16_t)0;
    }
    __syncthreads();
    const int o = min(o0 + lane, m - 1);  // (lanes past the last feature decode a copy; they store nothing)
    const int npairs = n >> 6;
    char* my = tile[wv];
    const int sw = (lane >> 1) & 7;
    for (int gp = blockIdx.y * 4 + wv; gp < npairs; gp += gridDim.y * 4) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int g = 2 * gp + half;
            uint32_t w[BITS];
#pragma unroll
            for (int b = 0; b < BITS; ++b) w[b] = qw[(int64_t)(g * BITS + b) * m + o];
#pragma unroll
            for (int oc = 0; oc < 4; ++oc) {
                uint32_t d[4];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    d[j] = (uint32_t)tbl[extract<BITS>(w, 8 * oc + 2 * j)][lane] | ((uint32_t)tbl[extract<BITS>(w, 8 * oc + 2 * j + 1)][lane] << 16);
                *reinterpret_cast<uint4*>(my + lane * 128 + (((4 * half + oc) ^ sw) << 4)) = make_uint4(d[0], d[1], d[2], d[3]);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int f = 8 * i + (lane >> 3), slot = lane & 7;
            const uint4 v = *reinterpret_cast<const uint4*>(my + f * 128 + ((slot ^ ((f >> 1) & 7)) << 4));
            if (o0 + f < m) *reinterpret_cast<uint4*>(Wq + (int64_t)(o0 + f) * n + 64 * gp + 8 * slot) = v;
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if ((n & 63) != 0 && blockIdx.y == 0 && wv == 0 && o0 + lane < m) {  // the odd last group of 32 columns
        const int g = (n >> 5) - 1;
        uint32_t w[BITS];
#pragma unroll
        for (int b = 0; b < BITS; ++b) w[b] = qw[(int64_t)(g * BITS + b) * m + o];
        uint16_t* out = Wq + (int64_t)o * n + 32 * g;
#pragma unroll
        for (int oc = 0; oc < 4; ++oc) {
            uint32_t d[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                d[j] = (uint32_t)tbl[extract<BITS>(w, 8 * oc + 2 * j)][lane] | ((uint32_t)tbl[extract<BITS>(w, 8 * oc + 2 * j + 1)][lane] << 16);
            *reinterpret_cast<uint4*>(out + 8 * oc) = make_uint4(d[0], d[1], d[2], d[3]);
        }
    }
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

// The ticket counters occupy a FIXED region at the head of the workspace (one int per block of output features, up to
// 16384 blocks), the partial tiles start behind it: one workspace serves layers of any shape in turn, and the floats
// a smaller layer left in the partial-tile region can never be read as counters by a larger one.
constexpr size_t LUT_COUNTER_BYTES = 64 * 1024;

struct LutPlan {
    bool inwg;  // decode kernel: the 16 waves of a workgroup split in_features among themselves (no exchange through memory)
    int nt;     // decode kernel: 16-feature tiles per workgroup (1: 256 workgroups at m = 4096; 2 from m = 8192)
    int ob, KS, kb_per_wg, kb_per_wave = 0;
    size_t counter_bytes, bytes;
};
static LutPlan lut_plan(int64_t M, int64_t m, int64_t n, int bits) {
    // split in_features over blockIdx.y until the launch has enough workgroups for 256 CUs; fewer splits for taller x
    // (the partial tiles are KS*M*m floats and the last workgroup of a feature block sums KS of them)
    LutPlan p;
    const int nkb = (int)(n >> 5);
    const int inwg_env = (int)opt_get(OPT_LUT_INWG);
    // enough workgroups to occupy the chip, and a reduction buffer that fits LDS (two row tiles)
    // (from 128 output features: 512 x 2048 takes 3.9 us with the decode kernel against 7.1 with the split-K kernel, 768 x 768
    // 3.6 against 6.3 -- the latter slower than fp16 F.linear)
    p.inwg = inwg_env >= 0 ? (inwg_env != 0 && M <= 32) : (M <= 32 && m >= 128);
    p.nt = 1;
    if (p.inwg) {
        const int nt_opt = (int)opt_get(OPT_LUT_NT);
        // 32 features per workgroup once that still gives every CU a workgroup (m >= 8192), or for more than one row of x
        // when it gives at least half of them one (every workgroup reads all of x: twice the features halve that traffic)
        p.nt = nt_opt > 0 ? (nt_opt >= 2 ? 2 : 1) : ((m + 31) / 32 >= 256 ? 2 : 1);
        if ((m & 1) || M > 16) p.nt = 1;  // the pair loads of the 32-feature variant want an even row pitch; LDS (see launch)
        p.ob = (int)((m + 16 * p.nt - 1) / (16 * p.nt));
        // fewer feature blocks than CUs: split in_features across workgroups as well, as long as every wave keeps at least
        // eight groups of 32 columns -- the exchange of the partial tiles costs ~2 us, so it only pays on long rows
        // (measured cold, 2048 x 8192: M = 1 7.9 -> 7.4 us, M = 16 12.9 -> 11.0 us with 2 splits, 9.1 / 13.7 us with 4;
        // 2048 x 2048 and 1024 x 4096 lose with any split).  GANQ_LUT_KS forces the factor.
        const int ks_opt = (int)opt_get(OPT_LUT_KS);
        int ks = ks_opt > 0 ? ks_opt : std::min(8, 256 / std::max(1, p.ob));
        while (ks_opt <= 0 && ks > 1 && nkb < 8 * LWK * ks) ks >>= 1;
        ks = std::max(1, std::min(ks, nkb));
        p.kb_per_wg = (nkb + ks - 1) / ks;
        p.KS = (nkb + p.kb_per_wg - 1) / p.kb_per_wg;
        p.kb_per_wave = (p.kb_per_wg + LWK - 1) / LWK;
        p.counter_bytes = LUT_COUNTER_BYTES;
        p.bytes = p.counter_bytes + (p.KS > 1 ? align_up((size_t)p.KS * (size_t)M * (size_t)m * sizeof(float), 256) : 0);
        return p;
    }
    p.ob = (int)((m + LUT_FB - 1) / LUT_FB);
    const int target_env = (int)opt_get(OPT_LUT_WGS);
    // measured on MI355X (tools/lut_trace.sh): ~512 workgroups, and at most 8 / 4 / 2 splits for M <= 16 / 32 / 64 --
    // the exchange of partial tiles goes through memory (device-scope accesses) and its cost grows with KS * M
    const int cap_env = (int)opt_get(OPT_LUT_KS);
    const int target = target_env > 0 ? target_env : 512;
    const int cap = cap_env > 0 ? cap_env : (M <= 16 ? 8 : (M <= 32 ? 4 : 2));
    int ks = std::max(1, std::min(std::min(nkb / 2, cap), (target + p.ob - 1) / p.ob));
    (void)bits;
    p.kb_per_wg = (nkb + ks - 1) / ks;
    p.KS = (nkb + p.kb_per_wg - 1) / p.kb_per_wg;
    p.counter_bytes = LUT_COUNTER_BYTES;
    p.bytes = p.counter_bytes + (p.KS > 1 ? align_up((size_t)p.KS * (size_t)M * (size_t)m * sizeof(float), 256) : 0);
    return p;
}

// lut_gemm.hip: the fused LUT-dequant GEMM that serves M > LUT_MAX_M (prefill)
int lut_gemm(const void* x, const uint32_t* qw, const void* lut, const void* bias, const float* addend, int dtype, int64_t M,
             int64_t m, int64_t n, int bits, void* y, float* partial, size_t partial_bytes, hipStream_t stream);
size_t lut_gemm_workspace_bytes(int64_t M, int64_t m, int64_t n);

// Prefill with many rows: dequantise the layer ONCE into the workspace and run the dense GEMM (gemm_h16.hip) -- every 256-row
// tile of the fused kernel re-decodes the same weights.  Measured (MI355X, 4-bit fp16, round 4): 4096 x 4096 M = 4096 fused
// 166 us, dense 116 + dequant; M = 2048 a tie; below, the dense tiles no longer fill the chip and the fused kernel's split of
// in_features wins.  From >= 1024 rows on, when 256-row tiles give (nearly) every CU one.
static bool lut_dense_path(int64_t M, int64_t m, int64_t n) {
    const long long thr = opt_get(OPT_LUT_DENSE_M);
    if (thr == 0 || !gemm_h16_supported(M, m, n) || (n & 31) != 0) return false;
    if (thr > 0) return M >= thr;
    const int ncu = std::max(1, current_device_cus());
    const int64_t t128 = ((M + 127) / 128) * ((m + 255) / 256);
    // the dense kernel's smaller tile (128 x 256) fills the chip at least once.  (4096 x 4096 at M = 2048, 256 such tiles: 73 + 12 us
    // against 94 for the fused kernel since the dense kernel walks its tiles in L2-sized blocks; at M = 1024, 128 tiles: 63 + 12
    // against 72 the other way round.)
    return M >= 1024 && t128 >= ncu;
}

}  // namespace ganq

using namespace ganq;

extern "C" size_t ganq_lut_linear_workspace_bytes(int64_t M, int64_t m, int64_t n, int bits) {
    if (M <= 0 || m <= 0 || n < 32) return 0;
    if (M > LUT_MAX_M) {  // counters + split-K partial tiles of the fused kernel, or the dequantised layer of the dense path
        if (lut_dense_path(M, m, n)) return LUT_COUNTER_BYTES + align_up((size_t)m * (size_t)n * 2, 256);
        return LUT_COUNTER_BYTES + lut_gemm_workspace_bytes(M, m, n);
    }
    return lut_plan(M, m, n, bits).bytes;
}

extern "C" int ganq_lut_linear_workspace_init(void* workspace, size_t workspace_bytes, void* stream_) {
    if (workspace_bytes == 0) return 0;
    if (!workspace) return fail(-3, "ganq_lut_linear_workspace_init: null workspace");
    GANQ_HIP_CHECK(hipMemsetAsync(workspace, 0, workspace_bytes, static_cast<hipStream_t>(stream_)));
    return 0;
}

struct LutCsr {  // sparse outliers of the layer (device pointers; rowptr == nullptr: none)
    const int32_t* rowptr = nullptr;
    const int32_t* cols = nullptr;
    const uint16_t* vals = nullptr;
};

template <int BITS, int RT, bool BF16, int NT>
static void launch_decode(const uint16_t* xp, const uint32_t* qw, const uint16_t* lp, const uint16_t* bp, const float* addend,
                          const LutCsr& csr, int M, int m, int n, const LutPlan& p, float* partial, int* counters, uint16_t* yp,
                          hipStream_t stream) {
    // groups of 32 columns a wave has in flight.  16 features per workgroup: 8 (n = 4096: all of its share).  32 features:
    // 4, which fits 64 registers, so two workgroups share a CU and the 448 workgroups of m = 14336 are resident in one
    // generation (measured cold, M = 1, 14336 x 4096: 14.6 -> 10.9 us; a double-buffered 4 + 4 loop for 16 features was
    // slower at 4096 x 4096, 6.6 vs 5.7 us)
    constexpr int KC = (RT == 1 && NT == 1) ? 8 : 4;
    if (p.KS > 1)
        hipLaunchKernelGGL((lut_decode_kernel<BITS, RT, BF16, NT, KC, true>), dim3((unsigned)p.ob, (unsigned)p.KS), dim3(1024), 0, stream,
                           xp, qw, lp, bp, addend, csr.rowptr, csr.cols, csr.vals, M, m, n, p.kb_per_wave, p.kb_per_wg, p.KS, partial,
                           counters, yp);
    else
        hipLaunchKernelGGL((lut_decode_kernel<BITS, RT, BF16, NT, KC, false>), dim3((unsigned)p.ob), dim3(1024), 0, stream, xp, qw, lp, bp,
                           addend, csr.rowptr, csr.cols, csr.vals, M, m, n, p.kb_per_wave, p.kb_per_wg, 1, partial, counters, yp);
}

template <int BITS, int RT>
static int launch_lut_rt(const void* x, const uint32_t* qw, const void* lut, const void* bias, const float* addend, const LutCsr& csr,
                         int dtype, int M, int m, int n,
                         const LutPlan& p, float* partial, int* counters, void* y, hipStream_t stream) {
    const dim3 grid((unsigned)p.ob, (unsigned)p.KS);
    const uint16_t* xp = static_cast<const uint16_t*>(x);
    const uint16_t* lp = static_cast<const uint16_t*>(lut);
    const uint16_t* bp = static_cast<const uint16_t*>(bias);
    uint16_t* yp = static_cast<uint16_t*>(y);
    if constexpr (RT <= 2) {
        if (p.inwg) {
            // 32 features per workgroup only with one row tile (the pair tables of two feature tiles plus the reduction
            // buffer of two row tiles do not fit the LDS); the plan knows
            if constexpr (RT == 1) {
                if (p.nt == 2) {
                    if (dtype == 1) launch_decode<BITS, RT, true, 2>(xp, qw, lp, bp, addend, csr, M, m, n, p, partial, counters, yp, stream);
                    else launch_decode<BITS, RT, false, 2>(xp, qw, lp, bp, addend, csr, M, m, n, p, partial, counters, yp, stream);
                    GANQ_LAUNCH_CHECK();
                    return 0;
                }
            }
            if (dtype == 1) launch_decode<BITS, RT, true, 1>(xp, qw, lp, bp, addend, csr, M, m, n, p, partial, counters, yp, stream);
            else launch_decode<BITS, RT, false, 1>(xp, qw, lp, bp, addend, csr, M, m, n, p, partial, counters, yp, stream);
            GANQ_LAUNCH_CHECK();
            return 0;
        }
    }
    if (csr.rowptr) return fail(-2, "ganq_lut_linear_fwd: the fused outlier path serves the decode kernel only");
    if (dtype == 1)
        hipLaunchKernelGGL((lut_mfma_kernel<BITS, RT, true, LW, false>), grid, dim3(LW * 64), 0, stream, xp, qw, lp, bp, addend, M, m, n,
                           p.kb_per_wg, p.KS, partial, counters, yp);
    else
        hipLaunchKernelGGL((lut_mfma_kernel<BITS, RT, false, LW, false>), grid, dim3(LW * 64), 0, stream, xp, qw, lp, bp, addend, M, m, n,
                           p.kb_per_wg, p.KS, partial, counters, yp);
    GANQ_LAUNCH_CHECK();
    return 0;
}

template <int BITS>
static int launch_lut(const void* x, const uint32_t* qw, const void* lut, const void* bias, const float* addend, const LutCsr& csr,
                      int dtype, int M, int m, int n,
                      const LutPlan& p, float* partial, int* counters, void* y, hipStream_t stream) {
    if (M <= 16) return launch_lut_rt<BITS, 1>(x, qw, lut, bias, addend, csr, dtype, M, m, n, p, partial, counters, y, stream);
    if (M <= 32) return launch_lut_rt<BITS, 2>(x, qw, lut, bias, addend, csr, dtype, M, m, n, p, partial, counters, y, stream);
    return launch_lut_rt<BITS, 4>(x, qw, lut, bias, addend, csr, dtype, M, m, n, p, partial, counters, y, stream);
}

static int check_lut_args(const char* who, int dtype, int64_t m, int64_t n, int bits) {
    if (dtype != 0 && dtype != 1) return fail(-2, "%s: dtype %d (0 = fp16, 1 = bf16)", who, dtype);
    if (bits != 2 && bits != 3 && bits != 4) return fail(-2, "%s: bits=%d not supported (2, 3, 4 are)", who, bits);
    if (n % 32 != 0) return fail(-2, "%s: in_features=%lld must be a multiple of 32", who, (long long)n);
    if (m > INT32_MAX / 2 || n > INT32_MAX / 2) return fail(-1, "%s: shape too large", who);
    return 0;
}

static int lut_linear_fwd(const void* x, const int32_t* qweight, const void* lut, const void* bias, const float* addend, int dtype,
                          int64_t M, int64_t m, int64_t n, int bits, void* y, void* workspace, size_t workspace_bytes,
                          void* stream_, const LutCsr& csr = LutCsr()) {
    if (M < 0 || m < 0 || n < 0) return fail(-1, "ganq_lut_linear_fwd: negative shape");
    if (M == 0 || m == 0) return 0;
    int rc = check_lut_args("ganq_lut_linear_fwd", dtype, m, n, bits);
    if (rc) return rc;
    if ((reinterpret_cast<uintptr_t>(x) & 15) != 0) return fail(-2, "ganq_lut_linear_fwd: x must be 16-byte aligned");
    if ((reinterpret_cast<uintptr_t>(lut) & 3) != 0) return fail(-2, "ganq_lut_linear_fwd: lut must be 4-byte aligned");
    if (!x || !qweight || !lut || !y) return fail(-3, "ganq_lut_linear_fwd: null pointer");
    if (M > LUT_MAX_M) {  // prefill: the fused LUT-dequant GEMM (lut_gemm.hip); the weight is never materialised
        if (csr.rowptr)
            return fail(-2, "ganq_lut_linear_fwd: outliers fused into the launch serve M <= %d; for larger M pass their product as "
                            "the fp32 addend (ganq_outlier_matmul + ganq_lut_linear_fwd_add)", LUT_MAX_M);
        hipStream_t gstream = static_cast<hipStream_t>(stream_);
        if (!workspace || workspace_bytes < LUT_COUNTER_BYTES)
            return fail(-4, "ganq_lut_linear_fwd: workspace %zu B < required %zu B", workspace_bytes, LUT_COUNTER_BYTES);
        ProfScope gprof(KID_LUT_GEMM, gstream);
        if (lut_dense_path(M, m, n)) {
            const size_t need = LUT_COUNTER_BYTES + align_up((size_t)m * (size_t)n * 2, 256);
            if (workspace_bytes < need) return fail(-4, "ganq_lut_linear_fwd: workspace %zu B < required %zu B", workspace_bytes, need);
            uint16_t* Wd = reinterpret_cast<uint16_t*>(static_cast<char*>(workspace) + LUT_COUNTER_BYTES);
            const uint32_t* qwp = reinterpret_cast<const uint32_t*>(qweight);
            const uint16_t* lp = static_cast<const uint16_t*>(lut);
            const int npairs = (int)(n >> 6);
            const int gx = (int)((m + 63) / 64);
            const int gy = std::max(1, std::min((npairs + 3) / 4, std::max(1, 2048 / gx)));
            const dim3 grid((unsigned)gx, (unsigned)gy);
            if (bits == 2) hipLaunchKernelGGL(lut_dequant_rows_kernel<2>, grid, dim3(256), 0, gstream, qwp, lp, (int)m, (int)n, Wd);
            else if (bits == 3) hipLaunchKernelGGL(lut_dequant_rows_kernel<3>, grid, dim3(256), 0, gstream, qwp, lp, (int)m, (int)n, Wd);
            else hipLaunchKernelGGL(lut_dequant_rows_kernel<4>, grid, dim3(256), 0, gstream, qwp, lp, (int)m, (int)n, Wd);
            GANQ_LAUNCH_CHECK();
            return gemm_h16(x, Wd, bias, addend, dtype, M, m, n, y, gstream);
        }
        return lut_gemm(x, reinterpret_cast<const uint32_t*>(qweight), lut, bias, addend, dtype, M, m, n, bits, y,
                        reinterpret_cast<float*>(static_cast<char*>(workspace) + LUT_COUNTER_BYTES), workspace_bytes - LUT_COUNTER_BYTES,
                        gstream);
    }
    const LutPlan p = lut_plan(M, m, n, bits);
    if ((size_t)p.ob * sizeof(int) > LUT_COUNTER_BYTES)
        return fail(-2, "ganq_lut_linear_fwd: out_features=%lld needs more than %zu ticket counters", (long long)m,
                    LUT_COUNTER_BYTES / sizeof(int));
    if (!workspace || workspace_bytes < p.bytes)
        return fail(-4, "ganq_lut_linear_fwd: workspace %zu B < required %zu B", workspace_bytes, p.bytes);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    int* counters = static_cast<int*>(workspace);
    float* partial = reinterpret_cast<float*>(static_cast<char*>(workspace) + p.counter_bytes);
    const uint32_t* qw = reinterpret_cast<const uint32_t*>(qweight);
    ProfScope prof(KID_LUT_GEMV, stream);
    if (bits == 2) rc = launch_lut<2>(x, qw, lut, bias, addend, csr, dtype, (int)M, (int)m, (int)n, p, partial, counters, y, stream);
    else if (bits == 3) rc = launch_lut<3>(x, qw, lut, bias, addend, csr, dtype, (int)M, (int)m, (int)n, p, partial, counters, y, stream);
    else rc = launch_lut<4>(x, qw, lut, bias, addend, csr, dtype, (int)M, (int)m, (int)n, p, partial, counters, y, stream);
    return rc;
}

extern "C" int ganq_lut_linear_fwd(const void* x, const int32_t* qweight, const void* lut, const void* bias, int dtype,
                                   int64_t M, int64_t m, int64_t n, int bits, void* y, void* workspace,
                                   size_t workspace_bytes, void* stream_) {
    return lut_linear_fwd(x, qweight, lut, bias, nullptr, dtype, M, m, n, bits, y, workspace, workspace_bytes, stream_);
}

// y = x @ dequant^T + addend + bias with addend fp32 [M, m] added before the one rounding to the activation dtype
extern "C" int ganq_lut_linear_fwd_add(const void* x, const int32_t* qweight, const void* lut, const void* bias,
                                       const float* addend, int dtype, int64_t M, int64_t m, int64_t n, int bits, void* y,
                                       void* workspace, size_t workspace_bytes, void* stream_) {
    return lut_linear_fwd(x, qweight, lut, bias, addend, dtype, M, m, n, bits, y, workspace, workspace_bytes, stream_);
}

// LUT forward of a layer with sparse outliers in one call: the sparse product goes to the tail of the workspace as fp32
// and is added by the LUT kernel before its rounding (two launches, no host work in between)
extern "C" int ganq_outlier_matmul(const void* x, int dtype, int64_t M, int64_t m, int64_t n, const int32_t* rowptr,
                                   const int32_t* cols, const void* vals, float* out, void* stream_);

extern "C" size_t ganq_lut_linear_outliers_workspace_bytes(int64_t M, int64_t m, int64_t n, int bits) {
    if (M <= 0 || m <= 0) return 0;
    return align_up(ganq_lut_linear_workspace_bytes(M, m, n, bits), 256) + (size_t)M * m * sizeof(float);
}

extern "C" int ganq_lut_linear_fwd_outliers(const void* x, const int32_t* qweight, const void* lut, const void* bias,
                                            const int32_t* rowptr, const int32_t* cols, const void* vals, int dtype, int64_t M,
                                            int64_t m, int64_t n, int bits, void* y, void* workspace, size_t workspace_bytes,
                                            void* stream_) {
    if (M < 0 || m < 0 || n < 0) return fail(-1, "ganq_lut_linear_fwd_outliers: negative shape");
    if (M == 0 || m == 0) return 0;
    const size_t base = align_up(ganq_lut_linear_workspace_bytes(M, m, n, bits), 256);
    if (!workspace || workspace_bytes < base + (size_t)M * m * sizeof(float))
        return fail(-4, "ganq_lut_linear_fwd_outliers: workspace %zu B < required %zu B", workspace_bytes,
                    base + (size_t)M * m * sizeof(float));
    if (n >= 32 && n % 32 == 0 && lut_plan(M, m, n, bits).inwg) {
        // decode sizes: ONE launch -- the sparse entries are added inside the LUT kernel (lut_decode_kernel)
        if (!rowptr) return fail(-3, "ganq_lut_linear_fwd_outliers: null rowptr");
        LutCsr csr;
        csr.rowptr = rowptr;
        csr.cols = cols;
        csr.vals = static_cast<const uint16_t*>(vals);
        return lut_linear_fwd(x, qweight, lut, bias, nullptr, dtype, M, m, n, bits, y, workspace, base, stream_, csr);
    }
    float* addend = reinterpret_cast<float*>(static_cast<char*>(workspace) + base);
    int rc = ganq_outlier_matmul(x, dtype, M, m, n, rowptr, cols, vals, addend, stream_);
    if (rc) return rc;
    return lut_linear_fwd(x, qweight, lut, bias, addend, dtype, M, m, n, bits, y, workspace, base, stream_);
}

extern "C" int ganq_lut_dequant(const int32_t* qweight, const void* lut, int dtype, int64_t m, int64_t n, int bits,
                                void* Wq_out, void* stream_) {
    if (m < 0 || n < 0) return fail(-1, "ganq_lut_dequant: negative shape");
    if (m == 0 || n == 0) return 0;
    int rc = check_lut_args("ganq_lut_dequant", dtype, m, n, bits);
    if (rc) return rc;
    if (!qweight || !lut || !Wq_out) return fail(-3, "ganq_lut_dequant: null pointer");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const uint32_t* qw = reinterpret_cast<const uint32_t*>(qweight);
    uint16_t* out = static_cast<uint16_t*>(Wq_out);
    ProfScope prof(KID_LUT_GEMM, stream);
    {
        const uint16_t* lp = static_cast<const uint16_t*>(lut);
        const int npairs = (int)(n >> 6);
        const int gx = (int)((m + 63) / 64);
        const int gy = std::max(1, std::min((npairs + 3) / 4, std::max(1, 2048 / gx)));
        const dim3 rgrid((unsigned)gx, (unsigned)gy);
        if (bits == 2) hipLaunchKernelGGL(lut_dequant_rows_kernel<2>, rgrid, dim3(256), 0, stream, qw, lp, (int)m, (int)n, out);
        else if (bits == 3) hipLaunchKernelGGL(lut_dequant_rows_kernel<3>, rgrid, dim3(256), 0, stream, qw, lp, (int)m, (int)n, out);
        else hipLaunchKernelGGL(lut_dequant_rows_kernel<4>, rgrid, dim3(256), 0, stream, qw, lp, (int)m, (int)n, out);
    }
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

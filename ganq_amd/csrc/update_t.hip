// T-update: closed-form codebook update of GANQ (reference ganq.py:570-591, CPU/gelsd branch)
//     A_i = S_i H S_i^T,  b_i = S_i (W H)_i^T,  T_i = minimum-norm lstsq(A_i, b_i)
// and, from the same quantities, the loss of ganq.py:392-395 / :621-622
//     dist = sum_i (w_i - S_i^T t_i)^T H (w_i - S_i^T t_i) = sum_i ( w_i^T H w_i - 2 t_i^T b_i + t_i^T A_i t_i )
// without ever materialising the one-hot tensor S [m,V,n] (1 GiB at 4096^2, ganq.py:505).
//
// A_i[a][b] = sum_{u,v} [Q_iu == a][Q_iv == b] H[u,v] is a bucket sum of ALL of H per row: m*n^2 additions.
// It runs on the integer matrix cores, exactly:
//   * once per layer H is converted to fixed point, H ~= scale * (I + J / 65536) with scale = max|H| / 2^30: I is a
//     31-bit integer split into 4 balanced base-256 digits (int8 planes 0..3 of Hq[p][v][u]); J is a 16-bit
//     EXTENSION word (2 more digits, planes 4..5) that exists only when the dynamic range of H needs it -- decided on
//     the device, max|H| > 16 mean(diag H): with massive-activation features (H_ii 10^4..10^6 x the typical one) a
//     31-bit word leaves the typical entry 10-17 significant bits, against the 24 of the fp32 entries the reference
//     sums (ganq.py:589-591); 47 bits keep every entry of a matrix spanning 2^23 to full fp32 precision.  The bucket
//     sums of I and J are kept as two exact int64 words and meet in fp64 in the per-row solve (t_prepare);
//   * per iteration the indices become bit masks bits[row][u/64][code] (one ballot per code);
//   * v_mfma_i32_16x16x64_i8 multiplies the one-hot matrix [16 codes x 64 u] (expanded from 16 mask bits per lane
//     with one integer multiply per 4 bytes) with a digit tile [64 u x 16 v]: int32 accumulators hold
//     Y_p[a][v] = sum_{u in a, u > v} digit_p(H[v][u]).  Integer sums are exact and order-independent, so the
//     result is deterministic whatever the scheduling;
//   * the v index is bucketed with 64-bit LDS atomics (digits recombined to one int64 first), again exact;
//   * H symmetric: only u > v is visited, A = M + M^T + diag.
// The per-row 16x16 systems are then solved in fp64 (round-robin Jacobi, gelsd cut-off); with A exact and
// b = S (W H) accumulated in fp64 the closed-form loss has no cancellation problem, so no (W-Wq)@H product is
// needed per iteration.
#include <climits>
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "update_t.h"
#include "wh_gemm.h"

namespace ganq {

// tile shape of onehot_accum_kernel (overridable for experiments).  Measured on MI355X, 4096x4096: (waves, u per tile) =
// (8, 256) 2.25 ms, (8, 128) 2.31, (4, 256) 3.44, (4, 128) 3.43 -- two smaller workgroups per CU lose more to the halved
// reuse of each staged digit tile than they gain from overlapping each other's barriers
#ifndef ACC_TW
#define ACC_TW 8
#endif
// ACC_VH = 32-column halves per chunk.  2: every expanded one-hot fragment (the kernel's vector work: 14 instructions per
// fragment) serves 8 MFMAs instead of 4 -- two rows per wave instead of four keep the accumulators at 128 registers, and
// 128-wide u tiles keep the two staged digit tiles of 64 columns inside the LDS.  Measured on MI355X, 4096x4096: 2.58 ms
// against 2.22 ms for the default (half the rows per workgroup stream every digit tile twice as often): the fragment
// expansion is not what bounds the kernel.
#ifndef ACC_VH
#define ACC_VH 1
#endif
#ifndef ACC_UT
#define ACC_UT (ACC_VH == 2 ? 128 : 256)
#endif
// ACC_PIN_B = 1 keeps the LDS reads of the next 32-column half in front of this half's matrix instructions (a scheduling barrier:
// the B fragments are then really double-buffered instead of re-using their registers right before use).  Measured: 2.14 vs 2.08 ms.
#ifndef ACC_PIN_B
#define ACC_PIN_B 0
#endif
constexpr int TW = ACC_TW;      // waves per workgroup
constexpr int VH = ACC_VH;
constexpr int RW = VH == 2 ? 2 : 4;  // rows per wave
constexpr int TR = TW * RW;     // rows of W per workgroup
constexpr int VCH = 32 * VH;    // v columns per chunk (32 per MFMA column tile)
constexpr int UT = ACC_UT;      // u per staged tile (UT / 64 MFMA steps of 64)
constexpr int KS64 = UT / 64;
static_assert(KS64 >= 2, "the code masks are requested KS64 - 1 steps ahead");
constexpr int UC16 = UT / 16;   // 16-byte pieces per tile row
constexpr int NP = 8;           // chunk c belongs to part c % NP; every part writes one partial A
constexpr int BROW = UT + 16;   // LDS row pitch of a digit tile in bytes (pad against bank conflicts)
constexpr int btile_bytes(int npl) { return npl * VCH * BROW; }

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef double f64x4 __attribute__((ext_vector_type(4)));

struct TPrep {              // device-side header written by t_prepare
    double scale;           // H ~= scale * (I + J / 65536)
    unsigned int absmax_bits;
    int ext;                // 1: the 16-bit extension word J is in use (planes 4..5, Jint, the *_lo bucket sums)
};
constexpr double EXT_UNIT = 1.0 / 65536.0;
// two balanced base-256 digits hold J in [-32896, 32639]: the int16 copy stores J + JBIAS, which is exactly [-32768, 32767]
constexpr int JBIAS = 128;

// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ H, int64_t total, TPrep* __restrict__ prep) {
    __shared__ unsigned int wmax[4];
    unsigned int mx = 0;
    const int64_t n4 = (reinterpret_cast<uintptr_t>(H) & 15) == 0 ? (total >> 2) : 0;  // 16-byte loads when aligned
    const uint4* H4 = reinterpret_cast<const uint4*>(H);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const uint4 v = H4[i];  // |x| bits order like the values
        mx = max(max(mx, v.x & 0x7fffffffu), max(max(v.y & 0x7fffffffu, v.z & 0x7fffffffu), v.w & 0x7fffffffu));
    }
    for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256)
        mx = max(mx, __builtin_bit_cast(unsigned int, H[i]) & 0x7fffffffu);
    for (int off = 32; off > 0; off >>= 1) mx = max(mx, (unsigned int)__shfl_xor((int)mx, off));
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = mx;
    __syncthreads();
    // one atomic per workgroup: thousands of same-address atomics were most of this kernel's time
    if (threadIdx.x == 0) atomicMax(&prep->absmax_bits, max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3])));
}

// scale = max|H| / 2^30; the extension word is switched on when the matrix has more dynamic range than one 31-bit
// word serves: a bucket sum A[a][a] is about (n / V) mean(diag H), its rounding error about (n / V) 0.29 scale, so with
// max|H| <= 16 mean(diag H) one word keeps A to 4e-9 relative -- beyond that the second word takes over.
// ext_mode: -1 decide here, 0 never, 1 always (developer switch GANQ_H_EXT).
__global__ __launch_bounds__(256) void prep_scale_kernel(const float* __restrict__ H, int n, TPrep* prep, int ext_mode) {
    __shared__ double sh[256];
    double s = 0.0;
    for (int u = threadIdx.x; u < n; u += 256) s += (double)H[(int64_t)u * n + u];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float mx = __builtin_bit_cast(float, prep->absmax_bits);
        const bool ok = mx > 0.0f && mx < __builtin_inff();
        prep->scale = ok ? (double)mx / 1073741824.0 : 0.0;
        const double mean_diag = sh[0] / (double)n;
        const bool wide = ok && !((double)mx <= 16.0 * mean_diag);  // also for a non-positive / NaN mean
        prep->ext = ext_mode < 0 ? (wide ? 1 : 0) : (ext_mode ? 1 : 0);
    }
}

// planes[p][v][u] (row pitch nq, zero padded; p = 0..3 digits of I, p = 4..5 digits of J), hdiag_int[u] / hdiag_j[u],
// optionally H64[v][u] = scale * (I + J / 65536), Hint = I (int32), Jint = J (int16)
__global__ __launch_bounds__(256) void hquant_kernel(const float* __restrict__ H, int n, int nq, const TPrep* __restrict__ prep,
                                                    int8_t* __restrict__ planes, int* __restrict__ hdiag_int,
                                                    int* __restrict__ hdiag_j, double* __restrict__ H64,
                                                    int* __restrict__ Hint, short* __restrict__ Jint) {
    const int64_t i4 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;  // 4 consecutive u of one row v
    const int64_t per_row = nq;
    if (i4 >= (int64_t)n * per_row) return;
    const int v = (int)(i4 / per_row), u0 = (int)(i4 % per_row);
    const double scale = prep->scale;
    const bool ext = prep->ext != 0;
    const double inv = scale > 0.0 ? 1.0 / scale : 0.0;
    uint32_t pk[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int u = u0 + k;
        int xi = 0, xj = 0;
        if (u < n) {
            const double x = fmin(fmax((double)H[(int64_t)v * n + u] * inv, -1073741824.0), 1073741824.0);
            xi = (int)__builtin_rint(x);
            if (ext) {
                // remainder in units of 2^-16: two balanced base-256 digits hold [-32896, 32639]
                xj = (int)__builtin_rint((x - (double)xi) * 65536.0);
                if (xj > 32639) {
                    xi += 1;
                    xj -= 65536;
                }
            }
            if (H64) H64[(int64_t)v * n + u] = scale * ((double)xi + (double)xj * EXT_UNIT);
            if (Hint) Hint[(int64_t)v * n + u] = xi;
            if (Jint) Jint[(int64_t)v * n + u] = (short)(xj + JBIAS);
            if (u == v) {
                hdiag_int[u] = xi;
                hdiag_j[u] = xj;
            }
        }
        int r = xi;
#pragma unroll
        for (int p = 0; p < 4; ++p) {  // balanced base-256 digits: r = d + 256 * r', d in [-128, 127]
            const int d = ((r + 128) & 255) - 128;
            r = (r - d) >> 8;
            pk[p] |= (uint32_t)(d & 255) << (8 * k);
        }
        r = xj;
#pragma unroll
        for (int p = 4; p < 6; ++p) {
            const int d = ((r + 128) & 255) - 128;
            r = (r - d) >> 8;
            pk[p] |= (uint32_t)(d & 255) << (8 * k);
        }
    }
#pragma unroll
    for (int p = 0; p < 6; ++p)
        if (p < 4 || ext) *reinterpret_cast<uint32_t*>(planes + ((int64_t)p * n + v) * nq + u0) = pk[p];
}

// per-feature exponents for the W @ H product: e[u] = floor(log2(H_uu) / 2), so that H~[u][v] = H[u][v] 2^(-e_u - e_v) has a
// diagonal in [1, 4) and, H being positive semi-definite, no entry above 4 (wh_gemm.hip)
__global__ __launch_bounds__(256) void dexp_kernel(const int* __restrict__ hdiag_int, const int* __restrict__ hdiag_j,
                                                  const TPrep* __restrict__ prep, int n, int* __restrict__ dexp,
                                                  double* __restrict__ hdiag64) {
    const int u = blockIdx.x * 256 + threadIdx.x;
    if (u >= n) return;
    const double h = prep->scale * ((double)hdiag_int[u] + (double)hdiag_j[u] * EXT_UNIT);
    hdiag64[u] = h;
    int e = 0;
    if (h > 0.0 && h < __builtin_inf()) {
        const int lg = ilogb(h);  // floor(log2 h)
        e = lg >= 0 ? lg / 2 : -((-lg + 1) / 2);  // floor(lg / 2)
    }
    dexp[u] = e;
}

// bits[(row * ng + g) * 16 + a] : bit l set  <=>  Q[row][64 g + l] == a
// `changed` (device, may be null): the full-accumulation kernels run only when *changed > thr
__global__ __launch_bounds__(256) void code_masks_kernel(const uint8_t* __restrict__ Q, int m, int n, int ng,
                                                        unsigned long long* __restrict__ bits,
                                                        const long long* __restrict__ changed, long long thr) {
    if (changed && *changed <= thr) return;
    const int lane = threadIdx.x & 63;
    // grid-stride over (row, 64-column group) items: the grid stays small, so the gated-off launches of the
    // incremental iterations cost a few microseconds instead of 15
    for (int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); item < (int64_t)m * ng; item += (int64_t)gridDim.x * 4) {
        const int row = (int)(item / ng), g = (int)(item % ng);
        const int u = 64 * g + lane;
        const uint32_t q = u < n ? Q[(int64_t)row * n + u] : 255u;
        unsigned long long mine = 0;
#pragma unroll
        for (uint32_t a = 0; a < 16; ++a) {
            const unsigned long long mk = __ballot(q == a);
            if ((uint32_t)lane == a) mine = mk;
        }
        // stored with the two middle 16-bit quarters swapped: dword kb of the word then holds, low half first, the bits of columns
        // 16 kb .. 16 kb + 15 and 32 + 16 kb .. 47 + 16 kb -- exactly what lane half kb of onehot_accum_kernel expands in the two
        // 32-column halves of a step (one 4-byte load per step instead of an 8-byte one)
        mine = (mine & 0xffff00000000ffffull) | ((mine & 0x00000000ffff0000ull) << 16) | ((mine & 0x0000ffff00000000ull) >> 16);
        if (lane < 16) bits[item * 16 + lane] = mine;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// NPL digit planes starting at plane P0: <4, 0> sums the 31-bit word I, <2, 4> the extension word J (into its own
// Mpart; that launch leaves at once when the extension is off)
template <int NPL, int P0>
__device__ __forceinline__ void onehot_accum_body(const int8_t* __restrict__ planes_all, const unsigned long long* __restrict__ bits,
                                                  const uint8_t* __restrict__ Q, int m, int n, int nq, int ng,
                                                  long long* __restrict__ Mpart, long long* __restrict__ stamps,
                                                  const TPrep* __restrict__ prep, const int bid) {
    if (P0 != 0 && prep->ext == 0) return;
    constexpr int BTILE = btile_bytes(NPL);
    const int8_t* planes = planes_all + (int64_t)P0 * n * nq;
    extern __shared__ __align__(16) char smem[];
    long long st_pro = 0, st_loop = 0, st_flush = 0, st_t0 = 0;
    char* Bbuf = smem;                                                        // 2 x BTILE
    long long(*Mrow)[RW][256] = reinterpret_cast<long long(*)[RW][256]>(smem + 2 * BTILE);  // [TW][RW][a*16+b]

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // v_mfma_i32_32x32x32_i8: A rows = (row of the pair, code), 32 u per instruction, B columns = the chunk's 32 v.
    // lane l: A row / B column i32 = l & 31, k bytes 16*(l >> 5) .. +15
    const int i32 = lane & 31, kb = lane >> 5;
    const int r2 = i32 >> 4, a16 = i32 & 15;
    const int nrg = (m + TR - 1) / TR;
    const int rg = bid % nrg, part = bid / nrg;
    const int row0 = rg * TR + wv * RW;
    int rowc[RW];
#pragma unroll
    for (int r = 0; r < RW; ++r) rowc[r] = min(row0 + r, m - 1);
    int mrow[RW / 2];  // the row whose masks this lane expands, per row pair
#pragma unroll
    for (int p = 0; p < RW / 2; ++p) mrow[p] = min(row0 + 2 * p + r2, m - 1);

    for (int i = lane; i < RW * 256; i += 64) Mrow[wv][i >> 8][i & 255] = 0;

    // staging of one digit tile: NPL planes x VCH rows x 256 B (4 planes: 2048 x 16 B, 4 per thread)
    constexpr int NE = NPL * VCH * UC16 / (TW * 64);  // 16-byte pieces of a digit tile per thread
    static_assert(NE * TW * 64 == NPL * VCH * UC16, "digit tile must split evenly over the threads");
    uint4 stage[NE];
    auto gload = [&](int v0, int t) {
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int idx = e * (TW * 64) + tid;
            const int c16 = idx % UC16, vi = (idx / UC16) & (VCH - 1), p = idx / (UC16 * VCH);
            const int v = v0 + vi, ub = t * UT + 16 * c16;
            uint4 x = make_uint4(0, 0, 0, 0);
            if (v < n && ub + 15 > v && ub < nq) {
                x = *reinterpret_cast<const uint4*>(planes + ((int64_t)p * n + v) * nq + ub);
                if (ub <= v) {  // chunk crosses the diagonal: keep bytes with u > v only
                    uint32_t w[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        const int cnt = min(4, max(0, v - (ub + 4 * d) + 1));  // leading bytes to clear
                        w[d] = cnt >= 4 ? 0u : (w[d] & (0xffffffffu << (8 * cnt)));
                    }
                    x = make_uint4(w[0], w[1], w[2], w[3]);
                }
            }
            stage[e] = x;
        }
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int idx = e * (TW * 64) + tid;
            const int c16 = idx % UC16, vi = (idx / UC16) & (VCH - 1), p = idx / (UC16 * VCH);
            *reinterpret_cast<uint4*>(Bbuf + buf * BTILE + (p * VCH + vi) * BROW + 16 * c16) = stage[e];
        }
    };

    const int nchunk = (n + VCH - 1) / VCH;
    const int ntile = (n + UT - 1) / UT;
    // code masks of this lane's (row, code): one dword per 64-column step (see code_masks_kernel), requested KS64 - 1 steps
    // ahead of their use -- one step ahead (round 2) every step waited for its masks: a step is 16 matrix instructions, half a
    // trip to L2
    uint32_t wb[KS64][RW / 2];
    const uint32_t* bits32 = reinterpret_cast<const uint32_t*>(bits);
    auto load_masks = [&](int gi, uint32_t (&dst)[RW / 2]) {
#pragma unroll
        for (int p = 0; p < RW / 2; ++p) dst[p] = gi < ng ? bits32[(((int64_t)mrow[p] * ng + gi) * 16 + a16) * 2 + kb] : 0u;
    };
    // The first digit tile and the first masks of a chunk are requested during the LAST tile of the chunk before it (round 3:
    // a chunk used to open with a load, a store and two barriers in a row -- 7 % of the kernel by its cycle stamps); only the
    // very first chunk of a workgroup waits for them here.  (t_first grows with the chunk: the first chunk past the last tile
    // ends the loop.)
    if (part < nchunk && (part * VCH + 1) / UT < ntile) {
        gload(part * VCH, (part * VCH + 1) / UT);
#pragma unroll
        for (int i = 0; i < KS64 - 1; ++i) load_masks((part * VCH + 1) / UT * KS64 + i, wb[i]);
    }
    for (int c = part; c < nchunk; c += NP) {
        const int v0 = c * VCH;
        const int t_first = (v0 + 1) / UT;
        if (t_first >= ntile) break;
        const int v0n = (c + NP) * VCH, tfn = (v0n + 1) / UT;  // the workgroup's next chunk
        const bool has_next = c + NP < nchunk && tfn < ntile;
        v16i acc[RW / 2][VH][NPL];
#pragma unroll
        for (int p = 0; p < RW / 2; ++p)
#pragma unroll
            for (int vh = 0; vh < VH; ++vh)
#pragma unroll
                for (int d = 0; d < NPL; ++d)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[p][vh][d][r] = 0;

        st_t0 = __builtin_amdgcn_s_memtime();
        __syncthreads();  // previous chunk's last tile fully consumed
        sstore(0);
        __syncthreads();
        { const long long tt = __builtin_amdgcn_s_memtime(); st_pro += tt - st_t0; st_t0 = tt; }
        // one staged tile of UT columns u.  Only the chunk's FIRST tile can touch the diagonal (t_first = (v0 + 1) / UT, so every later
        // tile starts right of v0): its 32-column halves are tested one by one; the others run without a branch in the loop, so the
        // LDS reads of the next half stay in flight across the matrix instructions of this one
        auto tile_body = [&](auto diag_tag, int t) {
            constexpr bool DIAG = decltype(diag_tag)::value;
            const int buf = (t - t_first) & 1;
            if (t + 1 < ntile) gload(v0, t + 1);
            else if (has_next) gload(v0n, tfn);  // (filed by the next chunk's opening)
            // masks of the steps behind this tile: the next tile's, or the first ones of the next chunk
            const int gnext = t + 1 < ntile ? (t + 1) * KS64 : (has_next ? tfn * KS64 : ng);
            const char* Bt = Bbuf + buf * BTILE + i32 * BROW + 16 * kb;
            auto read_b = [&](int ks, int kh, v4i (&bf)[VH][NPL]) {
#pragma unroll
                for (int vh = 0; vh < VH; ++vh)
#pragma unroll
                    for (int d = 0; d < NPL; ++d)
                        bf[vh][d] = *reinterpret_cast<const v4i*>(Bt + (d * VCH + 32 * vh) * BROW + ks * 64 + kh * 32);
            };
            v4i bf[2][VH][NPL];
            read_b(0, 0, bf[0]);
#pragma unroll
            for (int ks = 0; ks < KS64; ++ks) {
                load_masks(ks == 0 ? t * KS64 + KS64 - 1 : gnext + ks - 1, wb[(ks + KS64 - 1) % KS64]);  // KS64 - 1 steps ahead
#pragma unroll
                for (int kh = 0; kh < 2; ++kh) {
                    const int hh = ks * 2 + kh;
                    if (hh + 1 < 2 * KS64) read_b((hh + 1) >> 1, (hh + 1) & 1, bf[(hh + 1) & 1]);  // next 32-column half
#if ACC_PIN_B
                    if (!DIAG) __builtin_amdgcn_sched_barrier(0);  // keep those reads in front of this half's matrix instructions
#endif
                    if (DIAG && t * UT + ks * 64 + kh * 32 + 31 <= v0) continue;  // entirely on or above the diagonal (uniform)
#pragma unroll
                    for (int p = 0; p < RW / 2; ++p) {
                        const uint32_t b16 = (wb[ks][p] >> (16 * kh)) & 0xffffu;
                        v4i af;
#pragma unroll
                        for (int d = 0; d < 4; ++d) af[d] = (int)((((b16 >> (4 * d)) & 0xfu) * 0x00204081u) & 0x01010101u);
#pragma unroll
                        for (int vh = 0; vh < VH; ++vh)
#pragma unroll
                            for (int d = 0; d < NPL; ++d)
                                acc[p][vh][d] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af, bf[hh & 1][vh][d], acc[p][vh][d], 0, 0, 0);
                    }
                }
            }
            if (t + 1 < ntile) sstore(buf ^ 1);
            __syncthreads();
        };
        tile_body(std::true_type{}, t_first);
        for (int t = t_first + 1; t < ntile; ++t) tile_body(std::false_type{}, t);
        { const long long tt = __builtin_amdgcn_s_memtime(); st_loop += tt - st_t0; st_t0 = tt; }
        // bucket the chunk's columns by their code: Mrow[row][a][b] += sum_d 256^d Y_d[(row, a)][v], b = Q[row][v]
        // C layout of the 32x32 MFMA: column = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
        for (int vh = 0; vh < VH; ++vh) {
            const int v = v0 + 32 * vh + i32;
#pragma unroll
            for (int p = 0; p < RW / 2; ++p) {
                uint32_t bq[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) bq[h] = v < n ? Q[(int64_t)rowc[2 * p + h] * n + v] : 255u;
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int i = (reg & 3) + 8 * (reg >> 2) + 4 * kb;  // (row of the pair, code)
                    const int h = i >> 4, a = i & 15;
                    long long val = 0;
#pragma unroll
                    for (int d = 0; d < NPL; ++d) val += (long long)acc[p][vh][d][reg] << (8 * d);
                    const uint32_t b = bq[h];
                    if (b < 16u && val != 0)
                        atomicAdd(reinterpret_cast<unsigned long long*>(&Mrow[wv][2 * p + h][a * 16 + b]),
                                  (unsigned long long)val);
                }
            }
        }
        { const long long tt = __builtin_amdgcn_s_memtime(); st_flush += tt - st_t0; st_t0 = tt; }
    }
    if (stamps && bid == 0 && tid == 0) { stamps[0] = st_pro; stamps[1] = st_loop; stamps[2] = st_flush; }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < RW; ++r)
        if (row0 + r < m) {
            long long* out = Mpart + ((int64_t)part * m + row0 + r) * 256;
            for (int i = lane; i < 256; i += 64) out[i] = Mrow[wv][r][i];
        }
}

// One launch for both words: the first `nblk_hi` workgroups sum the four digit planes of the 31-bit word, the others the two planes
// of the extension word into its own partial sums (they leave at once when the extension is off).  As two launches the second
// one cost a dispatch (~5 us on the device) in every iteration of the loop, also when the full accumulation is gated off.
__global__ __launch_bounds__(TW * 64) void onehot_accum_kernel(const int8_t* __restrict__ planes_all,
                                                              const unsigned long long* __restrict__ bits,
                                                              const uint8_t* __restrict__ Q, int m, int n, int nq, int ng,
                                                              long long* __restrict__ Mpart_hi, long long* __restrict__ Mpart_lo,
                                                              long long* __restrict__ stamps, const long long* __restrict__ changed,
                                                              long long thr, const TPrep* __restrict__ prep, int nblk_hi) {
    if (changed && *changed <= thr) return;
    if ((int)blockIdx.x < nblk_hi) onehot_accum_body<4, 0>(planes_all, bits, Q, m, n, nq, ng, Mpart_hi, stamps, prep, (int)blockIdx.x);
    else onehot_accum_body<2, 4>(planes_all, bits, Q, m, n, nq, ng, Mpart_lo, nullptr, prep, (int)blockIdx.x - nblk_hi);
}

// ---------------------------------------------------------------------------------------------------------------
// 16 lanes per row, 4 rows per wave, 1 wave per workgroup.
#ifndef GANQ_JACOBI_SWEEPS
#define GANQ_JACOBI_SWEEPS 30
#endif
constexpr int JS = 17;
  // padded leading dimension of the fp64 16x16 matrices in LDS

__device__ __forceinline__ double row16_sum(double x) {
    x += __shfl_xor(x, 1, 16);
    x += __shfl_xor(x, 2, 16);
    x += __shfl_xor(x, 4, 16);
    x += __shfl_xor(x, 8, 16);
    return x;
}
__device__ __forceinline__ double row16_max(double x) {
    x = fmax(x, __shfl_xor(x, 1, 16));
    x = fmax(x, __shfl_xor(x, 2, 16));
    x = fmax(x, __shfl_xor(x, 4, 16));
    x = fmax(x, __shfl_xor(x, 8, 16));
    return x;
}
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// ---------------------------------------------------------------------------------------------------------------
// Incremental bucket sums.  F_i[a][b] = sum_{u != v} [Q_iu = a][Q_iv = b] Hint[u][v] (= M + M^T of the full
// accumulation) is an integer, so moving the entries of the columns whose index changed reproduces exactly what a
// full accumulation over the new indices gives.

// Mstate[row] = F = M + M^T with M = sum_p Mpart[p][row]; Qprev = Q   (after a full accumulation)
// is_lo: the sums of the extension word (own Mpart / Mstate; leaves at once when the extension is off; Qprev is the
// other launch's business)
__global__ __launch_bounds__(256) void m_reduce_kernel(const long long* __restrict__ Mpart_hi, const long long* __restrict__ Mpart_lo, int m,
                                                      long long* __restrict__ Mstate_hi, long long* __restrict__ Mstate_lo,
                                                      const uint8_t* __restrict__ Q, uint8_t* __restrict__ Qprev, int64_t qbytes,
                                                      const long long* __restrict__ changed, long long thr,
                                                      const TPrep* __restrict__ prep, int nblk_hi) {
    if (changed && *changed <= thr) return;
    // one launch for both words (a dispatch costs ~5 us on the device even when it leaves at once, and this one sits in every
    // iteration of the loop): the first nblk_hi workgroups sum the 31-bit word and bring Qprev up to date, the others the extension
    const int is_lo = (int)blockIdx.x >= nblk_hi;
    if (is_lo && prep->ext == 0) return;
    const long long* Mpart = is_lo ? Mpart_lo : Mpart_hi;
    long long* Mstate = is_lo ? Mstate_lo : Mstate_hi;
    const int64_t bid = is_lo ? (int64_t)blockIdx.x - nblk_hi : (int64_t)blockIdx.x;
    const int64_t nblk = is_lo ? (int64_t)gridDim.x - nblk_hi : (int64_t)nblk_hi;
    const int64_t total = (int64_t)m * 256;
    for (int64_t i = bid * 256 + threadIdx.x; i < total; i += nblk * 256) {
        const int64_t it = (i & ~255ll) | ((i & 15) << 4) | ((i >> 4) & 15);  // the transposed cell of the same row
        long long s = 0;
#pragma unroll
        for (int p = 0; p < NP; ++p) s += Mpart[(int64_t)p * total + i] + Mpart[(int64_t)p * total + it];
        Mstate[i] = s;
    }
    if (is_lo) return;
    const int64_t q16 = qbytes / 16;
    for (int64_t i = bid * 256 + threadIdx.x; i < q16; i += nblk * 256)
        reinterpret_cast<uint4*>(Qprev)[i] = reinterpret_cast<const uint4*>(Q)[i];
    for (int64_t i = q16 * 16 + bid * 256 + threadIdx.x; i < qbytes; i += nblk * 256) Qprev[i] = Q[i];
}

// one wave per row: columns whose index differs from the previous iteration -> chg[row][0..cnt), chgcnt[row]; *changed += cnt
constexpr int QD_WAVES = 16;  // rows per workgroup: one add to the global change counter per workgroup -- one per row
                              // made 4096 same-address atomics the longest part of the kernel (55 us)
__global__ __launch_bounds__(QD_WAVES * 64) void q_diff_kernel(const uint8_t* __restrict__ Q, const uint8_t* __restrict__ Qprev, int m, int n,
                                                    uint16_t* __restrict__ chg, int* __restrict__ chgcnt,
                                                    long long* __restrict__ changed) {
    __shared__ int wg_total;
    if (threadIdx.x == 0) wg_total = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int row_raw = blockIdx.x * QD_WAVES + (threadIdx.x >> 6);
    const bool active = row_raw < m;
    const int row = active ? row_raw : m - 1;  // a padding wave repeats the last row and writes nothing
    const uint8_t* q = Q + (int64_t)row * n;
    const uint8_t* qp = Qprev + (int64_t)row * n;
    uint16_t* out = chg + (int64_t)row * n;
    int cnt = 0;
    int xdone = 0;
    if ((n & 15) == 0) {  // rows are 16-byte aligned: 1 KB of each row per step, 4 steps in flight
        const int n16 = n >> 4;
        for (int i0 = 0; i0 < n16; i0 += 64 * 4) {
            uint4 a[4], b[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = i0 + 64 * k + lane;
                a[k] = i < n16 ? reinterpret_cast<const uint4*>(q)[i] : make_uint4(0, 0, 0, 0);
                b[k] = i < n16 ? reinterpret_cast<const uint4*>(qp)[i] : make_uint4(0, 0, 0, 0);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = i0 + 64 * k + lane;
                const uint32_t dw[4] = {a[k].x ^ b[k].x, a[k].y ^ b[k].y, a[k].z ^ b[k].z, a[k].w ^ b[k].w};
                if (__ballot((dw[0] | dw[1] | dw[2] | dw[3]) != 0) == 0) continue;  // uniform
#pragma unroll
                for (int w = 0; w < 4; ++w)
#pragma unroll
                    for (int by = 0; by < 4; ++by) {
                        const bool d = ((dw[w] >> (8 * by)) & 0xffu) != 0;
                        const unsigned long long mk = __ballot(d);
                        if (d && active) out[cnt + __popcll(mk & ((1ull << lane) - 1ull))] = (uint16_t)(16 * i + 4 * w + by);
                        cnt += __popcll(mk);
                    }
            }
        }
        xdone = n;
    }
    for (int x0 = xdone; x0 < n; x0 += 64) {
        const int x = x0 + lane;
        const bool d = x < n && q[x] != qp[x];
        const unsigned long long mk = __ballot(d);
        if (d && active) out[cnt + __popcll(mk & ((1ull << lane) - 1ull))] = (uint16_t)x;
        cnt += __popcll(mk);
    }
    if (lane == 0 && active) {
        chgcnt[row] = cnt;
        if (cnt) atomicAdd(&wg_total, cnt);
    }
    __syncthreads();
    if (threadIdx.x == 0 && wg_total) atomicAdd(reinterpret_cast<unsigned long long*>(changed), (unsigned long long)wg_total);
}

// One workgroup per row, its waves share out the changed columns.  Column c = list[e] goes a_old -> a_new.  Taking the
// changes in list order, the codes seen by step e are the new ones for list[e' < e] and the old ones elsewhere, so with
//   S_e[b] = sum_{x != c, old[x] = b} Hint[c][x]  +  sum_{e' < e} Hint[c][list[e']] * ([new[e'] = b] - [old[e'] = b])
// the step is   F[a_old][b] -= S_e[b], F[b][a_old] -= S_e[b], F[a_new][b] += S_e[b], F[b][a_new] += S_e[b]  for every b.
// No S_e depends on F or on another S, so the steps run in parallel and add into the row's F with LDS atomics.
// S_e is a 16-bucket histogram of one row of Hint: lane-private buckets in LDS ([bucket][lane]: no conflicts), then
// a transposed read sums each bucket over the lanes; the second sum is just 2 more entries per earlier change.
constexpr int MU_WAVES = 4;  // 4096x4096 benchmark layer: 2 -> 0.224, 4 -> 0.210, 8 -> 0.250 ms per iteration
// HT = int: the 31-bit word (Hint -> Mstate), updates Qprev at the end; HT = short: the extension word (Jint ->
// Mstate_lo), launched BEFORE the other one (it needs the old codes), leaves at once when the extension is off.
template <typename HT>
__global__ __launch_bounds__(MU_WAVES * 64) void m_update_kernel(const HT* __restrict__ Hint, const uint8_t* __restrict__ Q,
                                                                uint8_t* __restrict__ Qprev, int m, int n,
                                                                const uint16_t* __restrict__ chg, const int* __restrict__ chgcnt,
                                                                long long* __restrict__ Mstate,
                                                                const long long* __restrict__ changed, long long thr,
                                                                const TPrep* __restrict__ prep) {
    if (*changed > thr) return;  // the full accumulation runs instead
    constexpr bool IS_LO = sizeof(HT) == 2;
    constexpr int BIAS = IS_LO ? JBIAS : 0;  // the int16 copy of the extension word is stored biased
    if (IS_LO && prep->ext == 0) return;
    extern __shared__ __align__(16) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int row = blockIdx.x;
    const int cnt = chgcnt[row];
    if (cnt == 0) return;
    long long* Frow = reinterpret_cast<long long*>(smem);                                   // [16][16]
    long long* priv = reinterpret_cast<long long*>(smem) + 256 + wv * (16 * 64);            // [16][64] per wave
    uint8_t* codes = reinterpret_cast<uint8_t*>(smem) + (size_t)(256 + MU_WAVES * 16 * 64) * sizeof(long long);  // old
    long long* Fg = Mstate + (int64_t)row * 256;
    uint8_t* qp = Qprev + (int64_t)row * n;
    const uint8_t* qn = Q + (int64_t)row * n;
    for (int i = tid; i < 256; i += MU_WAVES * 64) Frow[i] = Fg[i];
    for (int x = tid; x < n; x += MU_WAVES * 64) codes[x] = min((int)qp[x], 15);
    __syncthreads();
    const uint16_t* list = chg + (int64_t)row * n;
    // The work of a wave is a stream of (its changed column, 4096-column chunk of that Hint row) items.  Every item is
    // a dependent L2 / HBM round trip, so the next item's 16 KB are always in flight (16 x 16 B per lane) while the
    // current one is bucketed.
    constexpr int CH = 16;                       // int4 loads per lane and chunk: 64 * 16 * 4 = 4096 columns
    const bool vec = !IS_LO && (n & 3) == 0;     // rows of Hint are 16-byte aligned (int32 only)
    const int nchunk = (n + 64 * CH * 4 - 1) / (64 * CH * 4);
    const int mycols = (cnt - wv + MU_WAVES - 1) / MU_WAVES;  // e = wv, wv + MU_WAVES, ..
    const int nitems = mycols * nchunk;
    // this lane's earlier-change partner e' = lane (the first 64 changes; later ones take the slow loop below)
    const int pc = lane < cnt ? list[lane] : 0;
    const int p_old = lane < cnt ? codes[pc] : 0, p_new = lane < cnt ? min((int)qn[pc], 15) : 0;
    auto load_item = [&](int it, int4 (&buf)[CH], int& corr) {
        const int e = wv + (it / nchunk) * MU_WAVES, k = it % nchunk;
        const HT* hrow = Hint + (int64_t)list[e] * n;
        corr = (k == 0 && lane < e) ? (int)hrow[pc] - BIAS : 0;
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const int x = 4 * (64 * (CH * k + j) + lane);
            int4 v = make_int4(0, 0, 0, 0);
            if (vec) {
                if (x < n) v = *reinterpret_cast<const int4*>(reinterpret_cast<const int*>(hrow) + x);
            } else {
                if (x < n) v.x = (int)hrow[x] - BIAS;
                if (x + 1 < n) v.y = (int)hrow[x + 1] - BIAS;
                if (x + 2 < n) v.z = (int)hrow[x + 2] - BIAS;
                if (x + 3 < n) v.w = (int)hrow[x + 3] - BIAS;
            }
            buf[j] = v;
        }
    };
    auto lds_add = [&](long long* p, long long v) {  // fire-and-forget LDS add
        __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    auto bucket_item = [&](int it, const int4 (&buf)[CH], int corr) {
        const int e = wv + (it / nchunk) * MU_WAVES, k = it % nchunk;
        const int c = list[e];
        if (k == 0) {
#pragma unroll
            for (int b = 0; b < 16; ++b) priv[b * 64 + lane] = 0;
            if (lane < e && lane < 64) {  // partner e' = lane < e already carries its new code
                lds_add(&priv[p_new * 64 + lane], (long long)corr);
                lds_add(&priv[p_old * 64 + lane], -(long long)corr);
            }
            for (int e2 = 64 + lane; e2 < e; e2 += 64) {  // more than 64 changes in the row: rare
                const int c2 = list[e2];
                const long long v = (long long)Hint[(int64_t)c * n + c2] - BIAS;
                lds_add(&priv[min((int)qn[c2], 15) * 64 + lane], v);
                lds_add(&priv[(int)codes[c2] * 64 + lane], -v);
            }
        }
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const int x = 4 * (64 * (CH * k + j) + lane);
            const int hv[4] = {buf[j].x, buf[j].y, buf[j].z, buf[j].w};
#pragma unroll
            for (int t = 0; t < 4; ++t)
                if (x + t < n && x + t != c) lds_add(&priv[(int)codes[x + t] * 64 + lane], (long long)hv[t]);
        }
        if (k != nchunk - 1) return;
        // column finished: bucket b = lane & 15 summed over lanes 16*(lane>>4) .. +15, then over the four quarters
        wave_sync();
        const int a_old = codes[c], a_new = min((int)qn[c], 15);
        long long sb = 0;
        {
            const long long* pb = priv + (lane & 15) * 64 + 16 * (lane >> 4);
#pragma unroll
            for (int t = 0; t < 16; ++t) sb += pb[t];
        }
        sb += __shfl_xor(sb, 16);
        sb += __shfl_xor(sb, 32);
        if (lane < 16) {  // lane = b
            lds_add(&Frow[a_old * 16 + lane], -sb);
            lds_add(&Frow[lane * 16 + a_old], -sb);
            lds_add(&Frow[a_new * 16 + lane], sb);
            lds_add(&Frow[lane * 16 + a_new], sb);
        }
        wave_sync();
    };
    if (nitems > 0) {
        int4 buf0[CH], buf1[CH];
        int corr0 = 0, corr1 = 0;
        load_item(0, buf0, corr0);
        for (int it = 0; it < nitems; it += 2) {
            if (it + 1 < nitems) load_item(it + 1, buf1, corr1);
            bucket_item(it, buf0, corr0);
            if (it + 1 >= nitems) break;
            if (it + 2 < nitems) load_item(it + 2, buf0, corr0);
            bucket_item(it + 1, buf1, corr1);
        }
    }
    __syncthreads();
    for (int i = tid; i < 256; i += MU_WAVES * 64) Fg[i] = Frow[i];
    if (!IS_LO)
        for (int e = tid; e < cnt; e += MU_WAVES * 64) qp[list[e]] = qn[list[e]];
}

// The same move on the integer matrix cores (rows whose length is a multiple of 16).  The 16-bucket histograms of the
// changed columns of one row are one small GEMM:  S[b][e] = sum_x [old[x] == b] * Hint[c_e][x]  =  onehot(old codes)
// [16 x n]  @  digits of the rows c_e of H [n x changes], with the 4 int8 digit planes t_prepare already holds
// (v_mfma_i32_16x16x64_i8, 16 changes per tile, up to 2 tiles per pass).  It is formed with the OLD codes everywhere;
// what the list-order semantics of m_update_kernel adds -- the changed column itself is left out, earlier changes of
// the row already carry their new code -- are cnt diagonal entries and cnt^2/2 single entries of Hint, applied to F
// directly.  One workgroup per row whose 4 waves split the columns (a row is a chain of dependent memory round
// trips; partial sums simply add into F, they are integers); the A operand (one-hot bytes) is built from 16 code bytes per lane with a zero-byte
// test, the B operand is 16 contiguous digit bytes of a changed column's row.  All sums are integers: the result is
// bit-identical to m_update_kernel and to the full accumulation.
constexpr int MG_TILES = 2;   // 32 changes per pass over the row (p99 of the benchmark layer's first update: 28)
constexpr int MG_WAVES = 4;   // the waves of a row's workgroup split the columns

__device__ __forceinline__ uint32_t bytes_equal(uint32_t x, uint32_t pat) {  // 0x01 in every byte of x equal to pat's
    const uint32_t y = x ^ pat;
    const uint32_t t = ((y & 0x7f7f7f7fu) + 0x7f7f7f7fu) | y;
    return (~t & 0x80808080u) >> 7;
}

template <int NT, int NPL>
__device__ __forceinline__ void mu_group(const int8_t* __restrict__ planes, int n, int nq, const uint8_t* __restrict__ qp,
                                         const int (&cj)[MG_TILES], int lane, int kb, int ke, v4i (&acc)[MG_TILES][6]) {
    const int i16 = lane & 15, kq = lane >> 4;
    const uint32_t pat = (uint32_t)i16 * 0x01010101u;
    const int64_t plane = (int64_t)n * nq;
    const int8_t* bp[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) bp[t] = planes + (int64_t)cj[t] * nq + 16 * kq;
    if (kb >= ke) return;
    auto load = [&](int k0, uint4& cw, v4i (&bf)[NT][NPL]) {
        cw = *reinterpret_cast<const uint4*>(qp + min(k0 + 16 * kq, n - 16));  // past n the digits are zero
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int p = 0; p < NPL; ++p) bf[t][p] = *reinterpret_cast<const v4i*>(bp[t] + p * plane + k0);
    };
    // every step is one dependent round trip to L2 / Infinity Cache: D steps of operands are kept in flight
    // (with the extension's six planes half as many: the kernel then fits 168 registers -- three workgroups per CU instead of two for
    // a kernel that is a chain of round trips per row; the six-plane path is the rare one)
    constexpr int D = NPL == 6 ? (NT == 1 ? 2 : 1) : (NT == 1 ? 4 : 2);
    uint4 cw[D];
    v4i bf[D][NT][NPL];
    auto step = [&](const uint4& c, const v4i (&b)[NT][NPL]) {
        v4i af;
        af[0] = (int)bytes_equal(c.x, pat);
        af[1] = (int)bytes_equal(c.y, pat);
        af[2] = (int)bytes_equal(c.z, pat);
        af[3] = (int)bytes_equal(c.w, pat);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int p = 0; p < NPL; ++p) acc[t][p] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af, b[t][p], acc[t][p], 0, 0, 0);
    };
#pragma unroll
    for (int d = 0; d < D; ++d) load(min(kb + 64 * d, ke - 64), cw[d], bf[d]);
    for (int k0 = kb; k0 < ke; k0 += 64 * D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            if (k0 + 64 * d < ke) step(cw[d], bf[d]);                    // uniform
            load(min(k0 + 64 * (d + D), ke - 64), cw[d], bf[d]);        // clamped: the tail re-reads its last step
        }
    }
}

// With the extension word on (prep->ext, uniform) the two extra digit planes ride along in the same pass and move the
// second set of sums Mstate_lo; the pair / diagonal corrections come from Jint / hdiag_j.
__global__ __launch_bounds__(MG_WAVES * 64, 3) void m_update_mfma_kernel(const int8_t* __restrict__ planes, const int* __restrict__ Hint,
                                                           const short* __restrict__ Jint,
                                                           const int* __restrict__ hdiag_int, const int* __restrict__ hdiag_j,
                                                           const uint8_t* __restrict__ Q,
                                                           uint8_t* __restrict__ Qprev, int m, int n, int nq,
                                                           const uint16_t* __restrict__ chg, const int* __restrict__ chgcnt,
                                                           long long* __restrict__ Mstate, long long* __restrict__ Mstate_lo,
                                                           const long long* __restrict__ changed, long long thr,
                                                           const TPrep* __restrict__ prep) {
    if (*changed > thr) return;  // the full accumulation runs instead
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int cnt = chgcnt[row];
    if (cnt == 0) return;
    const bool ext = prep->ext != 0;
    __shared__ long long Frow[256];
    __shared__ long long Flo[256];
    long long* Fg = Mstate + (int64_t)row * 256;
    long long* Fgl = Mstate_lo + (int64_t)row * 256;
    uint8_t* qp = Qprev + (int64_t)row * n;
    const uint8_t* qn = Q + (int64_t)row * n;
    const uint16_t* list = chg + (int64_t)row * n;
    Frow[tid] = Fg[tid];
    if (ext) Flo[tid] = Fgl[tid];
    // the row's changes (column, old code, new code): one parallel round of loads instead of dependent ones per use
    constexpr int CAP = 512;
    __shared__ uint16_t s_col[CAP];
    __shared__ uint8_t s_old[CAP], s_new[CAP];
    for (int e = tid; e < min(cnt, CAP); e += MG_WAVES * 64) {
        const int c = list[e];
        s_col[e] = (uint16_t)c;
        s_old[e] = (uint8_t)min((int)qp[c], 15);
        s_new[e] = (uint8_t)min((int)qn[c], 15);
    }
    __syncthreads();
    auto col_of = [&](int e) -> int { return e < CAP ? (int)s_col[e] : (int)list[e]; };
    auto old_of = [&](int e) -> int { return e < CAP ? (int)s_old[e] : min((int)qp[list[e]], 15); };
    auto new_of = [&](int e) -> int { return e < CAP ? (int)s_new[e] : min((int)qn[list[e]], 15); };
    const int ksteps = (n + 63) >> 6, kper = (ksteps + MG_WAVES - 1) / MG_WAVES;
    const int kb = min(wv * kper, ksteps) * 64, ke = min((wv + 1) * kper, ksteps) * 64;  // this wave's columns
    auto lds_add = [&](long long* p, long long v) { __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    // F[a_old][b] -= s, F[b][a_old] -= s, F[a_new][b] += s, F[b][a_new] += s
    auto move_in = [&](long long* F, int a_old, int a_new, int b, long long s) {
        lds_add(&F[a_old * 16 + b], -s);
        lds_add(&F[b * 16 + a_old], -s);
        lds_add(&F[a_new * 16 + b], s);
        lds_add(&F[b * 16 + a_new], s);
    };
    auto move = [&](int a_old, int a_new, int b, long long s) { move_in(Frow, a_old, a_new, b, s); };
    const int i16 = lane & 15, kq = lane >> 4;
    for (int g0 = 0; g0 < cnt; g0 += 16 * MG_TILES) {
        const int nt = min(MG_TILES, (cnt - g0 + 15) >> 4);
        int cj[MG_TILES];
#pragma unroll
        for (int t = 0; t < MG_TILES; ++t) {
            const int e = g0 + 16 * t + i16;
            cj[t] = col_of(e < cnt ? e : g0);  // padding columns repeat a valid one; their sums are dropped below
        }
        v4i acc[MG_TILES][6];
#pragma unroll
        for (int t = 0; t < MG_TILES; ++t)
#pragma unroll
            for (int p = 0; p < 6; ++p) acc[t][p] = v4i{0, 0, 0, 0};
        if (ext) {
            if (nt == 1) mu_group<1, 6>(planes, n, nq, qp, cj, lane, kb, ke, acc);
            else mu_group<2, 6>(planes, n, nq, qp, cj, lane, kb, ke, acc);
        } else {
            if (nt == 1) mu_group<1, 4>(planes, n, nq, qp, cj, lane, kb, ke, acc);
            else mu_group<2, 4>(planes, n, nq, qp, cj, lane, kb, ke, acc);
        }
        // D layout: lane (j = lane & 15, kq) holds S[b = 4 kq + r][change j], r = 0..3
#pragma unroll
        for (int t = 0; t < MG_TILES; ++t) {
            const int e = g0 + 16 * t + i16;
            if (t < nt && e < cnt) {
                const int a_old = old_of(e), a_new = new_of(e);
                const long long self = wv == 0 ? (long long)hdiag_int[cj[t]] : 0;
                const long long self_lo = (ext && wv == 0) ? (long long)hdiag_j[cj[t]] : 0;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int b = 4 * kq + r;
                    long long sb = (long long)acc[t][0][r] + ((long long)acc[t][1][r] << 8) + ((long long)acc[t][2][r] << 16) +
                                   ((long long)acc[t][3][r] << 24);
                    if (b == a_old) sb -= self;  // the column itself is not part of its own histogram (taken out once)
                    if (sb != 0) move(a_old, a_new, b, sb);
                    if (ext) {
                        long long sl = (long long)acc[t][4][r] + ((long long)acc[t][5][r] << 8);
                        if (b == a_old) sl -= self_lo;
                        if (sl != 0) move_in(Flo, a_old, a_new, b, sl);
                    }
                }
            }
        }
    }
    // earlier changes e' < e of the row already carry their new code when change e is applied: all pairs at once
    for (int64_t pidx = tid; pidx < (int64_t)cnt * cnt; pidx += MG_WAVES * 64) {
        const int e = (int)(pidx / cnt), e2 = (int)(pidx - (int64_t)e * cnt);
        if (e2 >= e) continue;
        const long long v = (long long)Hint[(int64_t)col_of(e) * n + col_of(e2)];
        if (v != 0) {
            const int a_old = old_of(e), a_new = new_of(e);
            move(a_old, a_new, new_of(e2), v);
            move(a_old, a_new, old_of(e2), -v);
        }
        if (ext) {
            const long long vl = (long long)Jint[(int64_t)col_of(e) * n + col_of(e2)] - JBIAS;
            if (vl != 0) {
                const int a_old = old_of(e), a_new = new_of(e);
                move_in(Flo, a_old, a_new, new_of(e2), vl);
                move_in(Flo, a_old, a_new, old_of(e2), -vl);
            }
        }
    }
    __syncthreads();
    Fg[tid] = Frow[tid];
    if (ext) Fgl[tid] = Flo[tid];
    for (int e = tid; e < cnt; e += MG_WAVES * 64) qp[list[e]] = qn[list[e]];
}

__device__ __forceinline__ void lds_add(double* p, double v) {
    (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_add(long long* p, long long v) {
    (void)__hip_atomic_fetch_add(reinterpret_cast<unsigned long long*>(p), (unsigned long long)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

template <typename WHT>
__global__ __launch_bounds__(64) void t_solve_kernel(const long long* __restrict__ Mpart, const long long* __restrict__ Mpart_lo,
                                                     const TPrep* __restrict__ prep, const int* __restrict__ hdiag_int,
                                                     const int* __restrict__ hdiag_j, const WHT* __restrict__ WH,
                                                     const double* __restrict__ wHw, const uint8_t* __restrict__ Q, int m,
                                                     int n, int V, double rcond, float* __restrict__ T_out,
                                                     float* __restrict__ A_out, float* __restrict__ b_out,
                                                     double* __restrict__ loss_rows, int nparts,
                                                     long long* __restrict__ changed_reset, int allow_fast,
                                                     const int* __restrict__ rowlist, const int* __restrict__ nrows) {
    // the change counter of this iteration has been consumed by the kernels in front of this one: leave it zero for
    // the next iteration's q_diff_kernel (saves a memset launch per iteration)
    if (changed_reset && blockIdx.x == 0 && threadIdx.x == 0) *changed_reset = 0;
    __shared__ double As[4][16][JS];
    __shared__ double Es[4][16][JS];
    __shared__ double A0[4][16][JS];       // unrounded A (for the loss), also scratch for the bucket sums
    __shared__ long long Di[4][16][JS];    // lane-private integer buckets of diag(H), then scratch for the transpose
    // the same for the extension word: aliases Es, which is first written after the bucket sums have been consumed
    long long(*Dj)[16][JS] = reinterpret_cast<long long(*)[16][JS]>(&Es[0][0][0]);
    __shared__ double CS[4][8][2];
    __shared__ double Coef[4][16];

    const int lane = threadIdx.x & 63;
    const int rs = lane >> 4, l = lane & 15;
    // rowlist (iterations >= 1 of the fused loop): only the rows whose indices changed in this iteration are solved -- an unchanged row has
    // the same bucket sums and the same b, hence the same codebook and loss, which stay where they are (T_out is then the previous
    // codebook array, updated in place).  The row's work is dominated by streaming its 32 KB of W H, so the kernel's time follows
    // the number of listed rows.
    int row = blockIdx.x * 4 + rs;
    if (rowlist) {
        const int cnt = *nrows;
        if ((int)blockIdx.x * 4 >= cnt) return;  // (uniform; behind the counter reset above)
        row = row < cnt ? rowlist[row] : m;      // surplus row slots of the last workgroup: compute a valid row, store nothing
    }
    const int rowc = min(row, m - 1);
    double(*A)[JS] = As[rs];
    double(*E)[JS] = Es[rs];
    double(*B0)[JS] = A0[rs];
    const double scale = prep->scale;
    const bool ext = prep->ext != 0;  // uniform

    // ---- b_i[a] = sum_{u in a} WH[row][u] and D[a] = sum_{u in a} H[u][u]: lane l takes u = 16 t + l into its own
    //      16 buckets (no atomics), then lane a sums bucket a over the lanes in fixed order
#pragma unroll
    for (int a = 0; a < 16; ++a) {
        B0[l][a] = 0.0;
        Di[rs][l][a] = 0;
        Dj[rs][l][a] = 0;
    }
    {
        const uint8_t* q = Q + (int64_t)rowc * n;
        const WHT* wh = WH + (int64_t)rowc * n;
        // whole blocks of KB x 16 columns: unconditional loads, all of a block's requests sent before its first element is
        // used -- the wave is alone on its SIMD, a block is one trip to L2 (blocks of 8 elements per lane: 32 trips in a row at
        // n = 4096, half of this kernel's time; 32 elements: 8)
        auto blocks = [&](auto kb_tag, int u_begin, int count) {
            constexpr int KB = decltype(kb_tag)::value;
            for (int blk = 0; blk < count; ++blk) {
                const int u0 = u_begin + blk * (16 * KB) + l;
                int av[KB], hv[KB];
                double wv8[KB];
#pragma unroll
                for (int k = 0; k < KB; ++k) {
                    av[k] = min((int)q[u0 + 16 * k], 15);
                    wv8[k] = (double)wh[u0 + 16 * k];
                    hv[k] = hdiag_int[u0 + 16 * k];
                }
                // LDS atomics without a return value instead of read-modify-write: a `+=` on a bucket is a dependent LDS round
                // trip per element (two buckets may be the same one, so the compiler keeps them in order); ds_add_f64 /
                // ds_add_u64 are sent off in program order -- the LDS executes a wave's operations in order, so every bucket
                // still sums its elements in ascending column order, the same bits as before
#pragma unroll
                for (int k = 0; k < KB; ++k) {
                    lds_add(&B0[l][av[k]], wv8[k]);
                    lds_add(&Di[rs][l][av[k]], (long long)hv[k]);
                }
            }
        };
        const int nbig = n / 512, nfull = (n - nbig * 512) / 128;
        blocks(std::integral_constant<int, 32>{}, 0, nbig);
        blocks(std::integral_constant<int, 8>{}, nbig * 512, nfull);
        for (int u = nbig * 512 + nfull * 128 + l; u < n; u += 16) {
            const int a = min((int)q[u], 15);
            lds_add(&B0[l][a], (double)wh[u]);
            lds_add(&Di[rs][l][a], (long long)hdiag_int[u]);
        }
        if (ext)
            for (int u = l; u < n; u += 16) lds_add(&Dj[rs][l][min((int)q[u], 15)], (long long)hdiag_j[u]);
    }
    wave_sync();
    double bsum = 0.0;
    long long dsum = 0, dsum_lo = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        bsum += B0[k][l];
        dsum += Di[rs][k][l];
        dsum_lo += Dj[rs][k][l];
    }
    wave_sync();

    // ---- A = M + M^T + diag  (exact integers), lane l owns column l ----
    long long col[16];
#pragma unroll
    for (int a = 0; a < 16; ++a) col[a] = 0;
    const bool sym_in = nparts == 0;  // Mpart is already F = M + M^T (one part)
    for (int p = 0; p < (sym_in ? 1 : nparts); ++p) {
        const long long* src = Mpart + ((int64_t)p * m + rowc) * 256;
#pragma unroll
        for (int a = 0; a < 16; ++a) col[a] += src[a * 16 + l];
    }
#pragma unroll
    for (int a = 0; a < 16; ++a) Di[rs][a][l] = col[a];
    if (ext) {  // the extension word's sums, same layout
#pragma unroll
        for (int a = 0; a < 16; ++a) col[a] = 0;
        for (int p = 0; p < (sym_in ? 1 : nparts); ++p) {
            const long long* src = Mpart_lo + ((int64_t)p * m + rowc) * 256;
#pragma unroll
            for (int a = 0; a < 16; ++a) col[a] += src[a * 16 + l];
        }
#pragma unroll
        for (int a = 0; a < 16; ++a) Dj[rs][a][l] = col[a];
    }
    wave_sync();
    double colA[16], colA0[16];
#pragma unroll
    for (int a = 0; a < 16; ++a) {
        long long x = sym_in ? Di[rs][a][l] : Di[rs][a][l] + Di[rs][l][a];
        if (a == l) x += dsum;
        double xd = (double)x;
        if (ext) {
            long long xl = sym_in ? Dj[rs][a][l] : Dj[rs][a][l] + Dj[rs][l][a];
            if (a == l) xl += dsum_lo;
            xd += (double)xl * EXT_UNIT;
        }
        colA0[a] = scale * xd;
        colA[a] = (double)(float)colA0[a];  // the reference holds A and b in fp32
    }
    const double bl0 = bsum;
    const double bl = (double)(float)bsum;
    wave_sync();
#pragma unroll
    for (int a = 0; a < 16; ++a) {
        A[a][l] = colA[a];
        B0[a][l] = colA0[a];
        E[a][l] = (a == l) ? 1.0 : 0.0;
    }
    if (row < m) {
        if (A_out && l < V)
            for (int a = 0; a < V; ++a) A_out[((int64_t)row * V + a) * V + l] = (float)colA[a];
        if (b_out && l < V) b_out[(int64_t)row * V + l] = (float)bl;
    }
    wave_sync();

    // ---- fast path.  gelsd keeps every singular value larger than rcond * sigma_max; when the row's A is positive
    //      definite with  sigma_min >= 1 / ||L^-1||_F^2  >  4 rcond trace(A) >= 4 rcond sigma_max  (a guaranteed bound
    //      from its Cholesky factor), nothing is cut off and the minimum-norm solution is simply A^-1 b.  Then
    //      t = L^-T (L^-1 b) needs 16 pivots instead of ~60 Jacobi rounds.  All in registers: lane l owns row l of L and
    //      column l of L^-1, values travel by 16-lane shuffles.  Rows with an unused code (a zero pivot), a tiny
    //      pivot or a large inverse take the Jacobi path below (the other rows of their wave walk through it with them and
    //      keep their own result).
    double x = 0.0;
    bool fast = false;
    if (allow_fast) {
        double r[16];
#pragma unroll
        for (int a = 0; a < 16; ++a) r[a] = (l < V && a < V) ? colA[a] : ((a == l) ? 1.0 : 0.0);  // A[l][a] (symmetric)
        bool ok = true;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const double dk = __shfl(r[k], k, 16);
            ok = ok && (dk > 0.0);
            const double sk = sqrt(dk);
            const double lk = (l == k) ? sk : r[k] / sk;  // L[l][k] for l >= k
            r[k] = lk;
#pragma unroll
            for (int j = k + 1; j < 16; ++j) r[j] -= lk * __shfl(lk, j, 16);  // A[l][j] -= L[l][k] L[j][k]
        }
        double xc[16], yv[16];  // column l of L^-1; y = L^-1 b (every lane holds all of it)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            double sx = (i == l) ? 1.0 : 0.0;
            double sy = __shfl(bl, i, 16);
#pragma unroll
            for (int j = 0; j < i; ++j) {
                const double lij = __shfl(r[j], i, 16);
                sx -= lij * xc[j];
                sy -= lij * yv[j];
            }
            const double lii = __shfl(r[i], i, 16);
            xc[i] = sx / lii;
            yv[i] = sy / lii;
        }
        double n2 = 0.0;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            n2 += xc[i] * xc[i];
            x += xc[i] * yv[i];  // t_l = (L^-T y)_l = sum_i L^-1[i][l] y_i
        }
        const double inv_norm2 = row16_sum(l < V ? n2 : 0.0);     // ||L^-1||_F^2 >= 1 / sigma_min(A)
        const double tr = row16_sum(l < V ? colA[l] : 0.0);       // trace(A) >= sigma_max(A)
        fast = ok && (4.0 * rcond * tr * inv_norm2 < 1.0);  // false for NaN; decided per row: a row's result must not
                                                             // depend on which rows share its wave
    }
    const double x_fast = x;
    if (!__all(fast)) {
        // ---- round-robin Jacobi: 15 rounds of 8 disjoint rotations per sweep ----
        const int t = l >> 1;  // pair handled (redundantly) by lanes 2t, 2t+1
        for (int sweep = 0; sweep < GANQ_JACOBI_SWEEPS; ++sweep) {
            double off = 0.0, dg = 0.0;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const double x = A[l][j];
                if (j == l) dg += x * x; else off += x * x;
            }
            off = row16_sum(off);
            dg = row16_sum(dg);
            // squared off-diagonal mass below 1e-22 of the diagonal's: the eigenvalues are then converged to ~1e-22 relative
            // (second order), the eigenvectors to ~1e-11 -- six orders below what the fp32 codebook keeps
            const bool done = (off <= 1e-22 * dg) || (off == 0.0);
            if (__all(done)) break;
            for (int r = 0; r < 15; ++r) {
                int p, q;
                if (t == 0) {
                    p = r;
                    q = 15;
                } else {
                    const int x = (r + t) % 15, y = (r - t + 15) % 15;
                    p = min(x, y);
                    q = max(x, y);
                }
                const double apq = A[p][q], app = A[p][p], aqq = A[q][q];
                double cc = 1.0, ss = 0.0;
                if (apq != 0.0 && !done) {
                    // t = sgn(theta) / (|theta| + sqrt(theta^2 + 1)) with theta = alpha / apq, written without forming theta:
                    // t = sgn(alpha) apq / (|alpha| + hypot(alpha, apq))  -- one division and one square root fewer
                    const double alpha = 0.5 * (aqq - app);
                    const double rr = sqrt(alpha * alpha + apq * apq);
                    const double tt = (alpha >= 0.0 ? apq : -apq) / (fabs(alpha) + rr);
                    cc = 1.0 / sqrt(tt * tt + 1.0);
                    ss = tt * cc;
                }
                if ((l & 1) == 0) {
                    CS[rs][t][0] = cc;
                    CS[rs][t][1] = ss;
                }
                wave_sync();
                // column phase: lane l rotates row l's entries (p_i, q_i) for all 8 pairs
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    int pi, qi;
                    if (i == 0) { pi = r; qi = 15; } else {
                        const int x = (r + i) % 15, y = (r - i + 15) % 15;
                        pi = min(x, y); qi = max(x, y);
                    }
                    const double ci = CS[rs][i][0], si = CS[rs][i][1];
                    const double akp = A[l][pi], akq = A[l][qi];
                    A[l][pi] = ci * akp - si * akq;
                    A[l][qi] = si * akp + ci * akq;
                    const double ekp = E[l][pi], ekq = E[l][qi];
                    E[l][pi] = ci * ekp - si * ekq;
                    E[l][qi] = si * ekp + ci * ekq;
                }
                wave_sync();
                // row phase: lane l rotates column l's entries of rows (p_i, q_i)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    int pi, qi;
                    if (i == 0) { pi = r; qi = 15; } else {
                        const int x = (r + i) % 15, y = (r - i + 15) % 15;
                        pi = min(x, y); qi = max(x, y);
                    }
                    const double ci = CS[rs][i][0], si = CS[rs][i][1];
                    const double apk = A[pi][l], aqk = A[qi][l];
                    A[pi][l] = ci * apk - si * aqk;
                    A[qi][l] = si * apk + ci * aqk;
                }
                wave_sync();
            }
        }

        // ---- minimum-norm solution: x = sum_k [|lam_k| > rcond * lam_max] (e_k . b / lam_k) e_k ----
        const double lam = A[l][l];
        const double lmax = row16_max(fabs(lam));
        Coef[rs][l] = bl;
        wave_sync();
        double proj = 0.0;
#pragma unroll
        for (int a = 0; a < 16; ++a) proj += E[a][l] * Coef[rs][a];
        wave_sync();
        const bool keep = fabs(lam) > rcond * lmax;
        Coef[rs][l] = keep ? proj / lam : 0.0;
        wave_sync();
        x = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) x += E[l][k] * Coef[rs][k];
    }
    if (fast) x = x_fast;
    const float tl = (float)x;
    if (row < m && l < V) T_out[(int64_t)row * V + l] = tl;

    // ---- loss of this row with the NEW codebook: w^T H w - 2 t^T b + t^T A t (fp64, unrounded A and b) ----
    if (loss_rows) {
        const double td = (l < V) ? (double)tl : 0.0;
        wave_sync();
        Coef[rs][l] = td;
        wave_sync();
        double at = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) at += B0[l][k] * Coef[rs][k];  // (A t)_l, A symmetric
        const double term = row16_sum(td * at - 2.0 * td * bl0);
        if (row < m && l == 0) loss_rows[row] = wHw[row] + term;
    }
}

// w_i^T (W H)_i per row, fixed order
__global__ __launch_bounds__(256) void whw_kernel(const float* __restrict__ W, const double* __restrict__ WH64, int m, int n,
                                                 double* __restrict__ wHw) {
    __shared__ double sh[256];
    const int row = blockIdx.x;
    double s = 0.0;
    for (int u = threadIdx.x; u < n; u += 256) s += (double)W[(int64_t)row * n + u] * WH64[(int64_t)row * n + u];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) wHw[row] = sh[0];
}

__global__ __launch_bounds__(256) void sum_rows_kernel(const double* __restrict__ rows, int m, double* __restrict__ out) {
    __shared__ double sh[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < m; i += 256) s += rows[i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = sh[0];
}

// ---------------------------------------------------------------------------------------------------------------
// fp64 GEMM C[M,N] = A[M,K] (fp32, widened on load) @ B[K,N] (fp64) on v_mfma_f64_16x16x4_f64; used once per layer
// for W @ H_fixed.  64x64 tile, 4 waves x (2x2 tiles of 16x16), K slabs of 16 through LDS.
// C[M,N] (fp64) = A[M,K] (fp32, widened) @ B[K,N] (fp64) on v_mfma_f64_16x16x4_f64.
// 128x128 tile per workgroup, 4 waves of 64x64 (4x4 MFMA tiles: 8 LDS reads per 16 MFMAs), 16-deep k slabs double
// buffered in LDS, the next slab's global loads in flight during the current slab's MFMAs.
constexpr int DM = 128, DN = 128, DK = 16;
constexpr int DPAD = 4;  // row pitch DM + 4 doubles: the 16 lanes x 4 k-rows of an operand read hit distinct banks

__global__ __launch_bounds__(256) void gemm_f64_kernel(const float* __restrict__ A, const double* __restrict__ B,
                                                      double* __restrict__ C, int M, int N, int K) {
    __shared__ double As[2][DK][DM + DPAD];
    __shared__ double Bs[2][DK][DN + DPAD];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int tiles_n = (N + DN - 1) / DN;
    const int bm = (blockIdx.x / tiles_n) * DM, bn = (blockIdx.x % tiles_n) * DN;
    const int wm = (wv >> 1) * 64, wn = (wv & 1) * 64;
    f64x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f64x4){0.0, 0.0, 0.0, 0.0};

    // slab loads: A 128 rows x 16 k (fp32): thread -> row tid/2, 8 consecutive k; B 16 k x 128 cols (fp64): thread ->
    // k tid/16, 8 consecutive columns
    const int a_row = tid >> 1, a_k = (tid & 1) * 8;
    const int b_k = tid >> 4, b_col = (tid & 15) * 8;
    float ra[8];
    double rb[8];
    auto gload = [&](int k0) {
        const int row = bm + a_row;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = k0 + a_k + e;
            ra[e] = (row < M && k < K) ? A[(int64_t)row * K + k] : 0.0f;
        }
        const int kk = k0 + b_k;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int col = bn + b_col + e;
            rb[e] = (kk < K && col < N) ? B[(int64_t)kk * N + col] : 0.0;
        }
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int e = 0; e < 8; ++e) As[buf][a_k + e][a_row] = (double)ra[e];
#pragma unroll
        for (int e = 0; e < 8; ++e) Bs[buf][b_k][b_col + e] = rb[e];
    };
    gload(0);
    sstore(0);
    __syncthreads();
    const int nslab = (K + DK - 1) / DK;
    for (int sidx = 0; sidx < nslab; ++sidx) {
        const int buf = sidx & 1;
        if (sidx + 1 < nslab) gload((sidx + 1) * DK);
#pragma unroll
        for (int kk = 0; kk < DK; kk += 4) {
            const int kq = kk + (lane >> 4);
            double a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = As[buf][kq][wm + 16 * i + (lane & 15)];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = Bs[buf][kq][wn + 16 * j + (lane & 15)];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (sidx + 1 < nslab) sstore(buf ^ 1);
        __syncthreads();
    }
    // f64 C/D layout: col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = bm + wm + 16 * i + (lane >> 4) + 4 * r;
                const int col = bn + wn + 16 * j + (lane & 15);
                if (row < M && col < N) C[(int64_t)row * N + col] = acc[i][j][r];
            }
}

// ---------------------------------------------------------------------------------------------------------------
// option GANQ_WH_F64=1: W @ H_fixed by the fp64 GEMM instead of the split-fp16 product -- the A/B reference
static bool wh_use_f64_gemm() { return opt_get(OPT_WH_F64) == 1; }

TLayout t_layout(int64_t m, int64_t n, bool with_f64) {
    TLayout lo;
    lo.nq = (n + UT - 1) / UT * UT;
    lo.ng = (n + 63) / 64;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        size_t o = off;
        off = align_up(off + bytes, 256);
        return o;
    };
    lo.off_prep = take(sizeof(TPrep));
    lo.off_planes = take((size_t)6 * n * lo.nq);  // 4 digit planes of I + 2 of the extension word J
    lo.off_hdiag = take((size_t)n * sizeof(int));
    lo.off_hdiag_j = take((size_t)n * sizeof(int));
    lo.off_bits = take((size_t)m * lo.ng * 16 * sizeof(unsigned long long));
    lo.off_mpart = take((size_t)NP * m * 256 * sizeof(long long));
    lo.off_mpart_lo = take((size_t)NP * m * 256 * sizeof(long long));
    lo.off_mstate_lo = lo.off_jint = lo.off_dexp = lo.off_hdiag64 = 0;
    lo.off_h64 = lo.off_wh64 = lo.off_whw = lo.off_lossrows = 0;
    lo.off_hint = lo.off_qprev = lo.off_mstate = lo.off_chg = lo.off_chgcnt = 0;
    lo.off_wp = lo.off_hp = lo.off_rexp = lo.off_wlo = 0;
    lo.off_active = 0;
    if (with_f64) {
        lo.off_hint = take((size_t)n * n * sizeof(int));
        lo.off_qprev = take((size_t)m * n);
        lo.off_mstate = take((size_t)m * 256 * sizeof(long long));
        lo.off_mstate_lo = take((size_t)m * 256 * sizeof(long long));
        lo.off_jint = take((size_t)n * n * sizeof(short));
        lo.off_dexp = take((size_t)n * sizeof(int));
        lo.off_hdiag64 = take((size_t)n * sizeof(double));
        lo.off_chg = take((size_t)m * n * sizeof(uint16_t));
        lo.off_chgcnt = take((size_t)m * sizeof(int) + 64);  // counts per row, then the 8-byte total
        lo.off_active = take(((size_t)m + 1) * sizeof(int));  // rows that changed in the last iteration, their number
        if (wh_use_f64_gemm()) {
            lo.off_h64 = take((size_t)n * n * sizeof(double));
        } else {
            const WhLayout wl = wh_layout(m, n);
            lo.off_wp = take(wl.wp_bytes);
            lo.off_hp = take(wl.hp_bytes);
            lo.off_rexp = take(wl.rexp_bytes);
            lo.off_wlo = take(wl.wlo_bytes);
        }
        lo.off_wh64 = take((size_t)m * n * sizeof(double));
        lo.off_whw = take((size_t)m * sizeof(double));
        lo.off_lossrows = take((size_t)m * sizeof(double));
    }
    lo.total = off;
    return lo;
}

__global__ __launch_bounds__(256) void hfixed_kernel(const int* __restrict__ Hint, const short* __restrict__ Jint,
                                                    const TPrep* __restrict__ prep, int64_t total, double* __restrict__ out) {
    const bool ext = prep->ext != 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256)
        out[i] = prep->scale * ((double)Hint[i] + (ext ? (double)((int)Jint[i] - JBIAS) * EXT_UNIT : 0.0));
}

// rows whose indices changed in this iteration (ascending) and their number.  A row without a change has reached a
// fixed point: its bucket sums, hence its codebook, hence its next indices repeat -- the S-solve skips it from now on.
__global__ __launch_bounds__(1024) void active_rows_kernel(const int* __restrict__ chgcnt, int m, int* __restrict__ list,
                                                           int* __restrict__ count) {
    __shared__ int wsum[16];
    __shared__ int carry_sh;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) carry_sh = 0;
    __syncthreads();
    for (int base = 0; base < m; base += 1024) {
        const int i = base + tid;
        const int v = (i < m && chgcnt[i] > 0) ? 1 : 0;
        int incl = v;
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(incl, off);
            if (lane >= off) incl += t;
        }
        if (lane == 63) wsum[wv] = incl;
        __syncthreads();
        int before = carry_sh;
        for (int k = 0; k < wv; ++k) before += wsum[k];
        if (v) list[before + incl - 1] = i;
        __syncthreads();
        if (tid == 1023) carry_sh = before + incl;
        __syncthreads();
    }
    if (tid == 0) *count = carry_sh;
}

// once per layer: fixed-point planes of H (+ optionally W @ H_fixed in fp64 and w^T H w per row)
int t_prepare(const float* W, const float* H, int64_t m, int64_t n, const TLayout& lo, char* ws, bool with_f64,
              hipStream_t stream) {
    TPrep* prep = reinterpret_cast<TPrep*>(ws + lo.off_prep);
    int8_t* planes = reinterpret_cast<int8_t*>(ws + lo.off_planes);
    int* hdiag = reinterpret_cast<int*>(ws + lo.off_hdiag);
    int* hdiag_j = reinterpret_cast<int*>(ws + lo.off_hdiag_j);
    double* H64 = (with_f64 && lo.off_h64) ? reinterpret_cast<double*>(ws + lo.off_h64) : nullptr;
    ProfScope prof(KID_T_PREP, stream);
    GANQ_HIP_CHECK(hipMemsetAsync(prep, 0, sizeof(TPrep), stream));
    hipLaunchKernelGGL(absmax_kernel, dim3(512), dim3(256), 0, stream, H, n * n, prep);
    hipLaunchKernelGGL(prep_scale_kernel, dim3(1), dim3(256), 0, stream, H, (int)n, prep, (int)opt_get(OPT_H_EXT));
    const int64_t quads = n * lo.nq / 4;
    int* Hint = with_f64 ? reinterpret_cast<int*>(ws + lo.off_hint) : nullptr;
    short* Jint = with_f64 ? reinterpret_cast<short*>(ws + lo.off_jint) : nullptr;
    hipLaunchKernelGGL(hquant_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, stream, H, (int)n, (int)lo.nq, prep,
                       planes, hdiag, hdiag_j, H64, Hint, Jint);
    GANQ_LAUNCH_CHECK();
    if (with_f64) {
        // change counter of the incremental bucket sums: zero once, every t_solve leaves it zero again
        GANQ_HIP_CHECK(hipMemsetAsync(ws + lo.off_chgcnt + align_up((size_t)m * sizeof(int), 8), 0, sizeof(long long), stream));
        double* WH64 = reinterpret_cast<double*>(ws + lo.off_wh64);
        double* wHw = reinterpret_cast<double*>(ws + lo.off_whw);
        const int tiles = (int)(((m + DM - 1) / DM) * ((n + DN - 1) / DN));
        if (H64) {
            hipLaunchKernelGGL(gemm_f64_kernel, dim3(tiles), dim3(256), 0, stream, W, H64, WH64, (int)m, (int)n, (int)n);
        } else {
            const WhLayout wl = wh_layout(m, n);
            int* dexp = reinterpret_cast<int*>(ws + lo.off_dexp);
            double* hdiag64 = reinterpret_cast<double*>(ws + lo.off_hdiag64);
            hipLaunchKernelGGL(dexp_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, hdiag, hdiag_j, prep, (int)n, dexp,
                               hdiag64);
            int rc = wh_gemm(W, Hint, Jint, &prep->ext, dexp, hdiag64, &prep->scale, m, n, wl, ws + lo.off_wp, ws + lo.off_hp,
                             reinterpret_cast<int*>(ws + lo.off_rexp), reinterpret_cast<int*>(ws + lo.off_wlo), WH64, stream);
            if (rc) return rc;
        }
        hipLaunchKernelGGL(whw_kernel, dim3((unsigned)m), dim3(256), 0, stream, W, WH64, (int)m, (int)n, wHw);
        GANQ_LAUNCH_CHECK();
    }
    return 0;
}

// per iteration: bucket sums (full: masks -> integer accumulation; or incremental) -> per-row solve (+ loss rows)
int t_iterate(const uint8_t* Q, int64_t m, int64_t n, int V, double rcond, const TLayout& lo, char* ws, const float* WH32,
              float* T_out, float* A_out, float* b_out, int loss_mode, double* loss_out, int iter, hipStream_t stream, bool changed_rows_only) {
    TPrep* prep = reinterpret_cast<TPrep*>(ws + lo.off_prep);
    int8_t* planes = reinterpret_cast<int8_t*>(ws + lo.off_planes);
    int* hdiag = reinterpret_cast<int*>(ws + lo.off_hdiag);
    int* hdiag_j = reinterpret_cast<int*>(ws + lo.off_hdiag_j);
    unsigned long long* bits = reinterpret_cast<unsigned long long*>(ws + lo.off_bits);
    long long* mpart = reinterpret_cast<long long*>(ws + lo.off_mpart);
    long long* mpart_lo = reinterpret_cast<long long*>(ws + lo.off_mpart_lo);
    // test switches (runtime.hip): GANQ_T_FULL=1 -> stateless full accumulation every call; GANQ_T_INCR_THR=<count> ->
    // device-side fallback threshold (changed indices per layer) instead of m*n/16
    const long long opt_thr = opt_get(OPT_T_INCR_THR);
    const bool stateful = iter >= 0 && lo.off_hint != 0 && opt_get(OPT_T_FULL) != 1;
    long long* mstate = stateful ? reinterpret_cast<long long*>(ws + lo.off_mstate) : nullptr;
    long long* mstate_lo = stateful ? reinterpret_cast<long long*>(ws + lo.off_mstate_lo) : nullptr;
    uint8_t* qprev = stateful ? reinterpret_cast<uint8_t*>(ws + lo.off_qprev) : nullptr;
    int* chgcnt = stateful ? reinterpret_cast<int*>(ws + lo.off_chgcnt) : nullptr;
    long long* changed = stateful ? reinterpret_cast<long long*>(ws + lo.off_chgcnt + align_up((size_t)m * sizeof(int), 8)) : nullptr;
    // more than 1/16 of all indices changed: the full accumulation is cheaper (decided on the device, no host sync)
    const long long thr = opt_thr >= 0 ? opt_thr : (long long)((m * n) >> 4);
    // From the second incremental iteration on the device-side fallback is not even launched (three launches that find the gate shut:
    // 14 us per iteration): the changes shrink from iteration to iteration (the incremental kernel's time on the benchmark layer:
    // 233, 135, 115, 96, 81, 60 .. us), and if a layer ever did change more than 1/16 of its indices that late, the incremental
    // update is still exact -- just slower than the full pass would have been.  (A forced threshold, GANQ_T_INCR_THR, keeps the gate.)
    const bool gated = !(stateful && iter >= 2 && opt_thr < 0);
    const long long* gate = nullptr;  // null: the full path runs unconditionally
    if (stateful && iter > 0) {
        ProfScope prof(KID_T_INCR, stream);
        uint16_t* chg = reinterpret_cast<uint16_t*>(ws + lo.off_chg);
        hipLaunchKernelGGL(q_diff_kernel, dim3((unsigned)((m + QD_WAVES - 1) / QD_WAVES)), dim3(QD_WAVES * 64), 0, stream, Q, qprev, (int)m, (int)n, chg, chgcnt,
                           changed);
        const size_t usmem = (size_t)(256 + MU_WAVES * 16 * 64) * sizeof(long long) + align_up((size_t)n, 16);
        const bool mu_lds = opt_get(OPT_MUPDATE_LDS) == 1;  // A/B switch
        const int* Hint = reinterpret_cast<const int*>(ws + lo.off_hint);
        const short* Jint = reinterpret_cast<const short*>(ws + lo.off_jint);
        if ((n & 15) == 0 && !mu_lds)
            hipLaunchKernelGGL(m_update_mfma_kernel, dim3((unsigned)m), dim3(MG_WAVES * 64), 0, stream, planes, Hint, Jint, hdiag,
                               hdiag_j, Q, qprev, (int)m, (int)n, (int)lo.nq, chg, chgcnt, mstate, mstate_lo, changed, gated ? thr : LLONG_MAX, prep);
        else {
            int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(m_update_kernel<int>), usmem);
            if (!rc) rc = ensure_dynamic_lds(reinterpret_cast<const void*>(m_update_kernel<short>), usmem);
            if (rc) return rc;
            // the extension word first (it needs the old codes; leaves at once when the extension is off), then the
            // 31-bit word, which also brings Qprev up to date
            hipLaunchKernelGGL(m_update_kernel<short>, dim3((unsigned)m), dim3(MU_WAVES * 64), usmem, stream, Jint, Q, qprev, (int)m,
                               (int)n, chg, chgcnt, mstate_lo, changed, gated ? thr : LLONG_MAX, prep);
            hipLaunchKernelGGL(m_update_kernel<int>, dim3((unsigned)m), dim3(MU_WAVES * 64), usmem, stream, Hint, Q, qprev, (int)m,
                               (int)n, chg, chgcnt, mstate, changed, gated ? thr : LLONG_MAX, prep);
        }
        GANQ_LAUNCH_CHECK();
        gate = changed;
    }
    if (gated) {
        ProfScope prof(KID_SORT_CODES, stream);
        const int64_t items = m * lo.ng;
        hipLaunchKernelGGL(code_masks_kernel, dim3((unsigned)std::min<int64_t>((items + 3) / 4, 4096)), dim3(256), 0, stream, Q, (int)m, (int)n,
                           (int)lo.ng, bits, gate, thr);
    }
    GANQ_LAUNCH_CHECK();
    const size_t smem = 2 * (size_t)btile_bytes(4) + (size_t)TW * RW * 256 * sizeof(long long);  // (the extension's two planes need less)
    {
        const int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(onehot_accum_kernel), smem);
        if (rc) return rc;
    }
    if (gated) {
        ProfScope prof(KID_SHT_ACCUM, stream);
        const int nrg = (int)((m + TR - 1) / TR);
        const bool dbg = opt_get(OPT_ACCUM_DEBUG) == 1;  // developer timing experiment
        long long* stamps = nullptr;
        if (dbg) (void)hipMalloc(&stamps, 64);
        hipLaunchKernelGGL(onehot_accum_kernel, dim3((unsigned)(2 * nrg * NP)), dim3(TW * 64), smem, stream, planes, bits, Q, (int)m, (int)n,
                           (int)lo.nq, (int)lo.ng, mpart, mpart_lo, stamps, gate, thr, prep, nrg * NP);
        if (stamps) {
            long long h[3];
            (void)hipStreamSynchronize(stream);
            (void)hipMemcpy(h, stamps, 24, hipMemcpyDeviceToHost);
            (void)hipFree(stamps);
            fprintf(stderr, "[onehot_accum stamps] wg0 wave0: prologue %lld, tile loop %lld, flush %lld cycles\n", h[0], h[1], h[2]);
        }
        if (stateful) {
            hipLaunchKernelGGL(m_reduce_kernel, dim3(1024 + 256), dim3(256), 0, stream, mpart, mpart_lo, (int)m, mstate, mstate_lo, Q, qprev,
                               m * n, gate, thr, prep, 1024);
        }
    }
    GANQ_LAUNCH_CHECK();
    // changed_rows_only (the fused loop, iterations >= 1, T_out = the previous codebooks): the per-row solve runs on the rows that
    // changed in this iteration only; the list is the one the next S-solve takes as well (t_active_rows hands it out without a
    // second launch)
    const int* rowlist = nullptr;
    const int* rowcount = nullptr;
    if (changed_rows_only && stateful && iter >= 1 && t_rows_listable(lo)) {
        int* l = reinterpret_cast<int*>(ws + lo.off_active);
        hipLaunchKernelGGL(active_rows_kernel, dim3(1), dim3(1024), 0, stream, chgcnt, (int)m, l, l + m);
        rowlist = l;
        rowcount = l + m;
    }
    {
        ProfScope prof(KID_T_SOLVE, stream);
        const dim3 grid((unsigned)((m + 3) / 4));
        const int allow_fast = opt_get(OPT_T_JACOBI) != 1;  // test switch: 1 -> every row by the Jacobi eigen-solve
        const long long* msrc = stateful ? mstate : mpart;
        const long long* msrc_lo = stateful ? mstate_lo : mpart_lo;
        const int nparts = stateful ? 0 : NP;  // 0: one part that is already symmetric
        if (WH32) {
            hipLaunchKernelGGL(t_solve_kernel<float>, grid, dim3(64), 0, stream, msrc, msrc_lo, prep, hdiag, hdiag_j, WH32,
                               static_cast<const double*>(nullptr), Q, (int)m, (int)n, V, rcond, T_out, A_out, b_out,
                               static_cast<double*>(nullptr), nparts, static_cast<long long*>(nullptr), allow_fast,
                               static_cast<const int*>(nullptr), static_cast<const int*>(nullptr));
        } else {
            const double* WH64 = reinterpret_cast<const double*>(ws + lo.off_wh64);
            const double* wHw = reinterpret_cast<const double*>(ws + lo.off_whw);
            double* loss_rows = reinterpret_cast<double*>(ws + lo.off_lossrows);
            hipLaunchKernelGGL(t_solve_kernel<double>, grid, dim3(64), 0, stream, msrc, msrc_lo, prep, hdiag, hdiag_j, WH64, wHw, Q, (int)m,
                               (int)n, V, rcond, T_out, A_out, b_out, loss_mode ? loss_rows : nullptr, nparts, changed, allow_fast, rowlist, rowcount);
            if (loss_mode == 2) hipLaunchKernelGGL(sum_rows_kernel, dim3(1), dim3(256), 0, stream, loss_rows, (int)m, loss_out);
        }
    }
    GANQ_LAUNCH_CHECK();
    return 0;
}

// after t_iterate(iter >= 1): device list / count of the rows that changed in that iteration (null when the bucket sums
// are not kept between iterations, i.e. no change lists exist)
bool t_rows_listable(const TLayout& lo) {
    // GANQ_SOLVE_ALL_ROWS=1 (test switch): never skip a row; GANQ_T_FULL=1: no change lists exist
    return lo.off_hint != 0 && lo.off_active != 0 && opt_get(OPT_T_FULL) != 1 && opt_get(OPT_SOLVE_ALL_ROWS) != 1;
}

int t_active_rows(int64_t m, const TLayout& lo, char* ws, const int** list, const int** count, hipStream_t stream, bool already_built) {
    *list = nullptr;
    *count = nullptr;
    if (!t_rows_listable(lo)) return 0;
    int* l = reinterpret_cast<int*>(ws + lo.off_active);
    int* c = l + m;
    if (!already_built)
        hipLaunchKernelGGL(active_rows_kernel, dim3(1), dim3(1024), 0, stream, reinterpret_cast<const int*>(ws + lo.off_chgcnt), (int)m, l, c);
    GANQ_LAUNCH_CHECK();
    *list = l;
    *count = c;
    return 0;
}

}  // namespace ganq

using namespace ganq;

// developer check (tests): the W @ H_fixed product of the fused driver and the fixed-point H it was formed with
extern "C" int ganq_debug_wh_product(const float* W, const float* H, int64_t m, int64_t n, double* WH_out, double* Hfixed_out,
                                     void* stream_) {
    if (m <= 0 || n <= 0 || m > INT32_MAX / 2 || n > INT32_MAX / 2) return fail(-1, "ganq_debug_wh_product: bad shape");
    if (!W || !H || !WH_out) return fail(-3, "ganq_debug_wh_product: null pointer");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const TLayout lo = t_layout(m, n, true);
    char* ws = nullptr;
    GANQ_HIP_CHECK(hipMalloc(&ws, lo.total));
    int rc = t_prepare(W, H, m, n, lo, ws, true, stream);
    if (rc == 0) {
        hipError_t e = hipMemcpyAsync(WH_out, ws + lo.off_wh64, (size_t)m * n * sizeof(double), hipMemcpyDeviceToDevice, stream);
        if (e == hipSuccess && Hfixed_out) {
            const int* Hint = reinterpret_cast<const int*>(ws + lo.off_hint);
            const short* Jint = reinterpret_cast<const short*>(ws + lo.off_jint);
            hipLaunchKernelGGL(hfixed_kernel, dim3(1024), dim3(256), 0, stream, Hint, Jint,
                               reinterpret_cast<const TPrep*>(ws + lo.off_prep), n * n, Hfixed_out);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        if (e != hipSuccess) rc = fail(-4, "ganq_debug_wh_product: %s", hipGetErrorString(e));
    }
    (void)hipFree(ws);
    return rc;
}

extern "C" size_t ganq_update_t_workspace_bytes(int64_t m, int64_t n, int V) {
    (void)V;
    if (m <= 0 || n <= 0) return 0;
    return t_layout(m, n, false).total;
}

extern "C" int ganq_update_t(const float* WH, const float* H, const uint8_t* Q, int64_t m, int64_t n, int V,
                             double rcond, float* T_out, float* A_out, float* b_out, void* workspace,
                             size_t workspace_bytes, void* stream_) {
    if (m < 0 || n < 0) return fail(-1, "ganq_update_t: negative shape");
    if (m == 0 || n == 0) return 0;
    if (V < 2 || V > 16) return fail(-2, "ganq_update_t: V=%d not supported (bits 2..4 are implemented)", V);
    if (m > INT32_MAX / 2 || n > INT32_MAX / 2) return fail(-1, "ganq_update_t: shape too large");
    if (!WH || !H || !Q || !T_out) return fail(-3, "ganq_update_t: null pointer");
    const TLayout lo = t_layout(m, n, false);
    if (!workspace || workspace_bytes < lo.total)
        return fail(-4, "ganq_update_t: workspace %zu B < required %zu B", workspace_bytes, lo.total);
    if (rcond < 0) rcond = 1.1920928955078125e-07 * (double)V;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    char* ws = static_cast<char*>(workspace);
    int rc = t_prepare(nullptr, H, m, n, lo, ws, false, stream);
    if (rc) return rc;
    return t_iterate(Q, m, n, V, rcond, lo, ws, WH, T_out, A_out, b_out, 0, nullptr, -1, stream, false);
}

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
#include <algorithm>
#include <mutex>
#include <type_traits>
#include <utility>
#include <vector>

#include "common.h"

namespace ganq {

constexpr int HT = 128;       // output tile edge
// tokens per slab (overridable for experiments).  Measured on MI355X, 2048 tokens: 32 -> 109 us at n = 4096, 326 us at
// 8192; 64 -> 138 / 360 us: the kernel is bound by the latency of a workgroup's slab chain at 2 workgroups per CU
// (528 tiles at n = 4096), longer slabs only lengthen the exposed part of each step
#ifndef HESS_HK
#define HESS_HK 32
#endif
constexpr int HK = HESS_HK;   // tokens per slab
constexpr int HL = HK / 16;   // 16-byte loads per thread, operand and slab (256 threads x 16 B = 16 tokens x 128 features)
// LDS row pitch in 16-bit elements: 320 B.  A transposed fragment read (ds_read_b64_tr_b16) is served in two groups of 32
// lanes; a group touches 4 consecutive token rows x 64 contiguous bytes (16 banks of 4 B), so it is conflict-free exactly when
// consecutive rows start 16 banks apart: pitch = 64 B (mod 256 B).  Rounds 1-2 used 272 B (rows 4 banks apart): the SQ counters
// showed 60 % of the kernel's LDS cycles as bank conflicts (SQ_LDS_BANK_CONFLICT 104 M of SQ_LDS_IDX_ACTIVE 174 M per
// 16384 x 4096 launch) and a third of the wave cycles stalled on LDS issue.
#ifndef HESS_PITCH_PAD
#define HESS_PITCH_PAD 32
#endif
constexpr int HP = HT + HESS_PITCH_PAD;

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

#ifdef GANQ_HESS_TRACE
__device__ unsigned long long g_hess_trace[8];  // developer: cycles of workgroup 0 / wave 0 in the phases of a slab step
#define HESS_T(k) do { if (blockIdx.x == 0 && tid == 0) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
        g_hess_trace[k] += now_ - last_; last_ = now_; } } while (0)
#else
#define HESS_T(k) do {} while (0)
#endif

// ---- one tile's sum over a run of tokens: acc[i][j] += X[:, u-block]^T X[:, v-block] for the token rows [0, rows) of X ------------
// (shared by the whole-tile kernel and the token-split kernels below)
// MH = 128-row blocks of the tile (1: 128 x 128, 4 waves; 2: 256 x 128, 8 waves -- 85 instead of 64 flop per byte streamed out of
// the L2s).  Xs = the workgroup's slab buffers, [buffer][image][token][feature] with images 0 .. MH-1 = the u blocks, MH = the v block.
template <bool BF16, int MH>
__device__ __forceinline__ void hess_tile_sum(uint16_t (*Xs)[HK][HP], const uint16_t* __restrict__ X, int rows, int n, int u0, int v0,
                                              f32x16 (&acc)[2][2]) {
    constexpr int NT = 256 * MH;       // threads
    constexpr int NIMG = MH + 1;
    constexpr int HLA = HL;            // 16-byte loads per thread and slab: u operand (MH x 512 pieces over NT threads)
    constexpr int HLB = HL / MH;       // ... v operand (512 pieces)
    static_assert(HL % MH == 0, "the v operand must split evenly over the threads");
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wmi = wv >> 1, wn = (wv & 1) * 64;  // this wave: rows 64 wmi .. +63 of the tile, columns wn .. +63
    auto img = [&](int buf, int i) -> uint16_t (*)[HP] { return Xs[buf * NIMG + i]; };
    // staging: slab = 32 tokens x 128 features = 512 x 16 B per image
    // four slabs of registers in rotation: the loads of slab s+3 are issued while slab s is multiplied (a slab's 64
    // MFMA-cycles are far shorter than one trip to L2 / HBM, and a tile is a chain of rows/32 such trips)
    constexpr int RD = MH == 1 ? 4 : 3;  // slabs of registers in rotation (the 8-wave shape has 128 registers per lane)
    uint4 ra4[RD][HLA], rb4[RD][HLB];
    // fast path (uniform per workgroup): full, 16-byte aligned tile columns -> unconditional loads (a token row past
    // the end is clamped and zeroed afterwards), so that the compiler can count the loads in flight instead of
    // draining them at every slab
    // (rows == 0, an empty batch, only decays H: the fast path's clamp to row rows-1 would read out of bounds)
    const bool fast = rows > 0 && (u0 + HT * MH <= n) && (v0 + HT <= n) && ((n & 7) == 0) && ((reinterpret_cast<uintptr_t>(X) & 15) == 0) &&
                      ((int64_t)(rows + 4 * HK) * n * 2 < (1ll << 31));  // 32-bit offsets of the buffer loads (incl. the look-ahead)
    // this thread's pieces of a slab: u operand piece idx = (token row, 16-byte piece of the MH x 128 features), v operand likewise
    // with 16 pieces per row; byte offsets inside a slab (a slab spans at most HK * n * 2 bytes, far below 4 GB)
    auto rowA = [&](int h) { return (h * NT + tid) / (16 * MH); };
    auto colA = [&](int h) { return ((h * NT + tid) % (16 * MH)) * 8; };
    auto rowB = [&](int h) { return (h * NT + tid) >> 4; };
    auto colB = [&](int h) { return ((h * NT + tid) & 15) * 8; };
    uint32_t offA[HLA], offB[HLB];
#pragma unroll
    for (int h = 0; h < HLA; ++h) offA[h] = (uint32_t)((rowA(h) * n + u0 + colA(h)) * 2);
#pragma unroll
    for (int h = 0; h < HLB; ++h) offB[h] = (uint32_t)((rowB(h) * n + v0 + colB(h)) * 2);
    const __amdgpu_buffer_rsrc_t rsrcX =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(X), 0, (int)((int64_t)rows * n * 2), 0x00020000);
    auto gload = [&](auto fast_tag, int t0, uint4 (&ra)[HLA], uint4 (&rb)[HLB]) {
        constexpr bool FAST = decltype(fast_tag)::value;
        if constexpr (FAST) {
            // (rows past the end are zeroed where the slab is filed, in sstore: a select HERE makes the compiler wait for the
            // load it has just issued -- vmcnt(0) right behind every pair of loads, the whole round trip exposed per slab)
            // buffer loads: the slab's byte offset in an SGPR, this thread's constant offset inside a slab in a VGPR -- no
            // per-load 64-bit address arithmetic (it was a fifth of a slab step: 610 of 2900 cycles by the stamps); the
            // resource ends with the last token row, so rows past the end read as zeros by themselves
            // Token rows past the end exist only in the last slabs (and the look-ahead behind them).  For THOSE the slab
            // offset travels in the VGPR offset, which the hardware compares with num_records in every addressing mode
            // (offset >= num_records reads 0, nothing is fetched); whether the SGPR offset takes part in that comparison
            // differs between descriptions of the gfx9 family, and this kernel does not depend on it.
            const int soff = t0 * n * 2;
            const bool past = t0 + HK > rows;  // uniform
            const int so = past ? 0 : soff, vo = past ? soff : 0;
#pragma unroll
            for (int h = 0; h < HLA; ++h)
                ra[h] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsrcX, (int)offA[h] + vo, so, 0));
#pragma unroll
            for (int h = 0; h < HLB; ++h)
                rb[h] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsrcX, (int)offB[h] + vo, so, 0));
        } else {
            auto ragged = [&](int t, int f0) -> uint4 {
                uint4 v = make_uint4(0, 0, 0, 0);
                if (t < rows) {
                    const uint16_t* pa = X + (int64_t)t * n + f0;
                    uint16_t tmp[8];
                    for (int e = 0; e < 8; ++e) tmp[e] = (f0 + e < n) ? pa[e] : (uint16_t)0;
                    v = *reinterpret_cast<uint4*>(tmp);
                }
                return v;
            };
#pragma unroll
            for (int h = 0; h < HLA; ++h) ra[h] = ragged(t0 + rowA(h), u0 + colA(h));
#pragma unroll
            for (int h = 0; h < HLB; ++h) rb[h] = ragged(t0 + rowB(h), v0 + colB(h));
        }
    };
    // t0 = first token of the slab: a slab that reaches past `rows` (uniform per workgroup: only the last ones do) has
    // its surplus token rows zeroed here
    auto sstore = [&](int buf, int t0, const uint4 (&ra)[HLA], const uint4 (&rb)[HLB]) {
        const bool tail = t0 + HK > rows;
#pragma unroll
        for (int h = 0; h < HLA; ++h) {
            uint4 va = ra[h];
            if (tail && t0 + rowA(h) >= rows) va = make_uint4(0, 0, 0, 0);
            *reinterpret_cast<uint4*>(&img(buf, MH == 1 ? 0 : (colA(h) >> 7))[rowA(h)][colA(h) & 127]) = va;
        }
#pragma unroll
        for (int h = 0; h < HLB; ++h) {
            uint4 vb = rb[h];
            if (tail && t0 + rowB(h) >= rows) vb = make_uint4(0, 0, 0, 0);
            *reinterpret_cast<uint4*>(&img(buf, MH)[rowB(h)][colB(h)]) = vb;
        }
    };

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
    auto compute = [&](int buf) {
        const uint16_t (*Sa)[HP] = img(buf, MH == 1 ? 0 : (wmi >> 1));
        const uint16_t (*Sb)[HP] = img(buf, MH);
        const int wm = (wmi & 1) * 64;
#pragma unroll
        for (int kk = 0; kk < HK; kk += 16) {
            s8v a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = frag(Sa, kk, wm + 32 * i);
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = frag(Sb, kk, wn + 32 * j);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = mfma16<BF16>(a[i], b[j], acc[i][j]);
        }
    };
    auto mainloop = [&](auto fast_tag) {
#pragma unroll
        for (int d = 0; d < RD - 1; ++d) gload(fast_tag, d * HK, ra4[d], rb4[d]);
        sstore(0, 0, ra4[0], rb4[0]);
        __syncthreads();
        // whole rounds of RD slabs; slabs past the end are zeros (they add nothing)
#ifdef GANQ_HESS_TRACE
        unsigned long long last_ = __builtin_amdgcn_s_memtime();
#endif
        for (int s = 0; s < nslab; s += RD) {
#pragma unroll
            for (int jj = 0; jj < RD; ++jj) {
                gload(fast_tag, (s + jj + RD - 1) * HK, ra4[(jj + RD - 1) % RD], rb4[(jj + RD - 1) % RD]);
                HESS_T(0);
                compute((s + jj) & 1);
                HESS_T(1);
                sstore((s + jj + 1) & 1, (s + jj + 1) * HK, ra4[(jj + 1) % RD], rb4[(jj + 1) % RD]);
                HESS_T(2);
                __syncthreads();
                HESS_T(3);
            }
        }
    };
    if (fast) mainloop(std::true_type{});
    else mainloop(std::false_type{});
}

// ---- a finished tile: H <- H * decay + scale * acc on the tile and, off the diagonal, on its mirror image ------------------------------
// mode 0: a tile below the diagonal (direct + mirror); 1: a square tile ON the diagonal, computed in full (X^T X is symmetric), so
// it needs no mirror; 2: a 256 x 128 tile that crosses the diagonal: element (u, v) is written directly where u >= v and mirrored
// where u > v, so that every element of H is written exactly once (the read-modify-write with the decay must not happen twice)
__device__ __forceinline__ void hess_tile_finish(float* __restrict__ H, int n, int u0, int v0, int mode, float decay, float scale,
                                                 f32x16 (&acc)[2][2], char* lds_scratch) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wm = (wv >> 1) * 64, wn = (wv & 1) * 64;
    const bool pred = mode == 2;
    // direct tile: H[u][v], lanes along v (128 B runs), read-modify-write with the running-average decay
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int u = u0 + wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int v = v0 + wn + 32 * j + (lane & 31);
                float val = 0.0f;
                if (u < n && v < n && (!pred || u >= v)) {
                    const int64_t o = (int64_t)u * n + v;
                    const float old = (decay != 0.0f) ? H[o] * decay : 0.0f;
                    val = old + scale * acc[i][j][r];
                    H[o] = val;
                }
                acc[i][j][r] = val;
            }
    if (mode == 1) return;
    // mirror tile H[v][u]: transposed through LDS (the slab buffers are free now) so that it is written in 128 B
    // runs as well -- as 4-byte scattered stores it cost more than everything else in the kernel together
    __syncthreads();
    // (one 32 x 32 block of the wave's 64 x 64 at a time: 4.2 KB of scratch per wave, so that eight waves stay inside the slab buffers)
    float(*Tr)[33] = reinterpret_cast<float(*)[33]>(lds_scratch + wv * (32 * 33 * sizeof(float)));
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int r = 0; r < 16; ++r) Tr[(r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)][lane & 31] = acc[i][j][r];
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const int rr = lane & 31;
            const int u = u0 + wm + 32 * i + rr;
            for (int cc = lane >> 5; cc < 32; cc += 2) {
                const int v = v0 + wn + 32 * j + cc;
                if (u < n && v < n && (!pred || u > v)) H[(int64_t)v * n + u] = Tr[rr][cc];
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// tile shape MH: workgroup threads, floats of a partial tile, LDS bytes (slab buffers; the finish's transpose scratch, 4224 B per wave)
template <int MH> struct HessShape {
    static constexpr int NT = 256 * MH;
    static constexpr int PART_FLOATS = HT * HT * MH;
    static constexpr size_t SLAB_BYTES = (size_t)2 * (MH + 1) * HK * HP * sizeof(uint16_t);
    static constexpr size_t SCRATCH_BYTES = (size_t)4 * MH * 32 * 33 * sizeof(float);
    static constexpr size_t LDS_BYTES = SLAB_BYTES > SCRATCH_BYTES ? SLAB_BYTES : SCRATCH_BYTES;
    static constexpr int WGS_PER_CU = MH == 1 ? 3 : 2;
};
// tile table entry -> first row / column and finish mode
template <int MH>
__device__ __forceinline__ void hess_tile_of(uint32_t pr, int& u0, int& v0, int& mode) {
    const int tu = (int)(pr >> 16), tv = (int)(pr & 0xffffu);
    u0 = tu * HT * MH;
    v0 = tv * HT;
    if (MH == 1) mode = tu == tv ? 1 : 0;
    else mode = v0 + HT > u0 ? 2 : 0;  // some column of the tile lies right of its first row: it crosses the diagonal
}

template <bool BF16>
__global__ __launch_bounds__(256, 3) void hessian_kernel(float* __restrict__ H, const uint16_t* __restrict__ X, int rows,
                                                      int n, float decay, float scale, int tiles_per_side,
                                                      const uint32_t* __restrict__ tile_order) {
    __shared__ __align__(16) uint16_t Xs[2][2][HK][HP];  // [buffer][operand][token][feature]; reused by the epilogue

    // blockIdx.x enumerates the lower-triangular tile pairs (tu >= tv) through a table in Z (Morton) order, and the
    // workgroups an XCD receives (every 8th) are mapped to one contiguous run of it: the 32 CUs behind one L2 then work
    // on a compact patch of H that needs ~16 of the 32 column blocks of X instead of all of them, which keeps the
    // token slabs they stream L2-resident even when the workgroups drift apart (PMC: 62 % L2 hits, 230 MB fetched per
    // 2048-token batch of a 16 MB X before this).
    int tu = 0, tv = 0;
    {
        int b = blockIdx.x;
        const int nblk = (int)gridDim.x;
        if ((nblk & 7) == 0) b = (b & 7) * (nblk >> 3) + (b >> 3);
        const uint32_t pr = tile_order[b];
        tu = (int)(pr >> 16);
        tv = (int)(pr & 0xffffu);
    }
    (void)tiles_per_side;
    const int u0 = tu * HT, v0 = tv * HT;
    static_assert(HessShape<1>::SCRATCH_BYTES <= sizeof(Xs), "transpose scratch must fit the slab buffers");

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    hess_tile_sum<BF16, 1>(&Xs[0][0], X, rows, n, u0, v0, acc);
    hess_tile_finish(H, n, u0, v0, tu == tv ? 1 : 0, decay, scale, acc, reinterpret_cast<char*>(&Xs[0][0][0][0]));
}

// ---- token-split launches (round 3) ---------------------------------------------------------------------------------------------
// The whole-tile kernel gives a workgroup one 128 x 128 tile of H and all the tokens.  That fills the chip only when there are
// several tiles per workgroup slot (3 per CU by registers and LDS: 768): at n = 4096 the 528 tiles leave 16 CUs with three
// workgroups and the rest with two -- the launch lasts as long as the CUs with three --, at n = 3072 there are 300 tiles, at
// n = 2048 136 for 256 CUs, at n = 768 21.  Here the first `bulk` tiles of the Z curve (a multiple of the CU count) keep one
// workgroup each and all the tokens; every other tile is cut into `parts` equal token ranges, one workgroup each, which fill
// the remaining slots.  All tiles are cut at the SAME tokens and workgroups of one part are neighbours in the grid: like the
// whole tiles they walk the tokens in step and share the slabs of X they stream through the L2s (cutting the linear
// (tile, slab) space into equal runs, every workgroup at its own token offset, was measured first: 643 us against 420 for whole
// tiles at 16384 x 4096 -- every tile then streams its 8 MB of X from HBM by itself).  A part is stored as a partial tile (the
// accumulators in register order, coalesced); hessian_fix_kernel adds a tile's parts IN TOKEN ORDER and finishes it: no atomics,
// no waiting between workgroups, the same bits whatever the scheduling.
template <bool BF16, int MH>
__global__ __launch_bounds__(256 * MH, MH == 1 ? 3 : 4) void hessian_sk_kernel(float* __restrict__ H, const uint16_t* __restrict__ X, int rows,
                                                                            int n, float decay, float scale,
                                                                            const uint32_t* __restrict__ tile_order, int bulk, int rest,
                                                                            int part_slabs, float* __restrict__ partial) {
    using Shape = HessShape<MH>;
    __shared__ __align__(16) char smem[Shape::LDS_BYTES];
    uint16_t(*Xs)[HK][HP] = reinterpret_cast<uint16_t(*)[HK][HP]>(smem);
    int w = blockIdx.x, ti, t_begin = 0, t_end = rows, slot = -1;
    if (w < bulk) {
        // the workgroups an XCD receives (every 8th) take one contiguous run of the Z curve, as in the whole-tile kernel
        if ((bulk & 7) == 0) w = (w & 7) * (bulk >> 3) + (w >> 3);
        ti = w;
    } else {
        slot = w - bulk;  // part-major: the workgroups of one part (one token range) follow each other
        const int part = slot / rest;
        ti = bulk + slot % rest;
        t_begin = min(rows, part * part_slabs * HK);
        t_end = min(rows, (part + 1) * part_slabs * HK);
    }
    int u0, v0, mode;
    hess_tile_of<MH>(tile_order[ti], u0, v0, mode);
    const int tid = threadIdx.x;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    if (t_end > t_begin) hess_tile_sum<BF16, MH>(Xs, X + (int64_t)t_begin * n, t_end - t_begin, n, u0, v0, acc);
    if (slot < 0) {
        hess_tile_finish(H, n, u0, v0, mode, decay, scale, acc, smem);
    } else {
        float* dst = partial + (int64_t)slot * Shape::PART_FLOATS;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) dst[((i * 2 + j) * 16 + r) * Shape::NT + tid] = acc[i][j][r];
    }
}

// The parts of the cut tiles, summed IN TOKEN ORDER by the whole chip (round 4): rest x parts partial tiles of 64 / 128 KB -- 32 MB
// at n = 4096 -- used to be read by `rest` = 16 workgroups, one per tile, part after part: 95 us of a 447 us group of 16384 tokens
// (rocprofv3, round 3).  Here every thread owns 8 floats of one tile (two float4 of its 2048-float slice) and adds the parts
// p = 1, 2, .. onto part 0 in place; hessian_fix_kernel then finishes the tile from that one part.  Same additions in the same
// order ((p0 + p1) + p2 ...), one element = one thread: the same bits.
constexpr int PRESUM_SLICE = 2048;  // floats per workgroup: 256 threads x 2 float4
template <int MH>
__global__ __launch_bounds__(256) void hessian_presum_kernel(int rest, int parts, float* __restrict__ partial) {
    using Shape = HessShape<MH>;
    constexpr int SLICES = Shape::PART_FLOATS / PRESUM_SLICE;
    const int j = (int)blockIdx.x / SLICES, sl = (int)blockIdx.x % SLICES;
    float4* base = reinterpret_cast<float4*>(partial + (int64_t)j * Shape::PART_FLOATS + (int64_t)sl * PRESUM_SLICE) + threadIdx.x;
    const int64_t stride4 = (int64_t)rest * Shape::PART_FLOATS / 4;  // one part further (float4 units)
    float4 a0 = base[0], a1 = base[256];
    for (int p = 1; p < parts; ++p) {
        const float4 b0 = base[p * stride4], b1 = base[p * stride4 + 256];
        a0.x += b0.x; a0.y += b0.y; a0.z += b0.z; a0.w += b0.w;
        a1.x += b1.x; a1.y += b1.y; a1.z += b1.z; a1.w += b1.w;
    }
    base[0] = a0;
    base[256] = a1;
}

// one workgroup per cut tile: the sum of its parts in token order, then the same finish as everywhere
template <int MH>
__global__ __launch_bounds__(256 * MH) void hessian_fix_kernel(float* __restrict__ H, int n, float decay, float scale,
                                                              const uint32_t* __restrict__ tile_order, int bulk, int rest, int parts,
                                                              const float* __restrict__ partial) {
    using Shape = HessShape<MH>;
    __shared__ __align__(16) char scratch[Shape::SCRATCH_BYTES];
    const int j = blockIdx.x, tid = threadIdx.x;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][jj][r] = 0.0f;
    for (int part = 0; part < parts; ++part) {
        const float* src = partial + ((int64_t)part * rest + j) * Shape::PART_FLOATS;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][jj][r] += src[((i * 2 + jj) * 16 + r) * Shape::NT + tid];
    }
    int u0, v0, mode;
    hess_tile_of<MH>(tile_order[bulk + j], u0, v0, mode);
    hess_tile_finish(H, n, u0, v0, mode, decay, scale, acc, scratch);
}

}  // namespace ganq

using namespace ganq;

namespace {
struct TileTable {
    int device, tiles, mh, count;
    uint32_t* dev;
};
std::vector<TileTable> g_tables;
std::mutex g_tables_mu;

uint32_t morton2(uint32_t x, uint32_t y) {
    auto spread = [](uint32_t v) {
        v &= 0xffffu;
        v = (v | (v << 8)) & 0x00ff00ffu;
        v = (v | (v << 4)) & 0x0f0f0f0fu;
        v = (v | (v << 2)) & 0x33333333u;
        v = (v | (v << 1)) & 0x55555555u;
        return v;
    };
    return spread(x) | (spread(y) << 1);
}

// lower-triangular tile pairs (tu << 16 | tv) sorted along the Z curve; cached per device, tile count and tile height
// (mh = 1: 128 x 128 tiles, tv <= tu; mh = 2: 256 rows x 128 columns, every tile with an element on or below the diagonal:
// tv <= 2 tu + 1).  `tiles` = 128-column blocks per side; *count = tiles in the table.
const uint32_t* tile_table(int tiles, int mh, int* count) {
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lk(g_tables_mu);
    for (const TileTable& t : g_tables)
        if (t.device == device && t.tiles == tiles && t.mh == mh) {
            *count = t.count;
            return t.dev;
        }
    std::vector<std::pair<uint32_t, uint32_t>> keyed;
    const int rows_blocks = (tiles + mh - 1) / mh;
    for (int tu = 0; tu < rows_blocks; ++tu)
        for (int tv = 0; tv < tiles && tv <= mh * tu + (mh - 1); ++tv)
            keyed.push_back({morton2((uint32_t)tv, (uint32_t)(mh * tu)), ((uint32_t)tu << 16) | (uint32_t)tv});
    std::sort(keyed.begin(), keyed.end());
    std::vector<uint32_t> host(keyed.size());
    for (size_t i = 0; i < keyed.size(); ++i) host[i] = keyed[i].second;
    uint32_t* dev = nullptr;
    if (hipMalloc(&dev, host.size() * sizeof(uint32_t)) != hipSuccess) return nullptr;
    // synchronous copy on purpose (pageable host memory, once per shape)
    if (hipMemcpy(dev, host.data(), host.size() * sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    g_tables.push_back({device, tiles, mh, (int)host.size(), dev});
    *count = (int)host.size();
    return dev;
}

}  // namespace
// (for hessian_w4.hip: the same table for its 256 x 256 tiles -- 32 consecutive entries of the Z curve are a 4 x 8 / 8 x 4 block)
namespace ganq {
const uint32_t* hessian_tile_table(int tiles, int* count) { return tile_table(tiles, 1, count); }
}  // namespace ganq
namespace {
// scratch of the token-split launches (partial tiles): caller-owned, see ganq_hessian_workspace_bytes
inline size_t sk_scratch_bytes(int ncu) { return (size_t)6 * (size_t)ncu * HT * HT * sizeof(float); }  // 6 ncu x 64 KB = 3 ncu x 128 KB (96 MB)
}  // namespace

#ifdef GANQ_HESS_TRACE
extern "C" int ganq_debug_hess_trace(unsigned long long* out8) {
    GANQ_HIP_CHECK(hipMemcpyFromSymbol(out8, HIP_SYMBOL(ganq::g_hess_trace), 8 * sizeof(unsigned long long)));
    unsigned long long z[8] = {};
    GANQ_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(ganq::g_hess_trace), z, sizeof(z)));
    return 0;
}
#endif

extern "C" size_t ganq_hessian_workspace_bytes(int64_t rows, int64_t n) {
    // only staged groups (>= 128 slabs of tokens) on layers whose tiles are all on the fast path are ever cut
    if (rows <= 0 || n <= 0 || (n % HT) != 0 || (rows + HK - 1) / HK < 128) return 0;
    const int ncu = current_device_cus();
    return ncu >= 8 ? sk_scratch_bytes(ncu) : 0;
}

extern "C" int ganq_hessian_accum(float* H, const void* X, int dtype, int64_t rows, int64_t n, int64_t nsamples_before,
                                  int64_t batch, void* workspace, size_t workspace_bytes, void* stream_) {
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
    int blocks = 0;
    const uint32_t* order = tile_table(tiles, 1, &blocks);
    if (!order) return fail(-100, "ganq_hessian_accum: could not build the tile table");
    ProfScope prof(KID_HESSIAN, stream);
    // The kernel's fast path addresses X through 32-bit buffer offsets: a batch of more than ~2 GB goes in pieces of whole
    // slabs -- the first with the batch's decay, the others adding to it (decay 1) with the same scale; the same sum in the
    // same token order.
    // (in_features beyond ~6.7 M would make a piece a single slab and every piece a read-modify-write of H; no layer is near)
    const int64_t piece_max = std::max<int64_t>(HK, (((int64_t)1 << 31) / (2 * n) - 4 * HK - 1) / HK * HK);
    const uint16_t* Xp = static_cast<const uint16_t*>(X);
    // Staged groups of batches (>= 128 slabs) on layers whose every tile is on the kernels' fast path go through
    // hessian_sk_kernel: tiles beyond a multiple of the CU count have their tokens cut into parts that fill the remaining
    // workgroup slots (at least 32 slabs per part, at most 40 parts per tile: the fix-up reads a tile's parts one after the
    // other), and -- GANQ_HESS_WIDE -- the tiles are 256 x 128 where in_features allows.  Measured, 16384 tokens: see DESIGN.md.
    // A single sequence of 2048 tokens loses 10-30 % to a second launch and stays on the whole-tile kernel.
    const int ncu = current_device_cus();
    const int64_t nslab = (rows + HK - 1) / HK;
    const bool sk_ok = rows > 0 && rows <= piece_max && (n % HT) == 0 && (reinterpret_cast<uintptr_t>(X) & 15) == 0 && ncu >= 8 && nslab >= 128;
    auto launch_sk = [&](auto mh_tag, bool whole_only) -> int {  // 1: launched, 0: not applicable, < 0: error
        constexpr int MH = decltype(mh_tag)::value;
        using Shape = HessShape<MH>;
        int count = 0;
        const uint32_t* ord = tile_table(tiles, MH, &count);
        if (!ord) return fail(-100, "ganq_hessian_accum: could not build the tile table");
        const int slots = Shape::WGS_PER_CU * ncu;
        int bulk = count, rest = 0, parts = 1, part_slabs = (int)nslab;
        float* part = nullptr;
        if (opt_get(OPT_HESS_SPLIT) != 0 && count % ncu != 0) {
            // the tiles beyond a multiple of the CU count are cut: into the slots the bulk leaves free when everything is resident
            // at once, else into one round of slots behind the bulk's rounds
            bulk = count / ncu * ncu;
            if (opt_get(OPT_HESS_BULK) >= 0) bulk = (int)std::min<long long>(bulk, opt_get(OPT_HESS_BULK) / ncu * ncu);  // developer
            rest = count - bulk;
            const int free_slots = count < slots ? slots - bulk : slots;
            parts = (int)std::min<int64_t>(std::min<int64_t>(free_slots / rest, nslab / 32), 40);
            if (opt_get(OPT_HESS_SPLIT) > 1) parts = (int)std::min<long long>(parts, opt_get(OPT_HESS_SPLIT));  // developer: cap the parts
            if (opt_get(OPT_HESS_PARTS) > 0) parts = (int)std::min<long long>(opt_get(OPT_HESS_PARTS), nslab / 4);  // developer: force them
            if (parts >= 2) {
                part_slabs = (int)(((nslab + parts - 1) / parts + 3) / 4 * 4);  // whole rounds of four slabs
                parts = (int)((nslab + part_slabs - 1) / part_slabs);           // (no empty parts)
                if ((int64_t)rest * parts <= (int64_t)6 * ncu / MH)             // what the scratch holds
                    part = (workspace && workspace_bytes >= sk_scratch_bytes(ncu)) ? static_cast<float*>(workspace) : nullptr;
            }
            if (parts < 2 || !part) {
                bulk = count;
                rest = 0;
                parts = 1;
            }
        }
        if (rest == 0 && !whole_only) return 0;  // nothing to cut: the caller's whole-tile path
        const unsigned grid = (unsigned)(bulk + rest * parts);
        if (dtype == 1)
            hipLaunchKernelGGL((hessian_sk_kernel<true, MH>), dim3(grid), dim3(Shape::NT), 0, stream, H, Xp, (int)rows, (int)n, decay, scale, ord,
                               bulk, rest, part_slabs, part);
        else
            hipLaunchKernelGGL((hessian_sk_kernel<false, MH>), dim3(grid), dim3(Shape::NT), 0, stream, H, Xp, (int)rows, (int)n, decay, scale, ord,
                               bulk, rest, part_slabs, part);
        if (rest > 0) {
            int fix_parts = parts;
            if (parts > 2) {  // the parts summed by the whole chip first (hessian_presum_kernel)
                hipLaunchKernelGGL(hessian_presum_kernel<MH>, dim3((unsigned)(rest * (Shape::PART_FLOATS / PRESUM_SLICE))), dim3(256), 0, stream, rest,
                                   parts, part);
                fix_parts = 1;
            }
            hipLaunchKernelGGL(hessian_fix_kernel<MH>, dim3((unsigned)rest), dim3(Shape::NT), 0, stream, H, (int)n, decay, scale, ord, bulk, rest,
                               fix_parts, part);
        }
        GANQ_LAUNCH_CHECK();
        return 1;
    };
    if (sk_ok) {
        // 256 x 128 tiles from 3072 in_features on (measured, 16384 tokens: n = 3072 284 -> 233 us, 14336 5720 -> 3815; below, the
        // few wide tiles leave CUs with one 8-wave workgroup: n = 2048 136 vs 145 us, 768 81 vs 91); GANQ_HESS_WIDE=2 forces them
        const long long wide = opt_get(OPT_HESS_WIDE);
        int rc = 0;
        if (wide != 0 && (n % (2 * HT)) == 0 && (wide == 2 || n >= 3072)) rc = launch_sk(std::integral_constant<int, 2>{}, true);
        else if (opt_get(OPT_HESS_SPLIT) != 0) rc = launch_sk(std::integral_constant<int, 1>{}, false);
        if (rc != 0) return rc < 0 ? rc : 0;
    }
    int64_t done = 0;
    do {
        const int64_t piece = std::min(rows - done, piece_max);
        const float dec = done == 0 ? decay : 1.0f;
        if (dtype == 1)
            hipLaunchKernelGGL(hessian_kernel<true>, dim3(blocks), dim3(256), 0, stream, H, Xp + done * n, (int)piece, (int)n, dec, scale,
                               tiles, order);
        else
            hipLaunchKernelGGL(hessian_kernel<false>, dim3(blocks), dim3(256), 0, stream, H, Xp + done * n, (int)piece, (int)n, dec, scale,
                               tiles, order);
        GANQ_LAUNCH_CHECK();
        done += piece;
    } while (done < rows);
    return 0;
}

/* ---- the transposed staging path (hessian_w4.hip): the host stages the calibration batches of a group as Xt [in_features][ldt tokens]
 * (ganq_hessian_stage_t per batch: the copy it made anyway, now transposing) and hands the group over once. ---- */
extern "C" int ganq_hessian_t_supported(int64_t n, int64_t ldt) { return hessian_w4_supported(n, ldt) ? 1 : 0; }

extern "C" size_t ganq_hessian_t_workspace_bytes(int64_t n) { return n > 0 ? hessian_w4_workspace_bytes(n) : 0; }

extern "C" int ganq_hessian_stage_t(void* Xt, int64_t ldt, const void* X, int64_t rows, int64_t n, int64_t tok0, void* stream_) {
    if (rows < 0 || n <= 0 || tok0 < 0 || ldt <= 0) return fail(-1, "ganq_hessian_stage_t: bad sizes");
    if (!Xt || (!X && rows > 0)) return fail(-3, "ganq_hessian_stage_t: null pointer");
    return hessian_w4_stage(Xt, ldt, X, rows, n, tok0, static_cast<hipStream_t>(stream_));
}

extern "C" int ganq_hessian_accum_t(float* H, const void* Xt, int64_t ldt, int dtype, int64_t rows, int64_t n, int64_t nsamples_before,
                                    int64_t batch, void* workspace, size_t workspace_bytes, void* stream_) {
    if (rows < 0 || n < 0 || nsamples_before < 0 || batch <= 0) return fail(-1, "ganq_hessian_accum_t: bad sizes");
    if (n == 0 || rows == 0) return rows == 0 && n > 0 ? fail(-1, "ganq_hessian_accum_t: an empty group") : 0;
    if (dtype != 0 && dtype != 1) return fail(-2, "ganq_hessian_accum_t: dtype %d (0 = fp16, 1 = bf16)", dtype);
    if (!H || !Xt) return fail(-3, "ganq_hessian_accum_t: null pointer");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const double total = (double)(nsamples_before + batch);
    const float decay = (float)((double)nsamples_before / total);
    const float scale = (float)(2.0 / total);
    ProfScope prof(KID_HESSIAN, stream);
    return hessian_w4(H, Xt, ldt, dtype, rows, n, decay, scale, workspace, workspace_bytes, stream);
}

// Dense fp16 / bf16 GEMM for the prefill side of the LUT forward (round 4):
//     y[M, N] = x[M, K] @ Wd[N, K]^T (+ bias) (+ addend)          fp32 accumulation, one rounding
// Wd is the layer's weight dequantised ONCE per call into the caller's workspace (lut_dequant_kernel, lut_linear.hip); from
// M >= ~1024 rows on, decoding the 4-bit stream inside the GEMM (lut_gemm.hip) costs more than writing 2 bytes per weight once:
// every 256-row tile of y re-decodes the same weights, and the decode's LDS lookups and packs sit in the issue slots of the
// matrix instructions.  No library GEMM: this is the hand-written CDNA4 structure for large tiles --
//   * one workgroup = a BM x 256 tile of y (BM = 256 or 128), 8 waves as 2 (rows) x 4 (features), wave tile (BM/2) x 64,
//     v_mfma_f32_16x16x32_{f16,bf16} with the weight fragment as the A operand and the activation fragment as B, so that a lane
//     ends up with 4 consecutive FEATURES of one row (8-byte stores);
//   * both operands go global -> LDS by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write), in full 128-byte
//     lines: one wave instruction = 8 rows x 128 B, written lane-linear; the 16-byte slot a lane FETCHES is XOR-swizzled with
//     (row >> 1) & 7 and the fragment reads apply the same XOR, so the ds_read_b128 of 16 consecutive rows (a 16-lane group)
//     covers all 64 banks once -- conflict-free without padding (padding would break the lane-linear DMA image);
//   * two LDS buffers per operand and a K tile of 64 split into 4 phases (one quadrant of the wave tile each: 16 or 8 matrix
//     instructions between two raw s_barriers).  Each phase issues the LDS-DMA of ONE half operand tile of a later K tile;
//     the loads stay in flight ACROSS the barriers -- the only waits are two counted s_waitcnt vmcnt(6) / vmcnt(4) per K tile
//     (never 0 in the steady state), each placed a phase before the first read of what it retires.  Hazards: a half buffer is
//     re-filled at the earliest in the phase after the one whose reads (retired by that phase's lgkmcnt(0) in front of its
//     barrier) were the last of the previous tile;
//   * the two wave rows run ONE BARRIER APART, so that on every SIMD one wave issues matrix instructions while the other
//     reads its fragments (v1, both rows in lockstep: 1.03 PF at 4096^3; see the kernel).
#include "common.h"
#include "mfma_h16.h"

#include <algorithm>
#include <type_traits>

namespace ganq {

namespace {

typedef _Float16 hg_f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 hg_bf16x8 __attribute__((ext_vector_type(8)));

constexpr int HBK = 64;    // in_features per K tile: one 128-byte line per row
constexpr int HBN = 256;   // output features per workgroup
constexpr int HTHREADS = 512;

template <int BM>
constexpr size_t hg_lds_bytes() { return 2 * (size_t)(BM + HBN) * HBK * 2; }  // 2 buffers x (x tile + weight tile)

template <bool BF16>
__device__ __forceinline__ hg_f32x4 hg_mfma(hg_u32x4 a, hg_u32x4 b, hg_f32x4 c) {
    if constexpr (BF16)
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(hg_bf16x8, a), __builtin_bit_cast(hg_bf16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(hg_f16x8, a), __builtin_bit_cast(hg_f16x8, b), c, 0, 0, 0);
}

// Tile of workgroup number `bid` (after the per-L2 renumbering: the 32 CUs behind one L2 hold consecutive numbers): panels of HG_PANEL
// rows of tiles, walked column by column, so that the workgroups resident behind one L2 cover a 4 x 8 block of tiles (12 operand
// panels fetched for 32 tiles) instead of a 1 x 32 or 2 x 16 strip (33 or 18): less traffic past the L2, which on this part is
// power as much as time -- the dense kernels run against the power limit, not the issue rate (tools/dev/mfma_peak.hip).
// (128-row tiles: 8 x 4 blocks -- the block whose operand panels are smallest, rows x BM + columns x 256)
template <int HG_PANEL>
__device__ __forceinline__ void hg_tile_of(int bid, int tiles_m, int tiles_n, int& tm, int& tn) {
    const int per_panel = HG_PANEL * tiles_n;
    const int p = bid / per_panel, idx = bid - p * per_panel;
    const int ph = min(HG_PANEL, tiles_m - p * HG_PANEL);  // rows of tiles in this panel (the last one may be short)
    tn = idx / ph;
    tm = p * HG_PANEL + (idx - tn * ph);
}

template <bool BF16, int BM>
__global__ __launch_bounds__(HTHREADS, 1) void gemm_h16_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ w,
                                                              const uint16_t* __restrict__ bias, const float* __restrict__ addend,
                                                              int M, int N, int K, int tiles_m, int tiles_n, uint16_t* __restrict__ y) {
    constexpr int MT = BM / 32;        // 16-row tiles of the wave tile (BM / 2 rows): 8 or 4
    constexpr int MH = MT / 2;         // ... per phase (one half of the wave's rows)
    constexpr int GLA = BM / 128;      // LDS-DMA instructions per thread for one HALF of the x tile (BM / 2 rows x 128 B)
    constexpr int GLB = 2;             // ... for one half of the weight tile (128 rows)
    constexpr int A_BYTES = BM * HBK * 2, B_BYTES = HBN * HBK * 2, BUF = A_BYTES + B_BYTES;
    extern __shared__ __align__(1024) char hg_smem[];  // [2][x tile | weight tile], rows of 128 B, slots XOR-swizzled

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wr = wv >> 2, wc = wv & 3;

    // tile of this workgroup: the workgroups behind one L2 (blockIdx % 8 by the hardware's round-robin) walk consecutive tiles
    // of a row of tiles -- they share the x tile and stream neighbouring weight tiles
    const int nwg = tiles_m * tiles_n;
    int bid = (int)blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;  // bijective also when nwg % 8 != 0
    }
    int tm, tn;
    hg_tile_of<(BM == 128 ? 8 : 4)>(bid, tiles_m, tiles_n, tm, tn);
    const int row0 = tm * BM, f0 = tn * HBN;
    const int nt = K / HBK;

    // ---- LDS-DMA sources: lane -> (row of the half tile, 16-byte slot); the slot FETCHED is the slot WRITTEN ^ ((row >> 1) & 7)
    // chunk c = 8 rows x 128 B = one wave instruction; a half tile of R rows has R / 8 chunks, dealt wave-major
    // (uniform base pointer + 32-bit lane offset: the instruction's saddr + voffset form -- the K tile advances the scalar base)
    uint32_t aoff[2][GLA], boff[2][GLB];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int i = 0; i < GLA; ++i) {
            const int c = wv * GLA + i;                       // chunk of the half tile (BM / 16 chunks)
            const int rl = h * (BM / 2) + c * 8 + (lane >> 3);  // row of the tile
            const int g = (lane & 7) ^ ((rl >> 1) & 7);
            const int row = min(row0 + rl, M - 1);
            aoff[h][i] = (uint32_t)(((size_t)row * K + g * 8) * 2);
        }
#pragma unroll
        for (int i = 0; i < GLB; ++i) {
            const int c = wv * GLB + i;
            const int rl = h * 128 + c * 8 + (lane >> 3);
            const int g = (lane & 7) ^ ((rl >> 1) & 7);
            const int row = min(f0 + rl, N - 1);
            boff[h][i] = (uint32_t)(((size_t)row * K + g * 8) * 2);
        }
    }
    const char* const xb = reinterpret_cast<const char*>(x);
    const char* const wb = reinterpret_cast<const char*>(w);
    auto stage_a = [&](int h, int t) {  // half h of the x tile of K tile t into buffer t & 1
        char* base = hg_smem + (t & 1) * BUF + h * (A_BYTES / 2) + wv * GLA * 1024;
        const char* src = xb + (size_t)t * (HBK * 2);
#pragma unroll
        for (int i = 0; i < GLA; ++i)
            __builtin_amdgcn_global_load_lds((hg_gptr)(src + aoff[h][i]), (hg_lptr)(base + i * 1024), 16, 0, 0);
    };
    auto stage_b = [&](int h, int t) {
        char* base = hg_smem + (t & 1) * BUF + A_BYTES + h * (B_BYTES / 2) + wv * GLB * 1024;
        const char* src = wb + (size_t)t * (HBK * 2);
#pragma unroll
        for (int i = 0; i < GLB; ++i)
            __builtin_amdgcn_global_load_lds((hg_gptr)(src + boff[h][i]), (hg_lptr)(base + i * 1024), 16, 0, 0);
    };

    // ---- fragment reads: lane l holds row (l & 15) of a 16-row tile, k group (l >> 4) of a 32-wide half of the K tile
    const uint32_t lds0 = (uint32_t)(uintptr_t)hg_smem;
    const uint32_t fr_off = (uint32_t)((lane & 15) * 128 + (((lane >> 4) ^ ((lane & 15) >> 1)) << 4));  // k half 1: ^ 64
    const uint32_t a_base = lds0 + (uint32_t)(wr * (BM / 2) * 128) + fr_off;
    const uint32_t b_base = lds0 + (uint32_t)A_BYTES + (uint32_t)(wc * 64 * 128) + fr_off;
    auto lds_read = [&](uint32_t addr) -> hg_u32x4 {
        typedef const hg_u32x4 __attribute__((address_space(3))) * lp;
        return *reinterpret_cast<lp>(addr);
    };

    hg_f32x4 acc[MT][4];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = hg_f32x4{0.f, 0.f, 0.f, 0.f};
    // fragments: [16-row tile of the half][k half]; weight features 0..31 of the wave tile, double-buffered over the K tiles (the
    // next tile's are read while this tile's last quadrant still uses the old ones); features 32..63
    hg_u32x4 af[MH][2], bf0[2][2][2], bf1[2][2];

    // ---- prologue: K tile 0 complete, the weight halves of tile 1 in flight
    stage_a(0, 0);
    stage_a(1, 0);
    stage_b(0, 0);
    stage_b(1, 0);
    if (nt > 1) {
        stage_b(0, 1);
        stage_b(1, 1);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) bf0[0][j][kh] = lds_read((b_base + j * 2048) ^ (kh * 64));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    // The two wave rows run one barrier apart: while the waves of row 0 issue the matrix instructions of a phase, those of row 1
    // read the fragments of theirs (and the other way round in the next epoch), so the LDS and the matrix pipe of a SIMD -- one
    // wave of each row -- work at the same time.  Every phase is  { reads + LDS-DMA issue; waits; BARRIER; matrix instructions;
    // BARRIER }; the reads of a phase are retired (lgkmcnt(0)) BEFORE its first barrier, so a buffer whose last reads were issued
    // in epoch e by the late row is free for the early row's LDS-DMA in epoch e + 1.
    const bool late = __builtin_amdgcn_readfirstlane(wr) != 0;
    if (late) __builtin_amdgcn_s_barrier();

#define HG_MMA(ACC_I0, ACC_J0, BFR)                                                                                   \
    do {                                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                            \
        __builtin_amdgcn_s_setprio(1);                                                                                \
        _Pragma("unroll") for (int kh = 0; kh < 2; ++kh)                                                              \
            _Pragma("unroll") for (int i = 0; i < MH; ++i)                                                            \
                _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                         \
                    acc[(ACC_I0) + i][(ACC_J0) + j] = hg_mfma<BF16>(BFR[j][kh], af[i][kh], acc[(ACC_I0) + i][(ACC_J0) + j]); \
        __builtin_amdgcn_s_setprio(0);                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                            \
    } while (0)

    auto ktile = [&](auto par_tag, const int t) {
        constexpr int PAR = decltype(par_tag)::value;
        const uint32_t ab = a_base + (uint32_t)(PAR * BUF), bb = b_base + (uint32_t)(PAR * BUF);
        const uint32_t bn = b_base + (uint32_t)((PAR ^ 1) * BUF);
        // ---- phase 1: rows half 0 x features half 0
#pragma unroll
        for (int i = 0; i < MH; ++i)
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) af[i][kh] = lds_read((ab + i * 2048) ^ (kh * 64));
        if (t + 1 < nt) stage_a(0, t + 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        HG_MMA(0, 0, bf0[PAR]);
        __builtin_amdgcn_s_barrier();
        // ---- phase 2: rows half 0 x features half 1
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) bf1[j][kh] = lds_read((bb + (2 + j) * 2048) ^ (kh * 64));
        if (t + 1 < nt) stage_a(1, t + 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        HG_MMA(0, 2, bf1);
        __builtin_amdgcn_s_barrier();
        // ---- phase 3: rows half 1 x features half 1.  The weight tile of this buffer was last read in phase 2: re-fill it.  The
        // weight tile of K tile t + 1 (issued two phases of the previous tile ago) must have landed before phase 4 reads it:
        // everything younger -- the two x halves of t + 1, this phase's weight half of t + 2 -- may stay in flight
#pragma unroll
        for (int i = 0; i < MH; ++i)
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) af[i][kh] = lds_read((ab + (MH + i) * 2048) ^ (kh * 64));
        if (t + 2 < nt) {
            stage_b(0, t + 2);
            if constexpr (GLA == 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        } else if (t + 1 < nt) {
            if constexpr (GLA == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        HG_MMA(MH, 2, bf1);
        __builtin_amdgcn_s_barrier();
        // ---- phase 4: rows half 1 x features half 0 (fragments in registers since the previous tile's phase 4); read the next
        // tile's; the x tile of K tile t + 1 must have landed before the next phase 1: all but the weight halves of t + 2
        if (t + 1 < nt) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int kh = 0; kh < 2; ++kh) bf0[PAR ^ 1][j][kh] = lds_read((bn + j * 2048) ^ (kh * 64));
        }
        if (t + 2 < nt) {
            stage_b(1, t + 2);
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        HG_MMA(MH, 0, bf0[PAR]);
        __builtin_amdgcn_s_barrier();
    };
    for (int t = 0; t < nt; t += 2) {
        ktile(std::integral_constant<int, 0>{}, t);
        if (t + 1 < nt) ktile(std::integral_constant<int, 1>{}, t + 1);
    }
#undef HG_MMA
    if (!late) __builtin_amdgcn_s_barrier();

    // ---- epilogue: acc[i][j][r] = y[row0 + wr BM/2 + 16 i + (l & 15)][f0 + 64 wc + 16 j + 4 (l >> 4) + r]
    const int fq = (lane >> 4) * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int f = f0 + wc * 64 + j * 16 + fq;
        if (f >= N) continue;  // (N is a multiple of 4: whole quads are inside or outside)
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (bias) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                bv[r] = BF16 ? __builtin_bit_cast(float, (uint32_t)bias[f + r] << 16) : (float)__builtin_bit_cast(_Float16, bias[f + r]);
        }
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int row = row0 + wr * (BM / 2) + i * 16 + (lane & 15);
            if (row >= M) continue;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r] + bv[r];
            if (addend) {
                const hg_f32x4 ad = *reinterpret_cast<const hg_f32x4*>(addend + (int64_t)row * N + f);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += ad[r];
            }
            uint16_t o[4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
                o[r] = BF16 ? __builtin_bit_cast(uint16_t, (__bf16)v[r]) : __builtin_bit_cast(uint16_t, (_Float16)v[r]);
            uint2 pk;
            pk.x = (uint32_t)o[0] | ((uint32_t)o[1] << 16);
            pk.y = (uint32_t)o[2] | ((uint32_t)o[3] << 16);
            *reinterpret_cast<uint2*>(y + (int64_t)row * N + f) = pk;
        }
    }
}

// ---- 256 x 256 tile on FOUR waves: one wave per SIMD, wave tile 128 x 128 (64 accumulator tiles in the 256 accumulation registers a
// lone wave has).  Fragment traffic per matrix instruction is a third less than with 8 waves (8 + 8 fragments feed 64 instructions
// instead of 8 + 4 feeding 32), and nothing is shared between waves but the LDS tiles.  What a lone wave cannot do is hide a wait
// behind another wave, so every wait is engineered away (measured with the cycle probes below, -DHG_PROBE):
//   * K is cut into SLICES of 32 (one matrix instruction deep), FOUR LDS stages of 32 KB.  During slice s a wave issues its 64 matrix
//     instructions from registers, reads the 16 fragments of slice s + 1 from stage (s + 1) % 4 (one ds_read behind every third
//     instruction) and sends its share of slice s + 4 into stage s % 4 -- free since the barrier that opened slice s -- by LDS-DMA (one
//     instruction behind every eighth: the texture path takes ~16 cycles per 1 KB instruction and four waves share it; issued
//     back-to-back they stalled the matrix stream by ~26 cycles each in the first version).  LDS-DMA has two to three slices
//     (2200 - 3300 cycles) to land; the first version (two 64-wide stages, one slice of flight) waited 170 cycles per tile for it.
//   * one barrier per slice, preceded by s_waitcnt vmcnt(16): the LDS-DMA of slice s + 1 has landed, those of s + 2, s + 3 fly on.
//   * rows are 64 B in LDS; the 16-byte slot is XORed with (row >> 2) & 3: the 16 lanes of a quarter wave read 16 different
//     16-byte bank groups (LDS-DMA writes lane-linear, so the permutation is applied to the global source slot).
#ifdef HG_PROBE
// developer build only (-DHG_PROBE, tools/dev/probe_build.sh): cycle counts of the four waves of workgroup 0, summed over the slices
__device__ unsigned long long hg_probe_buf[4 * 8];
#define HG_T(v) const unsigned long long v = __builtin_readcyclecounter()
#else
#define HG_T(v)
#endif
template <bool BF16>
__global__ __launch_bounds__(256, 1) void gemm_h16_w4_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ w,
                                                            const uint16_t* __restrict__ bias, const float* __restrict__ addend,
                                                            int M, int N, int K, int tiles_m, int tiles_n, uint16_t* __restrict__ y) {
    constexpr int BM = 256, SK = 32;
    constexpr int A_ST = BM * SK * 2, B_ST = HBN * SK * 2, ST = A_ST + B_ST;  // 16 KB + 16 KB per stage
    extern __shared__ __align__(1024) char hg_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);  // (scalar: the LDS-DMA destinations below are scalar arithmetic)
    const int wr = wv >> 1, wc = wv & 1;
    const int nwg = tiles_m * tiles_n;
    int bid = (int)blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    int tm, tn;
    hg_tile_of<4>(bid, tiles_m, tiles_n, tm, tn);
    const int row0 = tm * BM, f0 = tn * HBN;
    const int ns = K / SK;  // (even: K is a multiple of 64)

    // LDS-DMA: one instruction = 16 rows x 64 B; a stage = 16 + 16 chunks, wave wv takes chunks 4 wv .. 4 wv + 3 of both operands
    uint32_t aoff[4], boff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int rl = (wv * 4 + i) * 16 + (lane >> 2);
        const int g = (lane & 3) ^ ((rl >> 2) & 3);
        aoff[i] = (uint32_t)(((size_t)min(row0 + rl, M - 1) * K + g * 8) * 2);
        boff[i] = (uint32_t)(((size_t)min(f0 + rl, N - 1) * K + g * 8) * 2);
    }
    const char* const xb = reinterpret_cast<const char*>(x);
    const char* const wb = reinterpret_cast<const char*>(w);
    // instruction d (0 .. 7) of a slice: operand d & 1, chunk d >> 1 of this wave; `sa` / `sb` = the slice's scalar base pointers
    auto dma_one = [&](const char* sa, const char* sb, int d, int buf) {
        const int op = d & 1, ci = d >> 1;
        char* dst = hg_smem + buf * ST + op * A_ST + (wv * 4 + ci) * 1024;
        const char* src = op ? sb + boff[ci] : sa + aoff[ci];
        __builtin_amdgcn_global_load_lds((hg_gptr)src, (hg_lptr)dst, 16, 0, 0);
    };
    const uint32_t lds0 = (uint32_t)(uintptr_t)hg_smem;
    const uint32_t fr_off = (uint32_t)((lane & 15) * 64 + (((lane >> 4) ^ ((lane >> 2) & 3)) << 4));
    const uint32_t a_base = lds0 + (uint32_t)(wr * 128 * 64) + fr_off;
    const uint32_t b_base = lds0 + (uint32_t)A_ST + (uint32_t)(wc * 128 * 64) + fr_off;
    auto lds_read = [&](uint32_t addr) -> hg_u32x4 {
        typedef const hg_u32x4 __attribute__((address_space(3))) * lp;
        return *reinterpret_cast<lp>(addr);
    };

    hg_f32x4 acc[8][8];
    {
        const hg_u32x4 z = hg_u32x4{0u, 0u, 0u, 0u};
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) hg_mfma_zero<BF16>(acc[i][j], z);
    }
    hg_u32x4 fa[2][8], fb[2][8];  // fragments of slice parity 0 / 1: rows tile i, features tile j

    // prologue: slices 0 .. 3 in flight (a slice past the end re-fetches the last one: nobody reads it, and the counted waits stay uniform)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int sl = b < ns ? b : ns - 1;
        const char* sa = xb + (size_t)sl * (SK * 2);
        const char* sb = wb + (size_t)sl * (SK * 2);
#pragma unroll
        for (int d = 0; d < 8; ++d) dma_one(sa, sb, d, b);
    }
    asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        fa[0][i] = lds_read(a_base + i * 1024);
        fb[0][i] = lds_read(b_base + i * 1024);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

#ifdef HG_PROBE
    uint32_t pr[4] = {0, 0, 0, 0};
    const unsigned long long pstart = __builtin_readcyclecounter();
    const unsigned long long rstart = __builtin_amdgcn_s_memrealtime();
#endif
    auto slice = [&](auto b_tag, const int s) {
        constexpr int B = decltype(b_tag)::value, P = B & 1, BN = (B + 1) & 3;
        HG_T(p0);
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");  // slice s + 1 has landed (mine); behind the barrier: everybody's
        HG_T(p1);
        __builtin_amdgcn_s_barrier();  // ... and every wave has read its fragments of slice s: stage B is free
        HG_T(p2);
        const int sl = s + 4 < ns ? s + 4 : ns - 1;
        const char* sa = xb + (size_t)sl * (SK * 2);
        const char* sb = wb + (size_t)sl * (SK * 2);
        asm volatile("" : "+s"(sa), "+s"(sb));
        const uint32_t an = a_base + (uint32_t)(BN * ST), bn = b_base + (uint32_t)(BN * ST);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int n = i * 8 + j;
                hg_mfma_acc<BF16>(acc[i][j], fb[P][j], fa[P][i]);
                if (n % 3 == 1 && n / 3 < 16) {  // fragments of slice s + 1, in the order the next slice uses them
                    const int r = n / 3;
                    if (r == 0) fa[P ^ 1][0] = lds_read(an);
                    else if (r <= 8) fb[P ^ 1][r - 1] = lds_read(bn + (r - 1) * 1024);
                    else fa[P ^ 1][r - 8] = lds_read(an + (r - 8) * 1024);
                }
                if (n % 8 == 5) dma_one(sa, sb, n / 8, B);
            }
        }
        HG_T(p3);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifdef HG_PROBE
        HG_T(p4);
        pr[0] += (uint32_t)(p1 - p0); pr[1] += (uint32_t)(p2 - p1); pr[2] += (uint32_t)(p3 - p2); pr[3] += (uint32_t)(p4 - p3);
#endif
    };
    for (int s = 0; s < ns; s += 4) {
        slice(std::integral_constant<int, 0>{}, s);
        slice(std::integral_constant<int, 1>{}, s + 1);
        if (s + 2 >= ns) break;
        slice(std::integral_constant<int, 2>{}, s + 2);
        slice(std::integral_constant<int, 3>{}, s + 3);
    }
    // the matrix instructions are asm statements: hipcc's hazard recogniser does not know that the accumulation registers it is
    // about to read were written by the matrix pipe; and no LDS-DMA may be in flight when the workgroup's LDS is handed on
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_waitcnt vmcnt(0)" ::: "memory");
#ifdef HG_PROBE
    if (bid == 0 && lane == 0) {
        for (int k = 0; k < 4; ++k) hg_probe_buf[wv * 8 + k] = pr[k];
        hg_probe_buf[wv * 8 + 7] = __builtin_readcyclecounter() - pstart;
        hg_probe_buf[wv * 8 + 4] = __builtin_amdgcn_s_memrealtime() - rstart;
    }
#endif

    // ---- epilogue: acc[i][j][r] = y[row0 + 128 wr + 16 i + (l & 15)][f0 + 128 wc + 16 j + 4 (l >> 4) + r]
    const int fq = (lane >> 4) * 4;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int f = f0 + wc * 128 + j * 16 + fq;
        if (f >= N) continue;
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (bias) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                bv[r] = BF16 ? __builtin_bit_cast(float, (uint32_t)bias[f + r] << 16) : (float)__builtin_bit_cast(_Float16, bias[f + r]);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = row0 + wr * 128 + i * 16 + (lane & 15);
            if (row >= M) continue;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r] + bv[r];
            if (addend) {
                const hg_f32x4 ad = *reinterpret_cast<const hg_f32x4*>(addend + (int64_t)row * N + f);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += ad[r];
            }
            uint16_t o[4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
                o[r] = BF16 ? __builtin_bit_cast(uint16_t, (__bf16)v[r]) : __builtin_bit_cast(uint16_t, (_Float16)v[r]);
            uint2 pk;
            pk.x = (uint32_t)o[0] | ((uint32_t)o[1] << 16);
            pk.y = (uint32_t)o[2] | ((uint32_t)o[3] << 16);
            *reinterpret_cast<uint2*>(y + (int64_t)row * N + f) = pk;
        }
    }
#ifdef HG_PROBE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (bid == 0 && lane == 0) hg_probe_buf[wv * 8 + 5] = __builtin_amdgcn_s_memrealtime() - rstart;
#endif
}

}  // namespace

// shapes the kernel serves: whole K tiles of 64, features in quads, 16-byte aligned rows (all of them true for every layer whose
// in_features is a multiple of 64)
bool gemm_h16_supported(int64_t M, int64_t N, int64_t K) {
    return M >= 1 && N >= 4 && (N & 3) == 0 && K >= 64 && (K & 63) == 0 && M * K * 2 < (1ll << 32) && N * K * 2 < (1ll << 32) && M < (1 << 30);
}

// BM: 256-row tiles when they fill at least 7/8 of the chip once, else 128-row tiles (twice the workgroups)
int gemm_h16(const void* x, const void* w, const void* bias, const float* addend, int dtype, int64_t M, int64_t N, int64_t K, void* y,
             hipStream_t stream) {
    if (!gemm_h16_supported(M, N, K)) return fail(-1, "gemm_h16: shape %lld x %lld x %lld not supported", (long long)M, (long long)N, (long long)K);
    const int ncu = std::max(1, current_device_cus());
    const int tiles_n = (int)((N + HBN - 1) / HBN);
    const int64_t t256 = ((M + 255) / 256) * tiles_n;
    const long long force = opt_get(OPT_GEMM_H16_BM);
    const bool big = force > 0 ? force >= 256 : 8 * t256 >= 7 * ncu;  // (224 tiles of 256 rows on 256 CUs: 112 us against 139 us with 128-row tiles)
    if (force == 512) {  // developer: the four-wave kernel (128 x 128 wave tiles)
        const int tiles_m = (int)((M + 255) / 256);
        const size_t lds = 4 * (size_t)(256 + HBN) * 32 * 2;  // four stages of 32-wide slices
        if (dtype == 1) {
            int rc_ = ensure_dynamic_lds(reinterpret_cast<const void*>(gemm_h16_w4_kernel<true>), lds);
            if (rc_) return rc_;
            hipLaunchKernelGGL((gemm_h16_w4_kernel<true>), dim3((unsigned)(tiles_m * tiles_n)), dim3(256), lds, stream,
                               static_cast<const uint16_t*>(x), static_cast<const uint16_t*>(w), static_cast<const uint16_t*>(bias), addend, (int)M, (int)N, (int)K,
                               tiles_m, tiles_n, static_cast<uint16_t*>(y));
        } else {
            int rc_ = ensure_dynamic_lds(reinterpret_cast<const void*>(gemm_h16_w4_kernel<false>), lds);
            if (rc_) return rc_;
            hipLaunchKernelGGL((gemm_h16_w4_kernel<false>), dim3((unsigned)(tiles_m * tiles_n)), dim3(256), lds, stream,
                               static_cast<const uint16_t*>(x), static_cast<const uint16_t*>(w), static_cast<const uint16_t*>(bias), addend, (int)M, (int)N, (int)K,
                               tiles_m, tiles_n, static_cast<uint16_t*>(y));
        }
        GANQ_LAUNCH_CHECK();
        return 0;
    }
    const uint16_t* xp = static_cast<const uint16_t*>(x);
    const uint16_t* wp = static_cast<const uint16_t*>(w);
    const uint16_t* bp = static_cast<const uint16_t*>(bias);
    uint16_t* yp = static_cast<uint16_t*>(y);
#define GANQ_H16_LAUNCH(BF, BMV)                                                                                                    \
    do {                                                                                                                            \
        const int tiles_m = (int)((M + BMV - 1) / BMV);                                                                             \
        int rc_ = ensure_dynamic_lds(reinterpret_cast<const void*>(gemm_h16_kernel<BF, BMV>), hg_lds_bytes<BMV>());                  \
        if (rc_) return rc_;                                                                                                        \
        hipLaunchKernelGGL((gemm_h16_kernel<BF, BMV>), dim3((unsigned)(tiles_m * tiles_n)), dim3(HTHREADS), hg_lds_bytes<BMV>(), stream, \
                           xp, wp, bp, addend, (int)M, (int)N, (int)K, tiles_m, tiles_n, yp);                                        \
    } while (0)
    if (dtype == 1) {
        if (big) GANQ_H16_LAUNCH(true, 256);
        else GANQ_H16_LAUNCH(true, 128);
    } else {
        if (big) GANQ_H16_LAUNCH(false, 256);
        else GANQ_H16_LAUNCH(false, 128);
    }
#undef GANQ_H16_LAUNCH
    GANQ_LAUNCH_CHECK();
    return 0;
}

}  // namespace ganq

using namespace ganq;

/* developer / tests: the dense kernel on its own (x [M,K], w [N,K], y [M,N]; dtype 0 = fp16, 1 = bf16) */
extern "C" int ganq_debug_gemm_h16(const void* x, const void* w, const void* bias, const float* addend, int dtype, int64_t M, int64_t N,
                                   int64_t K, void* y, void* stream) {
    if (!x || !w || !y) return fail(-3, "ganq_debug_gemm_h16: null pointer");
    return gemm_h16(x, w, bias, addend, dtype, M, N, K, y, static_cast<hipStream_t>(stream));
}

#ifdef HG_PROBE
extern "C" int ganq_debug_gemm_probe(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(hg_probe_buf), sizeof(unsigned long long) * 32);
}
#endif

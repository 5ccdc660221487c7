// Hessian accumulation for staged groups of batches, second generation (round 4): H <- H * decay + scale * X^T X with 256 x 256 tiles,
// one wave per SIMD (128 x 128 wave tiles), LDS-DMA and a token-slice pipeline -- the structure of the four-wave dense GEMM
// (gemm_h16.hip, gemm_h16_w4_kernel) with two differences that the product X^T X brings:
//   * the matrix instructions want 8 consecutive TOKENS of one feature per lane, X is [token][feature].  The first version of this
//     kernel staged X as it lies and read the fragments through ds_read_b64_tr_b16 (the hardware transpose, as the 128 x 128 kernels
//     in hessian.hip do): correct, conflict-free by the counters -- and LDS-bound at 2080 cycles per 32-token slice against 1024 of
//     matrix work, because a transposed read moves 512 B in ~16 cycles (a ds_read_b128 moves 1 KB in 8) and a 128 x 128 wave tile
//     needs 32 of them per slice.  So the batch is transposed ONCE (hw_transpose_kernel: Xt [feature][token] in the workspace, the
//     transposed reads used there, where they cost nothing against the memory traffic) and the main kernel is the dense GEMM's
//     Xt Xt^T: 64-byte rows in LDS, slot XOR (row >> 2) & 3, plain 16-byte fragment reads;
//   * only tiles on or below the diagonal exist (136 at n = 4096, fewer than CUs), so the launch is cut STREAM-K: the (tile, slice)
//     pairs are one linear range, every workgroup (one per CU) takes an equal contiguous share of it -- at most two partial tiles
//     and any number of whole ones.  A partial tile is stored in register order, write-through (sc1), the workgroup takes a ticket on
//     the tile's counter and the LAST one to arrive sums the parts IN RANGE ORDER (its own from registers at its place in the order):
//     no atomics on data, no waiting, the same bits whoever finishes.  The mirror image is written from the same registers.
// fp16 x fp16 products are exact in fp32, so this differs from the other Hessian kernels only in the grouping of the fp32 sums.
#include <algorithm>
#include <type_traits>

#include "common.h"
#include "mfma_h16.h"

namespace ganq {

namespace {

constexpr int WT = 256;                 // tile edge
constexpr int WSK = 32;                 // tokens per slice
constexpr int W_IMG = WSK * WT * 2;     // one operand image of a stage: 32 token rows x 512 B
constexpr int W_ST = 2 * W_IMG;         // a stage: u image + v image
constexpr int W_NST = 4;
constexpr int W_PART = WT * WT;         // floats of a partial tile
constexpr int W_SLOTS = 3;             // partial tiles a workgroup may leave (a helper range touches at most three tiles)
constexpr int W_MAX_TILES = 4096;       // (n <= 23 k; the head of the workspace is reserved for as many words)

typedef short hw_s4 __attribute__((__vector_size__(4 * sizeof(short))));
typedef uint32_t hw_u32x2 __attribute__((ext_vector_type(2)));

// tile number -> (bu >= bv): the host's table walks the lower triangle along the Z curve, so that the 32 workgroups behind one L2,
// which hold consecutive tile numbers, cover a 4 x 8 / 8 x 4 block of tiles (12 operand panels per slice fetched past the L2 instead
// of 33 for a strip of a row of tiles)
__device__ __forceinline__ void hw_tile_of(const uint32_t* __restrict__ table, int t, int& bu, int& bv) {
    const uint32_t e = table[t];
    bu = (int)(e >> 16);
    bv = (int)(e & 0xffffu);
}

// The staging copy, transposing: Xt[f][tok0 + t] = X[t][f] for one calibration batch X [rows][n].  128 tokens x 128 features per
// workgroup through LDS, 16-byte loads along the features, transposed reads (ds_read_b64_tr_b16: a 16-lane group takes a 4 x 16 block and
// hands every lane 4 tokens of one feature), 16-byte stores along the tokens -- four lanes write 64 contiguous bytes of a feature
// row, workgroups that run side by side (token tiles vary fastest) the neighbouring lines of the same rows.  Tokens past `rows` are
// neither read nor written (groups of 8: tok0 and rows are multiples of 8).
constexpr int WTP = 144;  // LDS pitch in 16-bit elements (288 B)
constexpr int WTT = 128;  // tokens per workgroup (eight 16-byte loads per thread in flight; 64: 3.0 TB/s, 128: see DESIGN.md)
__global__ __launch_bounds__(256) void hw_stage_kernel(const uint16_t* __restrict__ X, uint16_t* __restrict__ Xt, int rows, int n, int64_t ldt) {
    __shared__ __align__(16) uint16_t S[WTT][WTP];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int tok0 = blockIdx.x * WTT, f0 = blockIdx.y * 128;
    constexpr int NL = WTT * 16 / 256;  // 16-byte pieces per thread
    uint4 v[NL];
#pragma unroll
    for (int k = 0; k < NL; ++k) {
        const int id = tid + 256 * k, row = id >> 4, col = (id & 15) * 8;
        v[k] = (tok0 + row < rows && f0 + col < n) ? *reinterpret_cast<const uint4*>(X + (size_t)(tok0 + row) * n + f0 + col) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int k = 0; k < NL; ++k) {
        const int id = tid + 256 * k, row = id >> 4, col = (id & 15) * 8;
        *reinterpret_cast<uint4*>(&S[row][col]) = v[k];
    }
    __syncthreads();
    const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    typedef __attribute__((address_space(3))) hw_s4* lp;
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
        for (int h = 0; h < WTT / 32; ++h) {
            const int fb = 32 * wv + 16 * blk;
            const hw_s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)&S[32 * h + 8 * g + q][fb + 4 * p]);
            const hw_s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)&S[32 * h + 8 * g + q + 4][fb + 4 * p]);
            const hw_u32x2 l2 = __builtin_bit_cast(hw_u32x2, lo), h2 = __builtin_bit_cast(hw_u32x2, hi);
            const int f = f0 + fb + (lane & 15), t = tok0 + 32 * h + 8 * g;
            if (f < n && t < rows) *reinterpret_cast<uint4*>(Xt + (size_t)f * (size_t)ldt + t) = make_uint4(l2[0], l2[1], h2[0], h2[1]);
        }
}

// ---- the cut of a launch (shared by the host, the kernels and the CPU test of the cut: tests/test_host_logic.py through
// ganq_debug_hessian_t_cut).  T tiles of Ks token slices on G workgroups: W = T div G whole tiles per workgroup first (tile w G + c:
// everybody walks the tokens of a row of tiles in step, the operand panels are shared in the L2).  The R = T mod G tiles left over are
// cut so that the workgroups still read the same tokens at the same time: workgroup c < nprim is the PRIMARY of left-over tile c and
// takes its slices [0, P); the others are HELPERS that share the tails [P, Ks) of all left-over tiles in equal contiguous ranges of Sh
// slices, each started at the tile boundary inside its range (primaries are at slice tau, helpers at P + tau or P + tau - (Sh - Lt) at any
// time tau; a plain linear cut of the (tile, slice) pairs spreads the workgroups over all token offsets, every panel is then fetched
// past the L2 by each of them: 294 us of slice loops at n = 4096 instead of 190).  nprim = 0, P = 0 is the plain linear cut of the
// left-over tiles (taken when a helper range would touch more than three tiles: Sh > 2 (Ks - P)).
struct HwCut { int Ks, G, W, R, nprim, P, Sh; };
struct HwSeg { int t, s0, s1, np, slot; };  // slices [s0, s1) of tile t: one of the tile's np parts, kept in slot `slot` of the workgroup's partial tiles when np > 1

inline HwCut hw_make_cut(int64_t T, int64_t Ks, int G) {
    HwCut q{(int)Ks, G, (int)(T / G), (int)(T % G), 0, 0, 1};
    if (q.R > 0) {
        q.P = (int)(((int64_t)q.R * Ks + G - 1) / G);
        const int64_t Lt = Ks - q.P;
        const int64_t sh = G > q.R && Lt > 0 ? ((int64_t)q.R * Lt + (G - q.R) - 1) / (G - q.R) : 0;
        if (q.P < Ks && sh >= 1 && sh <= 2 * Lt) {
            q.nprim = q.R;
            q.Sh = (int)sh;
        } else {  // the plain linear cut of the left-over tiles
            q.P = 0;
            q.Sh = (int)std::max<int64_t>(1, ((int64_t)q.R * Ks + G - 1) / G);
        }
    }
    return q;
}

// helpers whose ranges touch left-over tile r: the first one's number -> hf, their count returned
__host__ __device__ inline int hw_helpers_of(const HwCut& q, int r, int& hf) {
    const int Lt = q.Ks - q.P;
    hf = (int)(((long long)r * Lt) / q.Sh);
    return (int)((((long long)(r + 1)) * Lt - 1) / q.Sh) - hf + 1;
}

// parts of left-over tile r, in the order they are summed: the primary's (when P > 0), then the helpers' along the tail
__host__ __device__ inline int hw_parts_of(const HwCut& q, int r) {
    int hf;
    return (q.P > 0 ? 1 : 0) + hw_helpers_of(q, r, hf);
}
__host__ __device__ inline void hw_part_location(const HwCut& q, int r, int pi, int& wg, int& slot) {
    const int pb = q.P > 0 ? 1 : 0;
    wg = r;
    slot = 0;
    if (pi >= pb) {
        int hf;
        (void)hw_helpers_of(q, r, hf);
        const int h = hf + pi - pb, Lt = q.Ks - q.P;
        wg = q.nprim + h;
        slot = r - (int)(((long long)h * q.Sh) / Lt);
    }
}

// the left-over segments of workgroup c (at most three), in the order it works through them -> count
__host__ __device__ inline int hw_left_segments(const HwCut& q, int c, HwSeg (&left)[3]) {
    int nleft = 0;
    if (q.R <= 0) return 0;
    const int Lt = q.Ks - q.P, t0 = q.W * q.G;
    if (c < q.nprim) {
        left[nleft++] = HwSeg{t0 + c, 0, q.P, hw_parts_of(q, c), 0};
        return nleft;
    }
    const int h = c - q.nprim;
    const long long a = (long long)h * q.Sh, tot = (long long)q.R * Lt;
    const long long b = a + q.Sh < tot ? a + q.Sh : tot;
    if (a >= b) return 0;
    const int r0 = (int)(a / Lt);
    long long x = (a % Lt == 0) ? a : (long long)(r0 + 1) * Lt;  // the tile boundary inside the range: the helper starts there
    if (x >= b) x = a;
    for (int pass = 0; pass < 2; ++pass) {
        long long pos = pass == 0 ? x : a;
        const long long hi = pass == 0 ? b : x;
        while (pos < hi && nleft < 3) {
            const int r = (int)(pos / Lt), off = (int)(pos - (long long)r * Lt);
            const int len = (int)((long long)(Lt - off) < hi - pos ? (long long)(Lt - off) : hi - pos);
            left[nleft++] = HwSeg{t0 + r, q.P + off, q.P + off + len, hw_parts_of(q, r), r - r0};
            pos += len;
        }
    }
    return nleft;
}

#ifdef HW_PROBE
__device__ unsigned long long hw_probe_buf[8];
__device__ unsigned long long hw_probe_all[1024 * 4];  // per workgroup: kernel start, loop ticks, end ticks, finish (100 MHz)  // developer (-DHW_PROBE): workgroup 0 -- cycles in the slice loops, slices, cycles in the segment ends
#endif
template <bool BF16>
__global__ __launch_bounds__(256, 1) void hessian_w4_kernel(float* __restrict__ H, const uint16_t* __restrict__ Xt, int ldt, int n, float decay,
                                                           float scale, const HwCut cut,
                                                           float* __restrict__ partial, const uint32_t* __restrict__ table) {
    extern __shared__ __align__(1024) char hw_smem[];
    const int Ks = cut.Ks, nwg = cut.G, W = cut.W;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wv >> 1, wc = wv & 1;
    int c = (int)blockIdx.x;  // consecutive shares behind one L2: they walk the same tiles
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = c & 7, idx = c >> 3;
        c = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const char* const xb = reinterpret_cast<const char*>(Xt);
    const size_t row_bytes = (size_t)ldt * 2;  // one feature's tokens

    // fragment reads: lane l holds feature (l & 15) of a 16-feature tile, k group (l >> 4) of the 32-token slice
    const uint32_t lds0 = (uint32_t)(uintptr_t)hw_smem;
    const uint32_t fr_off = (uint32_t)((lane & 15) * 64 + (((lane >> 4) ^ ((lane >> 2) & 3)) << 4));
    const uint32_t a_base = lds0 + (uint32_t)(wr * 128 * 64) + fr_off;
    const uint32_t b_base = lds0 + (uint32_t)W_IMG + (uint32_t)(wc * 128 * 64) + fr_off;
    auto lds_read = [&](uint32_t addr) -> hg_u32x4 {
        typedef const hg_u32x4 __attribute__((address_space(3))) * lp;
        return *reinterpret_cast<lp>(addr);
    };

    hg_f32x4 acc[8][8];
    hg_u32x4 fa[2][8], fb[2][8];
    const __amdgpu_buffer_rsrc_t rs_part = __builtin_amdgcn_make_buffer_rsrc(partial, 0, 0xffffffff, 0x00020000);

    // One segment: the token slices [s0, s1) of tile t; one of the tile's `np` parts, kept in slot `slot` of this workgroup's partial
    // tiles when np > 1.
    auto process = [&](const HwSeg sg) {
        // (uniform by construction; said so explicitly: the scalar pointer arithmetic below must stay in scalar registers)
        const int t = __builtin_amdgcn_readfirstlane(sg.t), s0 = __builtin_amdgcn_readfirstlane(sg.s0), s1 = __builtin_amdgcn_readfirstlane(sg.s1);
        const int np = __builtin_amdgcn_readfirstlane(sg.np), slot = __builtin_amdgcn_readfirstlane(sg.slot);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the previous segment's stores to H: the counted waits below count LDS-DMA only)
        int bu, bv;
        hw_tile_of(table, t, bu, bv);
        const int u0 = bu * WT, v0 = bv * WT;
        const int ns = s1 - s0;

        // LDS-DMA sources of a slice: one instruction = 16 feature rows x 64 B (32 tokens); wave wv takes rows 64 wv .. 64 wv + 63 of both images
        uint32_t aoff[4], boff[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int rl = (wv * 4 + i) * 16 + (lane >> 2);
            const int gsl = (lane & 3) ^ ((rl >> 2) & 3);
            aoff[i] = (uint32_t)((size_t)min(u0 + rl, n - 1) * row_bytes + gsl * 16);
            boff[i] = (uint32_t)((size_t)min(v0 + rl, n - 1) * row_bytes + gsl * 16);
        }
        auto dma_one = [&](const char* sbase, int d, int buf) {  // instruction d (0 .. 7) of a slice: operand d & 1, chunk d >> 1
            const int op = d & 1, ci = d >> 1;
            char* dst = hw_smem + buf * W_ST + op * W_IMG + (wv * 4 + ci) * 1024;
            __builtin_amdgcn_global_load_lds((hg_gptr)(sbase + (op ? boff[ci] : aoff[ci])), (hg_lptr)dst, 16, 0, 0);
        };
        {
            const hg_u32x4 z = hg_u32x4{0u, 0u, 0u, 0u};
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) hg_mfma_zero<BF16>(acc[i][j], z);
        }
        __builtin_amdgcn_s_barrier();  // the previous segment's LDS reads (and its ticket word) are done in every wave
#ifdef HW_PROBE_ALIGN
        const char* const seg = xb;  // (timing experiment: every workgroup reads the same tokens -- wrong sums)
#else
        const char* const seg = xb + (size_t)s0 * (WSK * 2);
#endif
#pragma unroll
        for (int b = 0; b < 4; ++b) {  // slices 0 .. 3 in flight (past the end: the last one again, nobody reads it)
            const char* sb = seg + (size_t)(b < ns ? b : ns - 1) * (WSK * 2);
#pragma unroll
            for (int d = 0; d < 8; ++d) dma_one(sb, d, b);
        }
        asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            fa[0][i] = lds_read(a_base + i * 1024);
            fb[0][i] = lds_read(b_base + i * 1024);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

#ifdef HW_PROBE
        uint32_t pq[4] = {0, 0, 0, 0};
#endif
        auto slice = [&](auto b_tag, const int s) {
            constexpr int B = decltype(b_tag)::value, P = B & 1, BN = (B + 1) & 3;
#ifdef HW_PROBE
            const unsigned long long q0 = __builtin_readcyclecounter();
#endif
            asm volatile("s_waitcnt vmcnt(16)" ::: "memory");  // slice s + 1 has landed (mine); behind the barrier: everybody's
#ifdef HW_PROBE
            const unsigned long long q1 = __builtin_readcyclecounter();
#endif
            __builtin_amdgcn_s_barrier();                       // ... and every wave has read its fragments of slice s: stage B is free
#ifdef HW_PROBE
            const unsigned long long q2 = __builtin_readcyclecounter();
#endif
            const char* sb = seg + (size_t)(s + 4 < ns ? s + 4 : ns - 1) * (WSK * 2);
            asm volatile("" : "+s"(sb));
            const uint32_t an = a_base + (uint32_t)(BN * W_ST), bn = b_base + (uint32_t)(BN * W_ST);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int k = i * 8 + j;
                    hg_mfma_acc<BF16>(acc[i][j], fb[P][j], fa[P][i]);
                    if (k % 3 == 1 && k / 3 < 16) {  // fragments of slice s + 1, in the order the next slice uses them
                        const int r = k / 3;
                        if (r == 0) fa[P ^ 1][0] = lds_read(an);
                        else if (r <= 8) fb[P ^ 1][r - 1] = lds_read(bn + (r - 1) * 1024);
                        else fa[P ^ 1][r - 8] = lds_read(an + (r - 8) * 1024);
                    }
                    if (k % 8 == 6) dma_one(sb, k / 8, B);
                }
            }
#ifdef HW_PROBE
            const unsigned long long q3 = __builtin_readcyclecounter();
#endif
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifdef HW_PROBE
            const unsigned long long q4 = __builtin_readcyclecounter();
            pq[0] += (uint32_t)(q1 - q0); pq[1] += (uint32_t)(q2 - q1); pq[2] += (uint32_t)(q3 - q2); pq[3] += (uint32_t)(q4 - q3);
#endif
        };
#ifdef HW_PROBE
        const unsigned long long pt0 = __builtin_readcyclecounter(), pr0 = __builtin_amdgcn_s_memrealtime();
#endif
        for (int s = 0; s < ns; s += 4) {
            slice(std::integral_constant<int, 0>{}, s);
            if (s + 1 >= ns) break;
            slice(std::integral_constant<int, 1>{}, s + 1);
            if (s + 2 >= ns) break;
            slice(std::integral_constant<int, 2>{}, s + 2);
            if (s + 3 >= ns) break;
            slice(std::integral_constant<int, 3>{}, s + 3);
        }
        // (matrix instructions in asm statements: the hazard recogniser does not know the accumulation registers are still being written;
        // and no LDS-DMA may be in flight when the stages are re-used or handed on)
        asm volatile("s_nop 15\n\ts_nop 15\n\ts_waitcnt vmcnt(0)" ::: "memory");
#ifdef HW_PROBE
        const unsigned long long pt1 = __builtin_readcyclecounter(), pr1 = __builtin_amdgcn_s_memrealtime();
        if (tid == 0 && c < 1024) { hw_probe_all[c * 4 + 1] += pr1 - pr0; if (hw_probe_all[c * 4] == 0) hw_probe_all[c * 4] = pr0; }
        if (c == 0 && tid == 0) { hw_probe_buf[0] += pt1 - pt0; hw_probe_buf[1] += ns; hw_probe_buf[3] += pr1 - pr0; hw_probe_buf[5] += pq[0]; hw_probe_buf[6] += pq[1]; hw_probe_buf[7] += pq[3]; }
        struct ProbeEnd { unsigned long long t, r; bool on; int c; bool w; __device__ ~ProbeEnd() { const unsigned long long now = __builtin_amdgcn_s_memrealtime(); if (on) { hw_probe_buf[2] += __builtin_readcyclecounter() - t; hw_probe_buf[4] += now - r; } if (w) { hw_probe_all[c * 4 + 2] += now - r; hw_probe_all[c * 4 + 3] = now; } } } pe{pt1, pr1, c == 0 && tid == 0, c, tid == 0 && c < 1024};
#endif

        // ---- the segment's end.  acc[i][j][r] = sum over its tokens of X[., u] X[., v], u = u0 + 128 wr + 16 i + (l & 15),
        // v = v0 + 128 wc + 16 j + 4 (l >> 4) + r
        const uint32_t wave_off = (uint32_t)(wv * (W_PART / 4) * 4);
        if (np > 1) {
            // a part of a left-over tile: stored in register order (whole 1 KB per instruction); hessian_w4_fix_kernel sums the parts
            const uint32_t pbase = (uint32_t)(((size_t)c * W_SLOTS + slot) * W_PART * 4) + wave_off;
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(hg_u32x4, acc[i][j]), rs_part, lane * 16,
                                                           pbase + (uint32_t)((i * 8 + j) * 1024), 0);
            return;
        }
        // a whole tile: finished from the registers
        const bool diag = bu == bv;
        const int ur = u0 + wr * 128 + (lane & 15), vq = v0 + wc * 128 + 4 * (lane >> 4);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int u = ur + 16 * i;
            hg_f32x4 old[8];
            if (decay != 0.0f) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int v = vq + 16 * j;
                    old[j] = (u < n && v < n) ? *reinterpret_cast<const hg_f32x4*>(H + (size_t)u * n + v) : hg_f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int v = vq + 16 * j;
                if (u >= n || v >= n) continue;
                hg_f32x4 val;
#pragma unroll
                for (int r = 0; r < 4; ++r) val[r] = (decay != 0.0f ? old[j][r] * decay : 0.0f) + scale * acc[i][j][r];
                *reinterpret_cast<hg_f32x4*>(H + (size_t)u * n + v) = val;
                if (!diag) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) H[(size_t)(v + r) * n + u] = val[r];
                }
            }
        }
    };

    // ---- this workgroup's segments: its whole tiles, then its share of the left-over ones (hw_left_segments)
    HwSeg left[3];
    const int nleft = hw_left_segments(cut, c, left);
    for (int k = 0; k < W + nleft; ++k) {
        const int kl = k - W;
        const HwSeg sg = k < W ? HwSeg{k * nwg + c, 0, Ks, 1, 0} : (kl == 0 ? left[0] : (kl == 1 ? left[1] : left[2]));
        process(sg);
    }
}

// The left-over tiles: part sums in the order of the cut (the primary's, then the helpers' along the tails), then H <- H * decay +
// scale * sum on the tile and its mirror image.  One workgroup per QUARTER of a tile (the 128 x 128 block one wave of the main kernel
// held, 64 KB per part, register order: whole 1 KB lines per load instruction).  No atomics, no tickets: the kernel boundary is the
// hand-over, the order is fixed, the result the same from run to run.
__global__ __launch_bounds__(256) void hessian_w4_fix_kernel(float* __restrict__ H, int n, float decay, float scale, const HwCut cut,
                                                            const float* __restrict__ partial, const uint32_t* __restrict__ table) {
    const int tid = threadIdx.x;
    const int r = (int)blockIdx.x >> 2, wvq = (int)blockIdx.x & 3;  // left-over tile, quarter (wr, wc)
    const int t = cut.W * cut.G + r;
    int bu, bv;
    hw_tile_of(table, t, bu, bv);
    const int wr = wvq >> 1, wc = wvq & 1;
    const int np = hw_parts_of(cut, r);
    const bool diag = bu == bv;
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
        hg_f32x4 sum[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) sum[e] = hg_f32x4{0.f, 0.f, 0.f, 0.f};
        for (int pi0 = 0; pi0 < np; pi0 += 3) {  // three parts per round trip (the usual tile has two or three)
            hg_f32x4 v[3][8];
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const int pi = pi0 + d;
                if (pi >= np) continue;
                int cc, sl;
                hw_part_location(cut, r, pi, cc, sl);
                const float* src = partial + ((size_t)cc * W_SLOTS + sl) * W_PART + (size_t)wvq * (W_PART / 4);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[d][e] = *reinterpret_cast<const hg_f32x4*>(src + (size_t)((half * 8 + e) * 256 + tid) * 4);
            }
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                if (pi0 + d >= np) continue;
#pragma unroll
                for (int e = 0; e < 8; ++e) sum[e] += v[d][e];
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int idx = (half * 8 + e) * 256 + tid, ij = idx >> 6, l = idx & 63;
            const int u = bu * WT + wr * 128 + 16 * (ij >> 3) + (l & 15), v = bv * WT + wc * 128 + 16 * (ij & 7) + 4 * (l >> 4);
            if (u >= n || v >= n) continue;
            hg_f32x4 val;
            if (decay != 0.0f) {
                const hg_f32x4 old = *reinterpret_cast<const hg_f32x4*>(H + (size_t)u * n + v);
#pragma unroll
                for (int q = 0; q < 4; ++q) val[q] = old[q] * decay + scale * sum[e][q];
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) val[q] = 0.0f + scale * sum[e][q];
            }
            *reinterpret_cast<hg_f32x4*>(H + (size_t)u * n + v) = val;
            if (!diag) {
#pragma unroll
                for (int q = 0; q < 4; ++q) H[(size_t)(v + q) * n + u] = val[q];
            }
        }
    }
}

}  // namespace

// layers the transposed path serves: whole 16-byte pieces per token row / feature row, 32-bit offsets into Xt
bool hessian_w4_supported(int64_t n, int64_t ldt) {
    // (from 3072 in_features on: below, the 128 x 128 kernels of hessian.hip are as fast or faster -- n = 2048: 128 against 137 us per 16384
    // tokens, n = 1024: 90 against 78 -- and keep the plain staging copy; GANQ_HESS_W4 = 2 (tests) lowers the bar to 1024)
    if (n < (opt_get(OPT_HESS_W4) == 2 ? 1024 : 3072) || (n % 8) != 0 || ldt < WSK || (ldt % WSK) != 0 || current_device_cus() < 8) return false;
    const int64_t nt = (n + WT - 1) / WT;
    return nt * (nt + 1) / 2 <= W_MAX_TILES && n * ldt * 2 < ((int64_t)1 << 32) && opt_get(OPT_HESS_W4) != 0;
}

size_t hessian_w4_workspace_bytes(int64_t n) {
    return (size_t)W_MAX_TILES * sizeof(int) + (size_t)std::max(1, current_device_cus()) * W_SLOTS * W_PART * sizeof(float);
}

// the staging copy of one batch: Xt[:, tok0 : tok0 + rows] = X^T
int hessian_w4_stage(void* Xt, int64_t ldt, const void* X, int64_t rows, int64_t n, int64_t tok0, hipStream_t stream) {
    if (rows == 0) return 0;
    if ((tok0 % 8) != 0 || (rows % 8) != 0 || (n % 8) != 0 || tok0 + rows > ldt || (ldt % 8) != 0)
        return fail(-1, "ganq_hessian_stage_t: tokens %lld + %lld of %lld, in_features %lld (multiples of 8 expected)", (long long)tok0, (long long)rows,
                    (long long)ldt, (long long)n);
    if ((reinterpret_cast<uintptr_t>(X) & 15) != 0 || (reinterpret_cast<uintptr_t>(Xt) & 15) != 0) return fail(-3, "ganq_hessian_stage_t: 16-byte aligned buffers expected");
    hipLaunchKernelGGL(hw_stage_kernel, dim3((unsigned)((rows + WTT - 1) / WTT), (unsigned)((n + 127) / 128)), dim3(256), 0, stream,
                       static_cast<const uint16_t*>(X), static_cast<uint16_t*>(Xt) + tok0, (int)rows, (int)n, (int64_t)ldt);
    GANQ_LAUNCH_CHECK();
    return 0;
}

// H <- H * decay + scale * Xt[:, :rows] Xt[:, :rows]^T   (rows: a multiple of 32)
int hessian_w4(float* H, const void* Xt, int64_t ldt, int dtype, int64_t rows, int64_t n, float decay, float scale, void* workspace,
               size_t workspace_bytes, hipStream_t stream) {
    if (!hessian_w4_supported(n, ldt) || rows <= 0 || (rows % WSK) != 0 || rows > ldt)
        return fail(-1, "ganq_hessian_accum_t: in_features %lld, %lld of %lld tokens not served (ganq_hessian_t_supported)", (long long)n, (long long)rows,
                    (long long)ldt);
    const size_t need = hessian_w4_workspace_bytes(n);
    if (!workspace || workspace_bytes < need) return fail(-4, "ganq_hessian_accum_t: workspace %zu B < required %zu B", workspace_bytes, need);
    if ((reinterpret_cast<uintptr_t>(Xt) & 15) != 0 || (reinterpret_cast<uintptr_t>(H) & 15) != 0) return fail(-3, "ganq_hessian_accum_t: 16-byte aligned buffers expected");
    const int ncu = std::max(1, current_device_cus());
    const int64_t nt = (n + WT - 1) / WT;
    const int64_t T = nt * (nt + 1) / 2;
    const int64_t Ks = rows / WSK;
    const HwCut cut = hw_make_cut(T, Ks, ncu);
    const int nwg = cut.G, R = cut.R;
    int tcount = 0;
    const uint32_t* table = hessian_tile_table((int)nt, &tcount);
    if (!table || tcount != (int)T) return fail(-100, "ganq_hessian_accum_t: could not build the tile table");
    float* partial = reinterpret_cast<float*>(static_cast<char*>(workspace) + (size_t)W_MAX_TILES * sizeof(int));
    const size_t lds = (size_t)W_NST * W_ST;
    const uint16_t* Xp = static_cast<const uint16_t*>(Xt);
    if (dtype == 1) {
        const int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(hessian_w4_kernel<true>), lds);
        if (rc) return rc;
        hipLaunchKernelGGL(hessian_w4_kernel<true>, dim3((unsigned)nwg), dim3(256), lds, stream, H, Xp, (int)ldt, (int)n, decay, scale, cut, partial, table);
    } else {
        const int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(hessian_w4_kernel<false>), lds);
        if (rc) return rc;
        hipLaunchKernelGGL(hessian_w4_kernel<false>, dim3((unsigned)nwg), dim3(256), lds, stream, H, Xp, (int)ldt, (int)n, decay, scale, cut, partial, table);
    }
    if (R > 0)
        hipLaunchKernelGGL(hessian_w4_fix_kernel, dim3((unsigned)(4 * R)), dim3(256), 0, stream, H, (int)n, decay, scale, cut, partial, table);
    GANQ_LAUNCH_CHECK();
    return 0;
}

}  // namespace ganq

/* tests (no GPU needed): the cut of ganq_hessian_accum_t for in_features n, `rows` tokens on `ncu` workgroups, by the code the kernels
 * run.  hdr[7] = {Ks, G, W, R, nprim, P, Sh}; segs = [ncu][W + 3][5] ints {tile, s0, s1, parts, slot}, unused entries tile = -1;
 * parts_loc = [R][max_parts][2] ints {workgroup, slot} of every left-over tile's parts in the order they are summed (-1 padded). */
extern "C" int ganq_debug_hessian_t_cut(int64_t n, int64_t rows, int ncu, int* hdr, int* segs, int segs_cap, int* parts_loc, int max_parts) {
    using namespace ganq;
    if (n <= 0 || rows <= 0 || (rows % WSK) != 0 || ncu <= 0 || !hdr) return fail(-1, "ganq_debug_hessian_t_cut: bad arguments");
    const int64_t nt = (n + WT - 1) / WT;
    const HwCut q = hw_make_cut(nt * (nt + 1) / 2, rows / WSK, ncu);
    const int h[7] = {q.Ks, q.G, q.W, q.R, q.nprim, q.P, q.Sh};
    for (int i = 0; i < 7; ++i) hdr[i] = h[i];
    const int per = q.W + 3;
    if (segs) {
        if (segs_cap < ncu * per * 5) return fail(-4, "ganq_debug_hessian_t_cut: segs too small");
        for (int c = 0; c < ncu; ++c) {
            HwSeg left[3];
            const int nl = hw_left_segments(q, c, left);
            for (int k = 0; k < per; ++k) {
                HwSeg sg{-1, 0, 0, 0, 0};
                if (k < q.W) sg = HwSeg{k * q.G + c, 0, q.Ks, 1, 0};
                else if (k - q.W < nl) sg = left[k - q.W];
                int* o = segs + ((size_t)c * per + k) * 5;
                o[0] = sg.t; o[1] = sg.s0; o[2] = sg.s1; o[3] = sg.np; o[4] = sg.slot;
            }
        }
    }
    if (parts_loc) {
        for (int r = 0; r < q.R; ++r) {
            const int np = hw_parts_of(q, r);
            for (int pi = 0; pi < max_parts; ++pi) {
                int wg = -1, sl = -1;
                if (pi < np) hw_part_location(q, r, pi, wg, sl);
                parts_loc[((size_t)r * max_parts + pi) * 2] = wg;
                parts_loc[((size_t)r * max_parts + pi) * 2 + 1] = sl;
            }
        }
    }
    return 0;
}

#ifdef HW_PROBE
extern "C" int ganq_debug_hess_w4_probe_all(unsigned long long* out) {
    GANQ_HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(ganq::hw_probe_all), 4096 * sizeof(unsigned long long)));
    static unsigned long long z[4096];
    GANQ_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(ganq::hw_probe_all), z, sizeof(z)));
    return 0;
}
extern "C" int ganq_debug_hess_w4_probe(unsigned long long* out, int reset) {
    GANQ_HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(ganq::hw_probe_buf), 8 * sizeof(unsigned long long)));
    if (reset) { unsigned long long z[8] = {}; GANQ_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(ganq::hw_probe_buf), z, sizeof(z))); }
    return 0;
}
#endif

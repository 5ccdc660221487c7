// S-solve: the back-substitution assignment loop of GANQ (reference ganq.py:533-565; Metal
// kernel compute_s ganq.py:94-247).  Arithmetic contract: include/ganq_hip.h + oracle/ganq_oracle.c
// (ganq_oracle_solve_s) -- bit-exact indices.
//
// Decomposition (one workgroup = 16 rows of W for the whole solve, no inter-workgroup traffic; 8 waves in two roles,
// see solve_s_kernel):
//   columns are processed right-to-left in panels of 64.  For panel [j0, j0+64):
//   (G) left-looking residual GEMM on the fp32 matrix cores: R[16, 64] = Err[16, j0+64:n] @ L[j0+64:n, j0:j0+64],
//       one v_mfma_f32_16x16x4_f32 accumulation chain per output, k running over columns in
//       DESCENDING order (wave w owns panel columns 16w..16w+15);
//   (P) the 64 sequential steps of the panel: every 16-lane DPP row holds one row of W, lane v holds
//       codebook entry T[row][v]; argmin / select are 16-lane DPP all-reductions (first-minimum
//       tie-break), the in-panel rank-1 residual update is 4 fmaf per lane with the panel's
//       triangle of L read from LDS.
//   Err (= W - T[Q]) is kept in a per-tile transposed scratch ErrT[tile][col][16 rows] so that the
//   A operand of (G) is one coalesced 256 B read.
#include <algorithm>
#include <atomic>
#include <type_traits>
#include <utility>

#include "common.h"
#include "solve_s.h"

namespace ganq {

#ifndef GANQ_SOLVE_SB
#define GANQ_SOLVE_SB 64
#endif
constexpr int SB = GANQ_SOLVE_SB;  // panel width (columns): 64, or 48 (three 16-column tiles: the residual chain on three SIMDs,
                                   // the column steps alone on the fourth -- see solve_s_kernel)
static_assert(SB == 64 || SB == 48, "panel width");
constexpr int SKR = SB / 16;       // 16-column tiles per panel = panel columns per lane of (P) = chain waves of (G)
constexpr int SKG = SB / 4;        // k-groups (MFMAs) per source panel
constexpr int SBLK = SB * 16;      // floats per packed operand block
constexpr int SBLKB = SBLK * 4;    // ... bytes
constexpr bool SPLIT = SB == 48;   // the two roles on separate SIMDs
#ifndef GANQ_SOLVE_PREFETCH
#define GANQ_SOLVE_PREFETCH 1
#endif
constexpr bool SOLVE_PF = GANQ_SOLVE_PREFETCH != 0 && !SPLIT;  // a ninth wave that pulls the next step's blocks of L into L2
constexpr int SOLVE_THREADS = SPLIT ? 13 * 64 : (SOLVE_PF ? 576 : 512);
constexpr int SR = 16;   // rows per workgroup
#ifndef GANQ_SOLVE_RING
#define GANQ_SOLVE_RING 3
#endif
constexpr int SOLVE_LDS_PANELS = SB == 64 ? 28 : 36;  // packed Err blocks kept in LDS (28 x 4 KB / 37 x 3 KB = 112 KB next to the panel buffers)

template <int CTRL>
__device__ __forceinline__ uint32_t dpp_u(uint32_t x) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, 0xf, 0xf, true);
}
template <int CTRL>
__device__ __forceinline__ float dpp_f(float x) {
    return __builtin_bit_cast(float, dpp_u<CTRL>(__builtin_bit_cast(uint32_t, x)));
}
// all-reductions across the 16 lanes of a DPP row: xor1, xor2 (quad_perm), row_half_mirror, row_mirror
__device__ __forceinline__ uint32_t row_min_u(uint32_t x) {
    x = min(x, dpp_u<0xB1>(x));
    x = min(x, dpp_u<0x4E>(x));
    x = min(x, dpp_u<0x141>(x));
    x = min(x, dpp_u<0x140>(x));
    return x;
}
__device__ __forceinline__ uint32_t row_or_u(uint32_t x) {
    x |= dpp_u<0xB1>(x);
    x |= dpp_u<0x4E>(x);
    x |= dpp_u<0x141>(x);
    x |= dpp_u<0x140>(x);
    return x;
}

// value for the lanes (l & 15) == OWN of every 16-lane row, `old` elsewhere: the lane mask is a compile-time constant, so
// it is materialised by two scalar moves where it is used instead of living in a register pair for the whole panel (the
// 16 masks of a panel, kept live, were most of the kernel's scalar-register pressure)
template <int OWN>
__device__ __forceinline__ uint32_t sel_own(uint32_t old, uint32_t nw) {
    const unsigned long long mask = 0x0001000100010001ull << OWN;
    uint32_t out;
    asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(out) : "v"(old), "v"(nw), "s"(mask));
    return out;
}

struct PanelState {
    float r[SKR];     // running residual sums of this lane's panel columns (c16 + 16k)
    float w[SKR];     // W[row][j0 + c16 + 16k]
    float e[SKR];     // err captured at this lane's columns
    uint32_t q[SKR];  // index captured at this lane's columns
    float tv;       // T[row][c16] (+inf beyond V)
    uint32_t c16;
    uint32_t pad;   // all ones in the lanes beyond V: their distance key can never win (inf - inf there would be a NaN)
};

// ---- threshold form of the argmin (fast path of the panel steps) ---------------------------------------------------
// The step's cost is its DPP instructions (a lone wave issues one every ~12 cycles, and they do not overlap with the
// other role's MFMAs on the SIMD): the three 16-lane reductions of panel_step are 12 of its 14.  With the row's codebook
// SORTED, lane s holding the s-th smallest value t_s (and its original index), the first minimum of |eff - T[v]| is
// decided by per-lane thresholds instead: lo_s / hi_s are the exact fp32 switch-over points between t_s and its
// neighbours, found once per launch by bisection on the kernel's own predicate
//     "the right neighbour wins":  |RN(x - b)| < |RN(x - a)|,  or equal and b has the smaller ORIGINAL index
// (monotone in x: RN is monotone), and lane s is the argmin iff  lo_s < eff <= hi_s.  That needs no cross-lane step; only
// T[idx] is then broadcast by ONE reduction (6 DPP per step instead of 14).  Equal values are merged (lowest original
// index represents them).  The argument needs every distance to be rounded without collisions, which holds while
// |eff| <= xb := 2^22 * (smallest gap between distinct values) - max|T|; a step outside that range -- or a NaN, or a row
// whose codebook has near-duplicates -- selects no lane or several, the wave notices (popcount of the selection != 4
// rows) and the whole panel is redone by the reduction path.  Results are bit-identical either way
// (tests/test_hip_stages.py: unsorted codebooks, duplicates, exact mid-point ties, out-of-range residuals).
typedef uint32_t __attribute__((address_space(3))) lds_u32_t;
typedef uint8_t __attribute__((address_space(3))) lds_u8_t;

#ifdef GANQ_SOLVE_TRACE
// developer timeline: s_memtime stamps of workgroup GANQ_SOLVE_TRACE_WG, wave 0 (role P) and wave 4 (role G), per step
__device__ unsigned long long g_solve_trace[2][320][6];
#define GANQ_TRACE(role, s, k) do { if (blockIdx.x == GANQ_SOLVE_TRACE_WG && lane == 0 && gw == 0 && (s) < 320) \
        g_solve_trace[role][s][k] = __builtin_amdgcn_s_memtime(); } while (0)
#ifndef GANQ_SOLVE_TRACE_WG
#define GANQ_SOLVE_TRACE_WG 0
#endif
#else
#define GANQ_TRACE(role, s, k) do {} while (0)
#endif
#ifdef GANQ_SOLVE_DEBUG
__device__ unsigned long long g_solve_dbg[4];  // [0] panels on the fast path, [1] panels redone, [2] waves without fast path
#endif

struct FastRow {
    float tv;       // sorted value held by this lane (t_s)
    float lo, hi;   // lane s is the argmin iff lo <= eff <= hi (closed, already clamped to [-xb, xb]); lo = hi = +inf: never
    float xb;       // > 0: the row can use the thresholds
    uint32_t orig;  // original index of t_s
};

__device__ __forceinline__ uint32_t f2ord(float f) {  // order-preserving map float -> uint32
    const uint32_t b = __builtin_bit_cast(uint32_t, f);
    return b ^ ((b >> 31) ? 0xffffffffu : 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t o) {
    const uint32_t b = o ^ ((o >> 31) ? 0x80000000u : 0xffffffffu);
    return __builtin_bit_cast(float, b);
}
// does b (right neighbour, original index ob) beat a (original index oa) at x, by the kernel's own comparison?
__device__ __forceinline__ bool right_wins(float x, float a, uint32_t oa, float b, uint32_t ob) {
    const uint32_t da = __builtin_bit_cast(uint32_t, x - a) & 0x7fffffffu;
    const uint32_t db = __builtin_bit_cast(uint32_t, x - b) & 0x7fffffffu;
    return db < da || (db == da && ob < oa);
}

// rotation by K lanes within the 16-lane DPP row (row_ror); value and lane index are rotated together, so the direction
// does not matter to the caller
template <int K>
__device__ __forceinline__ uint32_t row_rot(uint32_t x) {
    return dpp_u<0x120 + K>(x);
}

// One-time setup per launch: sort the row's codebook across its 16 lanes, merge equal values, find the thresholds.
// `scratch` is 16 x 2 dwords of LDS private to this 16-lane row.  Returns false (for the whole row) when the row cannot
// use the fast path.
__device__ __forceinline__ FastRow fast_row_setup(float tv_orig, uint32_t c16, volatile lds_u32_t* scratch) {
    // rank by (value, original index): strict total order, stable; -0 and +0 are one value (they compare equal)
    uint32_t rank = 0;
    const uint32_t myo = f2ord(tv_orig == 0.0f ? 0.0f : tv_orig);
    auto count = [&](uint32_t oo, uint32_t ol) { rank += (oo < myo || (oo == myo && ol < c16)) ? 1u : 0u; };
#define GANQ_ROT(K) count(row_rot<K>(myo), row_rot<K>(c16))
    GANQ_ROT(1); GANQ_ROT(2); GANQ_ROT(3); GANQ_ROT(4); GANQ_ROT(5); GANQ_ROT(6); GANQ_ROT(7); GANQ_ROT(8);
    GANQ_ROT(9); GANQ_ROT(10); GANQ_ROT(11); GANQ_ROT(12); GANQ_ROT(13); GANQ_ROT(14); GANQ_ROT(15);
#undef GANQ_ROT
    scratch[2 * rank] = __builtin_bit_cast(uint32_t, tv_orig);
    scratch[2 * rank + 1] = c16;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    FastRow fr;
    fr.tv = __builtin_bit_cast(float, (uint32_t)scratch[2 * c16]);
    fr.orig = scratch[2 * c16 + 1];
    // neighbours in sorted order (lane c16 - 1 / c16 + 1); row ends see themselves
    const float tprev = __builtin_bit_cast(float, (uint32_t)scratch[2 * (c16 > 0 ? c16 - 1 : 0)]);
    // representative of a run of equal values = its first lane (lowest original index, by the sort order)
    const bool rep = c16 == 0 || !(tprev == fr.tv);
    // next DISTINCT value and the original index of its representative (inf entries beyond V are ordinary values here:
    // they are never chosen while eff is finite, and a non-finite eff leaves the fast path)
    float tnext = __builtin_inff();
    uint32_t onext = 0;
    bool has_next = false;
    for (int s2 = 15; s2 >= 1; --s2) {  // scan from the far end so that the nearest distinct value is kept last
        const int pos = (int)c16 + s2;
        if (pos < 16) {
            const float cand = __builtin_bit_cast(float, (uint32_t)scratch[2 * pos]);
            if (cand > fr.tv) {
                tnext = cand;
                onext = scratch[2 * pos + 1];
                has_next = true;
            }
        }
    }
    // the representative of the next distinct value is its FIRST lane: the scan above ends on the nearest position
    // greater than tv, which is that first lane
    float hi = __builtin_inff();
    if (has_next && fr.tv == fr.tv && tnext < __builtin_inff()) {
        // largest x in [tv, tnext] at which the right neighbour does not win yet (it does not at x = tv, it does at tnext)
        uint32_t lo_o = f2ord(fr.tv), hi_o = f2ord(tnext);
        while (hi_o - lo_o > 1u) {
            const uint32_t mid = lo_o + ((hi_o - lo_o) >> 1);
            if (right_wins(ord2f(mid), fr.tv, fr.orig, tnext, onext)) hi_o = mid; else lo_o = mid;
        }
        hi = ord2f(lo_o);
    }
    // lo of lane s = hi of the previous DISTINCT value's representative; pass it along through the scratch
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    scratch[2 * c16] = __builtin_bit_cast(uint32_t, hi);
    scratch[2 * c16 + 1] = rep ? 1u : 0u;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    float lo = -__builtin_inff();
    for (int pos = 0; pos < 16; ++pos)  // the nearest representative to the left
        if (pos < (int)c16 && scratch[2 * pos + 1] != 0u) lo = __builtin_bit_cast(float, (uint32_t)scratch[2 * pos]);
    // gaps between distinct FINITE values and the largest finite magnitude -> range in which no two distances collide
    const bool finite = fr.tv < __builtin_inff() && fr.tv > -__builtin_inff();
    float gap = __builtin_inff();
    if (rep && finite && has_next && tnext < __builtin_inff()) gap = tnext - fr.tv;
    uint32_t gmin = __builtin_bit_cast(uint32_t, gap), amax = finite ? (__builtin_bit_cast(uint32_t, fr.tv) & 0x7fffffffu) : 0u;
    gmin = row_min_u(gmin);                       // positive floats order like their bits
    amax = 0xffffffffu - row_min_u(0xffffffffu - amax);
    const float g = __builtin_bit_cast(float, gmin), a = __builtin_bit_cast(float, amax);
    fr.xb = (g < __builtin_inff()) ? (4194304.0f * g - a) : 3.0e38f;  // one distinct value: nothing can collide
    const bool nan_any = row_or_u((tv_orig != tv_orig) ? 1u : 0u) != 0u;
    if (nan_any || !(fr.xb > 0.0f)) fr.xb = -1.0f;                    // never satisfied: the row stays on the slow path
    // closed interval [next float above lo, hi], clamped to the collision-free range; an empty or unusable one becomes
    // [+inf, +inf], which no finite eff lies in (one v_med3 + one compare per step test the membership)
    float lo_c = lo == -__builtin_inff() ? -fr.xb : fmaxf(ord2f(f2ord(lo) + 1u), -fr.xb);
    float hi_c = fminf(hi, fr.xb);
    if (!rep || !(fr.tv < __builtin_inff()) || !(lo_c <= hi_c) || !(fr.xb > 0.0f)) {
        lo_c = __builtin_inff();  // merged duplicates, the +inf padding, rows without a usable range
        hi_c = __builtin_inff();
    }
    fr.lo = lo_c;
    fr.hi = hi_c;
    return fr;
}

// One column step on the fast path.  nsel counts this lane's selections: the intervals of a row are disjoint, so "every
// step selected exactly one lane" is equivalent to "the row's selections add up to the number of steps", which the
// caller checks once per panel (no scalar work inside the step).
template <int JJ>
__device__ __forceinline__ void panel_step_fast(PanelState& st, const FastRow& fr, const float2 dg, const float4 lrow,
                                                uint32_t qaddr, uint32_t qdummy, uint32_t& nsel) {
    constexpr int KREG = JJ >> 4, OWN = JJ & 15;
    const float q0 = st.r[KREG] * dg.y;
    const float qe = fmaf(-q0, dg.x, st.r[KREG]);
    const float quo = fmaf(qe, dg.y, q0);
    const float eff_l = st.w[KREG] + quo;
    const float eff = dpp_f<0x150 + OWN>(eff_l);
    const float wj = dpp_f<0x150 + OWN>(st.w[KREG]);
    const bool sel = __builtin_amdgcn_fmed3f(eff, fr.lo, fr.hi) == eff;  // lo <= eff <= hi (false for NaN)
    const uint32_t tb = row_or_u(sel ? __builtin_bit_cast(uint32_t, fr.tv) : 0u);
    const float err = wj - __builtin_bit_cast(float, tb);
    st.r[0] = fmaf(err, lrow.x, st.r[0]);
    st.r[1] = fmaf(err, lrow.y, st.r[1]);
    st.r[2] = fmaf(err, lrow.z, st.r[2]);
    if constexpr (SKR > 3) st.r[SKR - 1] = fmaf(err, lrow.w, st.r[SKR - 1]);
    nsel += sel ? 1u : 0u;
    // the selected lane files the ORIGINAL index of its value; the others write a dummy byte (no exec-mask change)
    *reinterpret_cast<lds_u8_t*>((uintptr_t)(sel ? qaddr + JJ : qdummy)) = (uint8_t)fr.orig;
    st.e[KREG] = __builtin_bit_cast(float, sel_own<OWN>(__builtin_bit_cast(uint32_t, st.e[KREG]), __builtin_bit_cast(uint32_t, err)));
}

template <bool FULL, int JJ>
__device__ __forceinline__ void panel_from_fast(PanelState& st, const FastRow& fr, const float4 (*Ld)[16], const float2* Dg, int wd,
                                                float2 dg, float4 lrow, uint32_t qaddr, uint32_t qdummy, uint32_t& nsel) {
    float2 dg_n = dg;
    float4 lrow_n = lrow;
    if constexpr (JJ > 0) {
        dg_n = Dg[JJ - 1];
        lrow_n = Ld[JJ - 1][st.c16];
    }
    if (FULL || JJ < wd) panel_step_fast<JJ>(st, fr, dg, lrow, qaddr, qdummy, nsel);
    if constexpr (JJ > 0) panel_from_fast<FULL, JJ - 1>(st, fr, Ld, Dg, wd, dg_n, lrow_n, qaddr, qdummy, nsel);
}

// One column step.  JJ = column inside the panel; its owner is lane (JJ & 15) of each DPP row, register JJ >> 4.
// dg = {L[j][j], RN(1 / L[j][j])} and lrow = L[j][j0 + c16 + 16k] (k = 0..3) were read from LDS one step earlier.
template <int JJ>
__device__ __forceinline__ void panel_step(PanelState& st, const float2 dg, const float4 lrow) {
    constexpr int KREG = JJ >> 4, OWN = JJ & 15;
    // r / L[j][j] as an exactly rounded quotient without the division sequence (Markstein): rinv = RN(1/L[j][j]) is
    // computed once per column with a true division; q0 = RN(r*rinv); e = fma(-q0, L, r) is exact; RN(q0 + e*rinv) is
    // the IEEE quotient (checked against the division by ganq_debug_div_check and, end to end, by the oracle tests)
    const float q0 = st.r[KREG] * dg.y;
    const float qe = fmaf(-q0, dg.x, st.r[KREG]);
    const float quo = fmaf(qe, dg.y, q0);
    const float eff_l = st.w[KREG] + quo;
    const float eff = dpp_f<0x150 + OWN>(eff_l);  // row_newbcast: owner lane -> its 16-lane row
    const float wj = dpp_f<0x150 + OWN>(st.w[KREG]);
    // |eff - T[v]| as ordered bits, shifted up by one so that key 0 is free for a NaN distance: torch.argmin (ganq.py:547)
    // treats a NaN as smaller than everything and returns the first one (pinned by the reference golden
    // tests/golden/large/nan48x256_b4_k1.npz); only this reduction path ever sees one -- a row with a NaN codebook entry
    // never qualifies for the threshold path, and a NaN residual fails its range test
    const uint32_t d0 = __builtin_bit_cast(uint32_t, eff - st.tv) & 0x7fffffffu;
    const uint32_t d = (d0 > 0x7f800000u ? 0u : d0 + 1u) | st.pad;
    const uint32_t dmin = row_min_u(d);
    const uint32_t cand = (d == dmin) ? st.c16 : 255u;
    const uint32_t idx = row_min_u(cand);  // first minimum
    // (Taking T[idx] as the OR over the minimal lanes -- exact whenever one lane per row attains the minimum, with a
    // wave-uniform branch to this path on ties -- moves the index reduction off the dependent chain but adds six
    // instructions; measured slower, 0.540 vs 0.523 ms for the P role alone: the step is bound by the number of DPP /
    // fp32 instructions a lone wave issues, not by the length of the chain.)
    const uint32_t tb = row_or_u((st.c16 == idx) ? __builtin_bit_cast(uint32_t, st.tv) : 0u);
    const float err = wj - __builtin_bit_cast(float, tb);
    st.r[0] = fmaf(err, lrow.x, st.r[0]);
    st.r[1] = fmaf(err, lrow.y, st.r[1]);
    st.r[2] = fmaf(err, lrow.z, st.r[2]);
    if constexpr (SKR > 3) st.r[SKR - 1] = fmaf(err, lrow.w, st.r[SKR - 1]);
    st.q[KREG] = sel_own<OWN>(st.q[KREG], idx);
    st.e[KREG] = __builtin_bit_cast(float, sel_own<OWN>(__builtin_bit_cast(uint32_t, st.e[KREG]), __builtin_bit_cast(uint32_t, err)));
}

// steps run from the panel's last column down to its first; the LDS operands of step JJ-1 are fetched before the
// dependent chain of step JJ starts (they depend on nothing the steps compute)
template <bool FULL, int JJ>
__device__ __forceinline__ void panel_from(PanelState& st, const float4 (*Ld)[16], const float2* Dg, int wd, float2 dg,
                                           float4 lrow) {
    float2 dg_n = dg;
    float4 lrow_n = lrow;
    if constexpr (JJ > 0) {
        dg_n = Dg[JJ - 1];
        lrow_n = Ld[JJ - 1][st.c16];
    }
    if (FULL || JJ < wd) panel_step<JJ>(st, dg, lrow);
    if constexpr (JJ > 0) panel_from<FULL, JJ - 1>(st, Ld, Dg, wd, dg_n, lrow_n);
}

template <bool FULL, int... I>
__device__ __forceinline__ void panel_all(PanelState& st, const float4 (*Ld)[16], const float2* Dg, int wd,
                                          std::integer_sequence<int, I...>) {
    panel_from<FULL, SB - 1>(st, Ld, Dg, wd, Dg[SB - 1], Ld[SB - 1][st.c16]);
}

// Two roles per workgroup (8 waves, one of each role per SIMD):
//   waves 0-3 (P) run the sequential steps of panel b+1 while
//   waves 4-7 (G) run the residual chain of panel b over every column right of panel b+1 (part 1); after the barrier
//   the G waves append the 64 columns of panel b+1 (part 2: Err handed over in LDS, the L block prefetched into
//   registers) and publish R for panel b.  The chain of an output is still ONE accumulator running over the columns
//   in descending order -- only who computes when has changed.  G also stages the next panel's triangle of L and
//   its diagonal (double-buffered), P prefetches its next W columns.
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4raw __attribute__((__vector_size__(16)));  // what the 16-byte buffer builtins carry

#ifndef GANQ_MFMA_INPLACE
#define GANQ_MFMA_INPLACE 1  // measured: 4096 x 4096 1.065 -> 1.039 ms, 2048 x 8192 2.91 -> 2.74 ms
#endif
__device__ __forceinline__ void mfma_inplace(f32x4& acc, float a, float b) {
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}

// keeps a stage of the chain loop where it was written: the memory clobber stops IR-level load motion across stage
// boundaries, the sched_barrier stops the machine scheduler
#define GANQ_PIN() do { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

// ---------------------------------------------------------------------------------------------------------------
// One-wave form of the column steps (split layout): 16 rows x 4 lanes.  Lane (row, sub) owns the panel columns
// sub + 4k (k = 0 .. SB/4 - 1) and four entries of the row's SORTED codebook with their thresholds (4 sub + i).  A step is
// the same arithmetic as panel_step_fast / panel_step -- exact quotient, threshold test or first-minimum argmin, err =
// w - T[idx], rank-1 update -- with quad_perm broadcasts and 2-stage quad reductions instead of the 16-lane ones, and it
// serves all 16 rows with one instruction stream (about 35 vector instructions per column step instead of 4 x 22).
constexpr int P4K = SB / 4;
struct P4State {
    float r[P4K];   // running residual sums of this lane's panel columns
    float tc[P4K];  // T[idx] captured at this lane's columns (err = w - tc, recomputed after the panel)
    // (the weights W[row][j0 + sub + 4k] stay in LDS, where the helper waves staged them: one read per step, a step ahead)
};
struct P4Fast {
    float tv[4], lo[4], hi[4];
    uint32_t orig[4];
};
__device__ __forceinline__ uint32_t quad_or_u(uint32_t x) {
    x |= dpp_u<0xB1>(x);
    x |= dpp_u<0x4E>(x);
    return x;
}
__device__ __forceinline__ uint32_t quad_min_u(uint32_t x) {
    x = min(x, dpp_u<0xB1>(x));
    x = min(x, dpp_u<0x4E>(x));
    return x;
}
template <int OWN>
__device__ __forceinline__ uint32_t sel_own4(uint32_t old, uint32_t nw) {  // lanes (l & 3) == OWN take nw
    const unsigned long long mask = 0x1111111111111111ull << OWN;
    uint32_t out;
    asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(out) : "v"(old), "v"(nw), "s"(mask));
    return out;
}

template <int JJ>
__device__ __forceinline__ void p4_step_fast(P4State& st, const P4Fast& f, const float2 dg, const float wk, const float (&l)[P4K],
                                             uint32_t qaddr, uint32_t qdummy, uint32_t& nsel) {
    constexpr int K = JJ >> 2, OWN = JJ & 3;
    const float q0 = st.r[K] * dg.y;
    const float qe = fmaf(-q0, dg.x, st.r[K]);
    const float quo = fmaf(qe, dg.y, q0);
    const float eff_l = wk + quo;
    const float eff = dpp_f<OWN * 0x55>(eff_l);  // quad_perm: the owner lane of the quad
    const float wj = dpp_f<OWN * 0x55>(wk);
    // the intervals are disjoint: at most one lane of the quad, and one entry in it, selects.  Four INDEPENDENT tests and an
    // OR tree (a chain of selects through one condition register made the step four times as long)
    uint32_t tq[4], oq[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const bool sel = __builtin_amdgcn_fmed3f(eff, f.lo[i], f.hi[i]) == eff;
        tq[i] = sel ? __builtin_bit_cast(uint32_t, f.tv[i]) : 0u;
        oq[i] = sel ? f.orig[i] : 0u;  // orig holds index + 1: 0 = not selected
    }
    const uint32_t t = (tq[0] | tq[1]) | (tq[2] | tq[3]);
    const uint32_t o = (oq[0] | oq[1]) | (oq[2] | oq[3]);
    const uint32_t tb = quad_or_u(t);
    const float err = wj - __builtin_bit_cast(float, tb);
#pragma unroll
    for (int k = 0; k <= K; ++k) st.r[k] = fmaf(err, l[k], st.r[k]);
    const bool any = o != 0u;
    nsel += any ? 1u : 0u;
    // (the dummy region has SB bytes per lane: the step's offset is an immediate of the store, not 48 hoisted addresses)
    reinterpret_cast<lds_u8_t*>((uintptr_t)(any ? qaddr : qdummy))[JJ] = (uint8_t)(o - 1u);
    st.tc[K] = __builtin_bit_cast(float, sel_own4<OWN>(__builtin_bit_cast(uint32_t, st.tc[K]), tb));
}

// the reduction form (first minimum of |eff - T[v]| over the ORIGINAL order: lane sub holds entries 4 sub + i)
template <int JJ>
__device__ __forceinline__ void p4_step_slow(P4State& st, const float (&To)[4], uint32_t sub, const float2 dg, const float wk,
                                             const float (&l)[P4K], uint32_t qaddr, uint32_t qdummy) {
    constexpr int K = JJ >> 2, OWN = JJ & 3;
    const float q0 = st.r[K] * dg.y;
    const float qe = fmaf(-q0, dg.x, st.r[K]);
    const float quo = fmaf(qe, dg.y, q0);
    const float eff_l = wk + quo;
    const float eff = dpp_f<OWN * 0x55>(eff_l);
    const float wj = dpp_f<OWN * 0x55>(wk);
    uint32_t m = __builtin_bit_cast(uint32_t, eff - To[0]) & 0x7fffffffu, li = 0u;
    float tl = To[0];
#pragma unroll
    for (int i = 1; i < 4; ++i) {
        const uint32_t d = __builtin_bit_cast(uint32_t, eff - To[i]) & 0x7fffffffu;
        const bool less = d < m;  // strict: the first minimum stays
        m = less ? d : m;
        li = less ? (uint32_t)i : li;
        tl = less ? To[i] : tl;
    }
    const uint32_t mq = quad_min_u(m);
    const uint32_t cand = (m == mq) ? (4u * sub + li) : 255u;
    const uint32_t idx = quad_min_u(cand);
    const uint32_t tb = quad_or_u((cand == idx) ? __builtin_bit_cast(uint32_t, tl) : 0u);
    const float err = wj - __builtin_bit_cast(float, tb);
#pragma unroll
    for (int k = 0; k <= K; ++k) st.r[k] = fmaf(err, l[k], st.r[k]);
    reinterpret_cast<lds_u8_t*>((uintptr_t)(sub == (uint32_t)OWN ? qaddr : qdummy))[JJ] = (uint8_t)idx;  // one lane per row files it
    st.tc[K] = __builtin_bit_cast(float, sel_own4<OWN>(__builtin_bit_cast(uint32_t, st.tc[K]), tb));
}

// L[j0 + jj][j0 + sub + 4k], k = 0 .. P4K-1, of the step: three 16-byte reads of the lane's slice of the row
template <int JJ>
__device__ __forceinline__ void p4_load_l(const float* ld4, uint32_t sub, float (&l)[P4K]) {
    constexpr int jj = JJ;
    const float4* src = reinterpret_cast<const float4*>(ld4 + (jj * 4 + (int)sub) * P4K);
#pragma unroll
    for (int v = 0; v <= (JJ >> 2) / 4; ++v) {  // the step updates the columns k <= JJ >> 2 only
        const float4 x = src[v];
        l[4 * v] = x.x;
        l[4 * v + 1] = x.y;
        l[4 * v + 2] = x.z;
        l[4 * v + 3] = x.w;
    }
}

template <bool FULL, int JJ>
__device__ __forceinline__ void p4_from_fast(P4State& st, const P4Fast& f, const float* ld4, const float2* Dg, const float* wrow,
                                             uint32_t sub, int wd, float2 dg, float wk, const float (&l)[P4K], uint32_t qaddr,
                                             uint32_t qdummy, uint32_t& nsel) {
    float2 dg_n = dg;
    float wk_n = wk;
    float ln[P4K] = {};
    if constexpr (JJ > 0) {
        dg_n = Dg[JJ - 1];
        wk_n = wrow[4 * ((JJ - 1) >> 2)];
        p4_load_l<JJ - 1>(ld4, sub, ln);
    }
    if (FULL || JJ < wd) p4_step_fast<JJ>(st, f, dg, wk, l, qaddr, qdummy, nsel);
    __builtin_amdgcn_sched_barrier(0);  // keep the steps apart: hoisted operand loads of later steps cost registers the wave lacks
    if constexpr (JJ > 0) p4_from_fast<FULL, JJ - 1>(st, f, ld4, Dg, wrow, sub, wd, dg_n, wk_n, ln, qaddr, qdummy, nsel);
}
template <bool FULL, int JJ>
__device__ __forceinline__ void p4_from_slow(P4State& st, const float (&To)[4], const float* ld4, const float2* Dg, const float* wrow,
                                             uint32_t sub, int wd, float2 dg, float wk, const float (&l)[P4K], uint32_t qaddr,
                                             uint32_t qdummy) {
    float2 dg_n = dg;
    float wk_n = wk;
    float ln[P4K] = {};
    if constexpr (JJ > 0) {
        dg_n = Dg[JJ - 1];
        wk_n = wrow[4 * ((JJ - 1) >> 2)];
        p4_load_l<JJ - 1>(ld4, sub, ln);
    }
    if (FULL || JJ < wd) p4_step_slow<JJ>(st, To, sub, dg, wk, l, qaddr, qdummy);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (JJ > 0) p4_from_slow<FULL, JJ - 1>(st, To, ld4, Dg, wrow, sub, wd, dg_n, wk_n, ln, qaddr, qdummy);
}

// ---------------------------------------------------------------------------------------------------------------
// Packed operand blocks.  The chain consumes, per (source panel p of 64 columns, 16-wide output tile), 16 k-groups of
// 4 columns; lane (ksub = lane >> 4, c16 = lane & 15) of the MFMA needs for k-group g the element with
// u = 64 p + 4 g + kslot(ksub).  A block stores those 16 values of a lane contiguously:
//     block[(g >> 2) * 256 + lane * 4 + (g & 3)]   (1024 floats = 4 KB; each of the four 16-byte loads of a wave reads
//                                                   1 KB contiguously, and LDS reads of it are bank-conflict free)
//   B blocks (L):   Lr[lr_block(ct, p) * SBLK + ..] = L[u][16 ct + c16]   built once per L by l_pack_kernel
//   A blocks (Err): ErrT[tile][p * 1024 + ..]     = Err[row c16][u]      written by the P waves
// so a batch of 16 MFMAs costs 4 + 4 sixteen-byte loads instead of 16 + 16 dword loads: the lone G wave of a SIMD
// shares its issue slots with the P wave, and every instruction saved there is matrix-core time gained.
// block of (output tile ct, source panel p) inside the packed L.  Tile-major: the blocks one chain reads one after the other
// (p descending, ct fixed) are neighbours in memory -- one sequential stream per wave instead of a stride of NT blocks
#ifndef GANQ_LR_PANEL_MAJOR
__host__ __device__ __forceinline__ int64_t lr_block(int ct, int p, int NT, int nb) { (void)NT; return (int64_t)ct * nb + p; }
#else
__host__ __device__ __forceinline__ int64_t lr_block(int ct, int p, int NT, int nb) { (void)nb; return (int64_t)p * NT + ct; }
#endif

__global__ __launch_bounds__(64) void l_pack_kernel(const float* __restrict__ L, int64_t ldl, int n, int NT, bool kasc,
                                                    float* __restrict__ Lr, uint32_t* __restrict__ ctrl, int ctrl_words) {
    const int ct = blockIdx.x, p = blockIdx.y;
    if (ct == 0 && p == 0) {  // the ticket counter and the flags of the helper workgroups start clean with every layer
        for (int i = threadIdx.x; i < ctrl_words; i += 64) ctrl[i] = 0u;
    }
    if (p <= ct / SKR) return;  // only source panels strictly right of the tile's panel are ever read
    const int lane = threadIdx.x, c16 = lane & 15, ksub = lane >> 4;
    const int kslot = kasc ? (3 - ksub) : ksub;
    const int col = 16 * ct + c16;
    f32x4v out[SKR];
#pragma unroll
    for (int g = 0; g < SKG; ++g) {
        const int u = SB * p + 4 * g + kslot;
        out[g >> 2][g & 3] = (u < n && col < n) ? L[(int64_t)u * ldl + col] : 0.0f;
    }
    f32x4v* dst = reinterpret_cast<f32x4v*>(Lr + lr_block(ct, p, NT, (int)gridDim.y) * SBLK + lane * 4);
#pragma unroll
    for (int j = 0; j < SKR; ++j) dst[j * 64] = out[j];
}

// Two roles per workgroup (8 waves, one of each role per SIMD):
//   waves 0-3 (P) run the sequential steps of panel b+1 while
//   waves 4-7 (G) run the residual chain of panel b over every column right of panel b+1 (part 1); after the barrier
//   the G waves append the 64 columns of panel b+1 (part 2: Err handed over in LDS, the L block prefetched into
//   registers) and publish R for panel b.  The chain of an output is still ONE accumulator running over the columns
//   in descending order -- only who computes when has changed.  G also stages the next panel's triangle of L and
//   its diagonal (double-buffered), P prefetches its next W columns.
// ---- helper workgroups ("duo" launches) ---------------------------------------------------------------------------------
// A tile's solve is a latency chain on ONE CU: its column steps and its residual chain share the SIMDs' issue slots and add up
// (see DESIGN "Where the S-solve stands").  When the launch has at most half as many active tiles as the chip has CUs -- layers
// of up to 2048 rows, the late iterations of taller ones (converged rows are skipped), row shards of a multi-GPU run -- every
// tile gets a second workgroup on another CU that computes the FAR part of each panel's chain: the chain of panel b runs over
// the source panels nb-1, nb-2, .., b+2 in this order; the helper takes the first h of them (they were solved long ago), hands
// the 16 x 64 accumulators over through memory, and the tile's own chain waves continue with the x nearest panels and the
// panel just solved.  Same MFMAs in the same order on the same accumulator values: bit-identical to the single-workgroup solve.
//   main -> helper: every packed Err block goes to the tile's scratch with agent-scope stores, then `solved` = step (one flag
//                   per tile, written by one lane after the step's barrier, i.e. after every wave's stores were acknowledged);
//   helper -> main: accumulators to Facc[tile][step][wave] with agent-scope stores, then `ready[wave]` = step.
// Flags carry a per-launch tag, so a stale flag of an earlier launch never matches; l_pack_kernel zeroes them once per layer.
// Roles come from a ticket drawn at start, not from blockIdx (one counter per blockIdx mod 8): even tickets of a class are tiles,
// the next odd one its helper -- whatever set of workgroups is resident, all but at most 8 tiles have their helper resident too,
// and those finish without waiting for anybody, so a helper that starts late only delays.  The tile never depends on it either: a chain
// wave that waits longer than DUO_TIMEOUT for accumulators computes the whole chain itself from then on (same bits).
// Launches with at most a third as many tiles as CUs ("trio") give a tile TWO helpers: the second takes the farthest half of
// the helpers' share and hands its accumulators to the first, which continues and hands over to the tile -- a pipeline of three
// CUs per chain; the tile's own share (trio_pol) is then small enough for its column steps to set the pace to the last panel.
constexpr int DUO_MAX_TILES = 128;            // tiles with a helper (scratch is sized for them)
constexpr int DUO_TRIO_TILES = 80;            // ... with two helpers each (a third of the chip)
constexpr int DUO_TW = 9;                     // flag words per tile: solved, ready[4] (helper -> tile), ready2[4] (second helper -> first)
constexpr int DUO_CTRL_WORDS = 8 + DUO_MAX_TILES * DUO_TW;  // 8 ticket counters, then the tiles' flags
// s_memrealtime ticks (constant 100 MHz): 0.5 ms, about 60 column-panel steps (6-8 us each; 25 us for the last steps of n = 16384)
// -- long enough for a helper that is resident but behind (it starts with a cold operand ring), short enough that a helper that is
// NOT resident (another launch holds the CUs: side streams, a second process) costs a tile one wait of the order of the kernel's
// own time.  (Rounds 1-3 counted 20 M ticks of s_memtime and called them 0.2 s; s_memtime ticks are shader cycles, so that was
// ~9 ms -- still 10 x a whole launch.)  The helpers' own waits are 16 x this.
constexpr unsigned long long DUO_TIMEOUT = 50ull * 1000;
// number of source panels the tile's own waves keep, of the c = s - 1 panels of step s's chain (the nearest ones);
// pol = xa | xb << 8 | xmin << 16 | cmin << 24:  x = max(xmin, xa c / 64 - xb) from c >= cmin on, everything below
__host__ __device__ __forceinline__ int duo_near(int c, int pol) {
    const int xa = pol & 255, xb = (pol >> 8) & 255, xmin = (pol >> 16) & 255, cmin = (pol >> 24) & 127;
    if (pol == 0 || c < cmin) return c;
    int x = ((xa * c) >> 6) - xb;
    x = x > xmin ? x : xmin;
    return x < c ? x : c;
}

#ifndef GANQ_SOLVE_PRIME
#define GANQ_SOLVE_PRIME 0  // measured: the chains start 0.2-0.5 us earlier and part 2 takes 0.3 us longer -- the chain waves are not
                            // the critical path of most steps; 4096 x 4096 1.05 vs 1.04 ms.  Off.  (The primed head assumes a chain
                            // that starts at the top panel: a build with it switches the helper workgroups off.)
#endif
template <bool KASC>
__global__ __launch_bounds__(SOLVE_THREADS) void solve_s_kernel(const float* __restrict__ W, const float* __restrict__ L,
                                                      int64_t ldl, const float* __restrict__ Lr, int NT,
                                                      const float* __restrict__ T, int m, int n, int V,
                                                      uint8_t* __restrict__ Q, float* __restrict__ ErrOut,
                                                      float* __restrict__ ErrT, int pbase, const int* __restrict__ rowlist,
                                                      const int* __restrict__ nactive, int opt_fast,
                                                      uint32_t* __restrict__ ctrl, float* __restrict__ Facc,
                                                      float* __restrict__ ErrH, uint32_t tag, int duo_pol, int ncu, int trio_pol,
                                                      int64_t facc_stride, int64_t errh_stride) {
    __shared__ float4 Ld[SPLIT ? 1 : 2][SPLIT ? 1 : SB][16];  // panel triangle of L, [jj][c16][k] <-> L[j0+jj][j0 + c16 + 16k]
    __shared__ __align__(16) float Ld4[SPLIT ? 2 * SB * SB : 4];  // split layout: [buf][jj][sub][k] <-> L[j0+jj][j0 + sub + 4k]
    __shared__ float Fs[SPLIT ? 3 : 1][SPLIT ? 16 : 1][16];    // split layout: sorted codebook, lo, hi of the 16 rows (set-up -> P wave)
    __shared__ uint32_t Fo[SPLIT ? 16 : 1][16];                // ... original indices
    __shared__ int Fok[16];                                    // ... row can use the thresholds
    __shared__ float Wp[SPLIT ? 2 : 1][SPLIT ? 16 : 1][SB];    // split layout: W of the next panel's columns, staged by the helper waves
    __shared__ float Rp[2][SR][SB + 4];   // residual panel handed from (G) to (P)
    __shared__ float2 Dg[2][SB];          // {L[j][j], 1 / L[j][j]} of the panel's columns
    __shared__ __align__(16) float ErrPk[SBLK];  // packed Err block of the panel just solved (zero beyond its width)
    __shared__ __align__(16) float Bp2[SPLIT ? SKR * SBLK : 4];  // split layout: the B operands of part 2 (packed L blocks of source
                                                               // panel bG+1, one per tile), staged by (P)
    __shared__ uint8_t Qst[4][4][SB];            // fast path: original index filed by the selected lane, per (P wave, row, step)
    __shared__ uint8_t Qdm4[SPLIT ? 64 : 1][SPLIT ? SB : 1];  // split layout: the same, SB bytes per lane (offset = step)
    __shared__ uint32_t Qdm[4][64];              // ... and where the lanes that were NOT selected put theirs: one dword slot per
                                                 // lane (60 lanes storing to ONE address would serialise in the LDS)
    extern __shared__ __align__(16) float ErrL[];  // packed Err blocks of the panels >= pbase, the A operand's hot part

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    // rowlist / nactive (device, may be null): solve only these rows (the loop driver passes the rows that have not
    // reached their fixed point yet); tile t then holds rows rowlist[16 t .. 16 t + 15]
    const int nact = nactive ? *nactive : m;
    int tile = blockIdx.x;
    bool duo = false, helper = false;
    int hstage = 0, nhelp = 1;  // this workgroup: 0 the tile, 1 / 2 its first / second helper; helpers per tile
    int pol = duo_pol;          // the split policy in force
    if constexpr (!SPLIT) {
        if (duo_pol != 0 && GANQ_SOLVE_PRIME == 0) {
            const int A = (min(nact, m) + SR - 1) / SR;  // active tiles
            duo = A <= DUO_MAX_TILES && 16 * ((A + 7) >> 3) <= min((int)gridDim.x, ncu);
            if (trio_pol != 0 && A <= DUO_TRIO_TILES && 24 * ((A + 7) >> 3) <= min((int)gridDim.x, ncu)) {
                duo = true;
                nhelp = 2;
                pol = trio_pol;
            }
            if (duo) {  // roles by ticket (see above); a launch without helpers keeps tile = blockIdx and draws nothing
                // one counter per residue class of blockIdx mod 8 (an XCD, when workgroups are dealt round-robin): 256 draws on one
                // address serialise for tens of microseconds, and a pair drawn from one class shares an L2
                __shared__ uint32_t s_tk;
                const int cls = (int)(blockIdx.x & 7u);
                if (tid == 0) {
                    const uint32_t t = __hip_atomic_fetch_add(&ctrl[cls], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const uint32_t in_class = (gridDim.x - (uint32_t)cls + 7u) >> 3;
                    // the last one of the class to draw: everybody has, the counter is clean for the next launch
                    if (t == in_class - 1) __hip_atomic_store(&ctrl[cls], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    s_tk = t;
                }
                __syncthreads();
                const int tk = (int)s_tk;
                tile = cls + 8 * (tk / (nhelp + 1));
                hstage = tk % (nhelp + 1);
                helper = hstage != 0;
            }
        }
    }
    if (tile * SR >= nact) return;  // uniform for the workgroup, before any other barrier
    bool roleG;
    int gw;  // index of the wave inside its role
    if constexpr (SPLIT) {
        // 13 waves are launched so that, with the hardware dealing consecutive waves to the four SIMDs in turn, one SIMD
        // holds four of them: those run (P), one wave on each of the other three SIMDs runs the chain of one 16-column
        // tile (G), the rest leave at once.  The fp32 matrix-core instructions and the fp32 / DPP vector instructions of
        // (P) execute on the same lanes of a SIMD and do not overlap (measured: a column step takes 3.4 times as long
        // while a chain runs beside it, and the chain twice as long); apart, each role runs at the speed it has alone.
        // Where a wave really sits is read from HW_ID, so the result never depends on the dealing order -- an unexpected
        // order only costs the separation.
        __shared__ int s_simd[16];
        uint32_t hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        if (lane == 0) s_simd[wv] = (int)((hwid >> 4) & 3u);
        __syncthreads();
        const int nw = (int)(blockDim.x >> 6);
        const int ps = s_simd[0];
        int np = 0, my_p = -1, ng = 0, my_g = -1;
        unsigned seen = 1u << ps;
        for (int w = 0; w < nw; ++w) {
            const int sd = s_simd[w];
            if (sd == ps) {
                if (w == wv) my_p = np;
                ++np;
            } else if (!((seen >> sd) & 1u)) {
                seen |= 1u << sd;
                if (w == wv) my_g = ng;
                ++ng;
            }
        }
        if (np < 4 || ng < SKR) {  // not the expected dealing: roles by wave index (uniform over the workgroup)
            my_p = wv < 4 ? wv : -1;
            my_g = (wv >= 4 && wv < 4 + SKR) ? wv - 4 : -1;
        }
        if (my_p >= 4) my_p = -1;
        if (my_p < 0 && my_g < 0) return;  // the waves nobody needs: gone before the first barrier of the loops
        my_p = __builtin_amdgcn_readfirstlane(my_p);  // wave-uniform by construction: keep them in scalar registers
        my_g = __builtin_amdgcn_readfirstlane(my_g);
        roleG = my_g >= 0;
        gw = roleG ? my_g : my_p;
    } else {
        roleG = wv >= 4;
        gw = wv & 3;
    }
    const int gtid = gw * 64 + lane;  // thread index inside the role
    if (SOLVE_PF && wv == 8) {
        if (helper) return;  // (a helper workgroup is four chain waves)
        // The ninth wave.  (a) It carries every solved panel to memory (late round 3): the packed Err block from ErrPk to the tile's
        // scratch -- where the chain waves or the helper workgroups read it -- and the indices from Qst to Q, 16 bytes per lane,
        // between the two barriers of a step.  The column-step waves used to store both themselves, and since loads and stores
        // share one counter their wait for the next panel's weights also waited for the stores just issued: 0.8 us of every panel
        // step, on the critical path of the launch.  Now those waves issue no stores at all (stage API with an Err output aside),
        // and this wave, which nobody waits for, sees its own stores acknowledged at the top of the next step and then tells
        // the helper workgroups.
        // (b) Prefetch, where asked for (bit 1 of opt_fast).  Every workgroup walks the packed L in the same order at about the
        // same pace, and every block is read exactly once per workgroup: the first of the 32 workgroups behind one L2 to ask for
        // a block waits for HBM (the 67 MB do not stay in the Infinity Cache between launches), and the others, in step with it,
        // wait along -- the chain ran at the latency of that miss, not at the rate of the matrix cores.  This wave touches, one
        // step ahead, the lines the chain waves of its XCD will read in the next step (a tile's blocks are contiguous in the
        // tile-major layout; the workgroups of an XCD share the lines out among themselves).
        const bool pf = (opt_fast & 2) != 0;
        const int nb_ = (n + SB - 1) / SB;
        const int nwg = (min(nact, m) + SR - 1) / SR;   // tiles that run
        const int xi = tile >> 3, nx = (nwg + 7) >> 3;  // this tile among those of its XCD (blockIdx % 8)
        float* __restrict__ errt_ = ErrT + (int64_t)tile * nb_ * SBLK;
        const __amdgpu_buffer_rsrc_t rsT = __builtin_amdgcn_make_buffer_rsrc(errt_, 0, 0xffffffff, 0x00020000);
        // indices: lane = (row of the tile, 16-column piece of the panel)
        const int qrow = lane >> 2, qpiece = lane & 3;
        const int qslot = tile * SR + qrow;
        const bool qrow_ok = qslot < nact;
        const int qprow = rowlist ? rowlist[min(qslot, nact - 1)] : min(qslot, m - 1);
        uint8_t* __restrict__ qdst = Q + (int64_t)qprow * n + 16 * qpiece;
        const bool q16 = (n & 15) == 0 && (reinterpret_cast<uintptr_t>(Q) & 15) == 0;  // 16-byte stores possible
        float sink = 0.0f;
        for (int s_ = 0; s_ <= nb_; ++s_) {
            if (s_ >= 2) {
                // the stores of the step before are acknowledged before this step's first barrier: the chain waves read that block
                // from the scratch after it at the earliest, and the helpers may hear of it now
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (duo && lane == 0)
                    __hip_atomic_store(&ctrl[8 + DUO_TW * tile], tag | (uint32_t)(s_ - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            const int bn = nb_ - 2 - s_;  // the panel whose chain runs in the NEXT step
            if (pf && bn >= 0 && bn + 1 <= nb_ - 1) {
                const int64_t lines = (int64_t)(nb_ - 1 - bn) * (SBLKB / 128);  // 128-byte lines per tile
                const int64_t per = (lines + nx - 1) / nx, l0 = (int64_t)xi * per, l1 = min(lines, l0 + per);
#pragma unroll
                for (int t = 0; t < SKR; ++t) {
                    const float* base = Lr + lr_block(SKR * bn + t, bn + 1, NT, nb_) * SBLK;
                    for (int64_t l = l0 + lane; l < l1; l += 64) sink += base[l * 32];
                }
            }
            __syncthreads();  // panel bP solved: its packed Err block (ErrPk) and its indices (Qst) are in LDS
            const int bP = nb_ - s_;
            if (bP <= nb_ - 1) {
                const int j0 = bP * SB, wd = min(SB, n - j0);
                if (duo || bP < pbase) {  // (blocks >= pbase live in LDS; only helper workgroups want them from memory)
#pragma unroll
                    for (int q4 = 0; q4 < SKR; ++q4) {
                        const u32x4raw v = __builtin_bit_cast(u32x4raw, *reinterpret_cast<const f32x4v*>(ErrPk + (q4 * 64 + lane) * 4));
                        if (duo) __builtin_amdgcn_raw_buffer_store_b128(v, rsT, (q4 * 64 + lane) * 16, bP * SBLKB, 16);  // sc1: for another CU
                        else __builtin_amdgcn_raw_buffer_store_b128(v, rsT, (q4 * 64 + lane) * 16, bP * SBLKB, 0);
                    }
                }
                if (qrow_ok && 16 * qpiece < wd) {
                    uint4 qv = *reinterpret_cast<const uint4*>(&Qst[0][0][0] + qrow * SB + 16 * qpiece);
                    if (q16 && 16 * qpiece + 16 <= wd) {
                        *reinterpret_cast<uint4*>(qdst + j0) = qv;
                    } else {
                        const uint32_t w4[4] = {qv.x, qv.y, qv.z, qv.w};
#pragma unroll
                        for (int e = 0; e < 16; ++e)
                            if (16 * qpiece + e < wd) qdst[j0 + e] = (uint8_t)((w4[e >> 2] >> (8 * (e & 3))) & 0xffu);
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // LDS reads done (ErrPk / Qst are free again); stores in flight
        }
        if (sink == 1.2345e-30f) ErrT[0] = sink;  // never true: keeps the loads
        return;
    }
    if (helper && (!roleG || duo_pol < 0)) return;  // a helper workgroup is four chain waves (pol bit 31, tests: helpers that never answer)
    const int rsub = lane >> 4;
    const int c16 = lane & 15;
    const int prow_in_tile = 4 * gw + rsub;  // row handled by this 16-lane group in phase (P)
    const int slot = tile * SR + prow_in_tile;
    const bool prow_ok = slot < nact;
    const int prow = rowlist ? rowlist[min(slot, nact - 1)] : min(slot, m - 1);
    const int nb = (n + SB - 1) / SB;
    float* __restrict__ errt = ErrT + (int64_t)tile * nb * SBLK;

    PanelState st;
    st.c16 = (uint32_t)c16;
    st.tv = (c16 < V) ? T[(int64_t)prow * V + c16] : __builtin_inff();
    st.pad = (c16 < V) ? 0u : 0xffffffffu;
    float wnext[SKR] = {};
    // (P) threshold form of the argmin: sorted codebook, exact switch-over points (see FastRow); ErrPk is free until the
    // first panel has been solved and serves as the sort's scratch (32 dwords per 16-lane row)
    FastRow fr = {};
    bool fast_ok = false;
    if (!roleG) {
        fr = fast_row_setup(st.tv, (uint32_t)c16,
                            reinterpret_cast<volatile lds_u32_t*>((lds_u32_t*)(uintptr_t)(uint32_t)(uintptr_t)ErrPk) + 32 * (4 * gw + rsub));
        fast_ok = __ballot(fr.xb > 0.0f) == ~0ull && (opt_fast & 1);  // all four rows of the wave
    }
    const uint32_t qaddr = (uint32_t)(uintptr_t)&Qst[gw][rsub][0], qdummy = (uint32_t)(uintptr_t)&Qdm[gw][lane];
    lds_u8_t* qstage = (lds_u8_t*)(uintptr_t)qaddr;

    // (P) where this lane's four Err values go inside a packed block: column col = c16 + 16 k is k-group col >> 2,
    // slot col & 3, i.e. MFMA lane (ksub, row)
    int pk_idx[SKR];
#pragma unroll
    for (int k = 0; k < SKR; ++k) {
        const int col = c16 + 16 * k;
        const int ks = col & 3, ksub_w = KASC ? (3 - ks) : ks;
        const int g = col >> 2;
        pk_idx[k] = (g >> 2) * 256 + (ksub_w * 16 + prow_in_tile) * 4 + (g & 3);
    }

    if constexpr (SPLIT) {
        if (roleG) {
            __syncthreads();  // the set-up barrier of the P side
        } else {
            // ---- the SIMD without chains: wave gw = 0 runs the column steps of all 16 rows (4 lanes per row); the other three
            // waves, after helping with the set-up, stage what the next panel needs -----------------------------------------
            Fs[0][4 * gw + rsub][c16] = fr.tv;
            Fs[1][4 * gw + rsub][c16] = fr.lo;
            Fs[2][4 * gw + rsub][c16] = fr.hi;
            Fo[4 * gw + rsub][c16] = fr.orig;
            if (c16 == 0) Fok[4 * gw + rsub] = fr.xb > 0.0f ? 1 : 0;
            __syncthreads();
            if (gw != 0) {
                const int htid = (gw - 1) * 64 + lane;  // 0 .. 191
                constexpr int HT = 192, NST = (SB * SB) / HT, NB2 = (SKR * SBLK) / HT;
                static_assert((SB * SB) % HT == 0 && (SKR * SBLK) % HT == 0, "staging shares");
                // ... and they pull the blocks of L that the chains of this XCD read in the NEXT step into L2, one line per
                // thread and tile: every block is read once per workgroup, and the 32 workgroups behind an L2 would otherwise
                // miss together (see the prefetch wave of the 64-column layout)
                const int nwg = (min(nact, m) + SR - 1) / SR;
                const int xi = (int)blockIdx.x >> 3, nx = (nwg + 7) >> 3;
                float sink = 0.0f;
                for (int s = 0; s <= nb; ++s) {
                    const int bP = nb - s, bN = bP - 1;  // bN: the panel whose operands are staged in this step
                    {
                        const int bn = nb - 2 - s;  // the panel whose chain runs in the next step
                        if (bn >= 0 && bn + 1 <= nb - 1) {
                            const int64_t lines = (int64_t)(nb - 1 - bn) * (SBLKB / 128);
                            const int64_t per = (lines + nx - 1) / nx, l0 = (int64_t)xi * per, l1 = min(lines, l0 + per);
#pragma unroll
                            for (int t = 0; t < SKR; ++t) {
                                const float* base = Lr + lr_block(SKR * bn + t, bn + 1, NT, nb) * SBLK;
                                for (int64_t l = l0 + htid; l < l1; l += HT) sink += base[l * 32];
                            }
                        }
                    }
                    if (bN >= 0) {
                        float lst[NST], bst[NB2], dst_d = 1.0f;
                        const int j0n = bN * SB, wdn = min(SB, n - j0n);
#pragma unroll
                        for (int e = 0; e < NST; ++e) {
                            const int idx = e * HT + htid;
                            const int jj = idx / SB, col = idx % SB;
                            lst[e] = (jj < wdn && col < wdn) ? L[(int64_t)(j0n + jj) * ldl + j0n + col] : 0.0f;
                        }
                        if (htid < SB && htid < wdn) dst_d = L[(int64_t)(j0n + htid) * ldl + j0n + htid];
                        const bool part2 = bP <= nb - 1;
                        if (part2) {
#pragma unroll
                            for (int e = 0; e < NB2; ++e) {
                                const int idx = e * HT + htid;
                                bst[e] = Lr[lr_block(SKR * bN + idx / SBLK, bP, NT, nb) * SBLK + idx % SBLK];
                            }
                        }
#pragma unroll
                        for (int e = 0; e < NST; ++e) {
                            const int idx = e * HT + htid;
                            const int jj = idx / SB, col = idx % SB;
                            Ld4[(bN & 1) * SB * SB + (jj * 4 + (col & 3)) * P4K + (col >> 2)] = lst[e];
                        }
                        if (htid < SB) Dg[bN & 1][htid] = make_float2(dst_d, 1.0f / dst_d);
                        if (part2) {
#pragma unroll
                            for (int e = 0; e < NB2; ++e) Bp2[e * HT + htid] = bst[e];
                        }
                        // the 16 rows' weights of panel bN (zero beyond n)
#pragma unroll
                        for (int e = 0; e < (16 * SB) / HT; ++e) {
                            const int idx = e * HT + htid;
                            const int rr = idx / SB, col = idx % SB;
                            const int sl = tile * SR + rr;
                            const int pr = rowlist ? rowlist[min(sl, nact - 1)] : min(sl, m - 1);
                            Wp[bN & 1][rr][col] = (j0n + col < n) ? W[(int64_t)pr * n + j0n + col] : 0.0f;
                        }
                    }
                    __syncthreads();  // panel bP solved: its packed Err block (ErrPk) and its indices (Qst) are in LDS
                    // The helpers carry the panel's results to memory, so that the P wave never waits for a store: Err block ->
                    // global scratch (panels the chains read from memory), indices -> Q, errors -> ErrOut.  The stores are only
                    // waited for at the NEXT step's first barrier (a full __syncthreads): the scratch block is first read two
                    // steps later.
                    if (bP <= nb - 1) {
                        const int j0 = bP * SB, wd = min(SB, n - j0);
                        if (bP < pbase) {
                            const float4 v = reinterpret_cast<const float4*>(ErrPk)[htid];  // SBLK = 4 * 192 floats
                            reinterpret_cast<float4*>(errt + bP * SBLK)[htid] = v;
                        }
                        const int rr = htid / (SB / 4), g4 = htid % (SB / 4);  // row, group of four columns
                        const int sl = tile * SR + rr;
                        if (sl < nact) {
                            const int pr = rowlist ? rowlist[sl] : sl;
                            const uint32_t qq = *reinterpret_cast<const uint32_t*>(&Qst[0][0][0] + rr * SB + 4 * g4);
#pragma unroll
                            for (int c = 0; c < 4; ++c) {
                                const int col = 4 * g4 + c;
                                if (col < wd) {
                                    Q[(int64_t)pr * n + j0 + col] = (uint8_t)min((qq >> (8 * c)) & 0xffu, (uint32_t)(V - 1));
                                    if (ErrOut) {
                                        const int pk = (g4 >> 2) * 256 + ((KASC ? 3 - c : c) * 16 + rr) * 4 + (g4 & 3);
                                        ErrOut[(int64_t)pr * n + j0 + col] = ErrPk[pk];
                                    }
                                }
                            }
                        }
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // second barrier: LDS reads done, stores in flight
                }
                if (sink == 1.2345e-30f) ErrT[0] = sink;  // never true: keeps the prefetch loads
                return;
            }
            // ---- the P wave ------------------------------------------------------------------------------------------------
            const int row4 = lane >> 2;
            const uint32_t sub = (uint32_t)(lane & 3);
            const int slot4 = tile * SR + row4;
            const int prow4 = rowlist ? rowlist[min(slot4, nact - 1)] : min(slot4, m - 1);
            P4Fast pf;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int e = 4 * (int)sub + i;
                pf.tv[i] = Fs[0][row4][e];
                pf.lo[i] = Fs[1][row4][e];
                pf.hi[i] = Fs[2][row4][e];
                pf.orig[i] = Fo[row4][e] + 1u;
            }
            const bool fast4 = __ballot(Fok[row4] != 0) == ~0ull && (opt_fast & 1);
            const uint32_t qaddr4 = (uint32_t)(uintptr_t)&Qst[0][0][0] + (uint32_t)(row4 * SB);
            const uint32_t qdummy4 = (uint32_t)(uintptr_t)&Qdm4[lane][0];
            // column sub + 4k is k-group k, slot sub of a packed block: index pk4b + (k >> 2) * 256 + (k & 3)
            const int pk4b = ((KASC ? (3 - (int)sub) : (int)sub) * 16 + row4) * 4;
            P4State ps;
            for (int s = 0; s <= nb; ++s) {
                const int bP = nb - s;
                GANQ_TRACE(0, s, 0);
                if (bP <= nb - 1) {
                    const int j0 = bP * SB;
                    const int wd = min(SB, n - j0);
#pragma unroll
                    for (int k = 0; k < P4K; ++k) {
                        ps.tc[k] = 0.0f;
                        ps.r[k] = Rp[bP & 1][row4][(int)sub + 4 * k];
                    }
                    const float* wrow = &Wp[bP & 1][row4][(int)sub];  // this lane's columns: wrow[4k]
                    const float wk0 = wrow[4 * ((SB - 1) >> 2)];
                    const float* ld4 = Ld4 + (bP & 1) * SB * SB;
                    const float2* Dgp = Dg[bP & 1];
                    float l0[P4K];
                    p4_load_l<SB - 1>(ld4, sub, l0);
                    bool bad = !fast4;
                    if (fast4) {
                        uint32_t nsel = 0;
                        if (wd == SB) p4_from_fast<true, SB - 1>(ps, pf, ld4, Dgp, wrow, sub, wd, Dgp[SB - 1], wk0, l0, qaddr4, qdummy4, nsel);
                        else p4_from_fast<false, SB - 1>(ps, pf, ld4, Dgp, wrow, sub, wd, Dgp[SB - 1], wk0, l0, qaddr4, qdummy4, nsel);
                        nsel += dpp_u<0xB1>(nsel);
                        nsel += dpp_u<0x4E>(nsel);
                        bad = __ballot(nsel != (uint32_t)wd) != 0ull;
                    }
#ifdef GANQ_SOLVE_DEBUG
                    if (lane == 0) atomicAdd(&g_solve_dbg[fast4 ? (bad ? 1 : 0) : 2], 1ull);
#endif
                    if (bad) {  // wave-uniform: redo the panel by the reduction form
#pragma unroll
                        for (int k = 0; k < P4K; ++k) {
                            ps.tc[k] = 0.0f;
                            ps.r[k] = Rp[bP & 1][row4][(int)sub + 4 * k];
                        }
                        float To[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int e = 4 * (int)sub + i;
                            To[i] = (e < V) ? T[(int64_t)prow4 * V + e] : __builtin_inff();
                        }
                        if (wd == SB) p4_from_slow<true, SB - 1>(ps, To, ld4, Dgp, wrow, sub, wd, Dgp[SB - 1], wk0, l0, qaddr4, qdummy4);
                        else p4_from_slow<false, SB - 1>(ps, To, ld4, Dgp, wrow, sub, wd, Dgp[SB - 1], wk0, l0, qaddr4, qdummy4);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");  // the indices filed in LDS by the steps
                    __builtin_amdgcn_wave_barrier();
                    GANQ_TRACE(0, s, 1);
#pragma unroll
                    for (int k = 0; k < P4K; ++k) {
                        const int col = (int)sub + 4 * k;
                        const float ev = (col < wd) ? wrow[4 * k] - ps.tc[k] : 0.0f;  // err = w - T[idx]; zero beyond n in the top panel
                        const int pk = pk4b + (k >> 2) * 256 + (k & 3);
                        ErrPk[pk] = ev;  // (the helper waves take it, and the indices in Qst, to memory after the barrier)
                        if (bP >= pbase) ErrL[(bP - pbase) * SBLK + pk] = ev;
                    }
                }
                GANQ_TRACE(0, s, 2);
                __syncthreads();
                GANQ_TRACE(0, s, 3);
                __syncthreads();
                GANQ_TRACE(0, s, 4);
            }
            return;
        }
    }

    // The two roles run SEPARATE loops with the same number of barriers (s_barrier counts arrivals, not places): the
    // registers of one role are then not live in the other's loop, and the kernel needs max(P, G) of them instead of
    // the sum.
    if (!roleG) {
        for (int s = 0; s <= nb; ++s) {
            const int bP = nb - s;  // panel solved in this step (none at s = 0)
            GANQ_TRACE(0, s, 0);
                // ---- (P) ---------------------------------------------------------------------------------------
                if (bP <= nb - 1) {
                    const int j0 = bP * SB;
                    const int wd = min(SB, n - j0);
#pragma unroll
                    for (int k = 0; k < SKR; ++k) {
                        st.w[k] = wnext[k];
                        st.q[k] = 0;
                        st.e[k] = 0.0f;
                        st.r[k] = Rp[bP & 1][prow_in_tile][c16 + 16 * k];
                    }
                    if (bP >= 1) {
#pragma unroll
                        for (int k = 0; k < SKR; ++k) wnext[k] = W[(int64_t)prow * n + j0 - SB + c16 + 16 * k];  // full panel
                    }
#ifndef GANQ_SOLVE_NO_P  // timing experiment: results are meaningless without the panel steps
                    bool bad = !fast_ok;
                    if (fast_ok) {
                        const float4(*Ldp)[16] = Ld[bP & 1];
                        const float2* Dgp = Dg[bP & 1];
                        uint32_t nsel = 0;
#ifdef GANQ_SOLVE_DEBUG
                        const unsigned long long t_in = __builtin_amdgcn_s_memtime();
#endif
                        if (wd == SB) panel_from_fast<true, SB - 1>(st, fr, Ldp, Dgp, wd, Dgp[SB - 1], Ldp[SB - 1][st.c16], qaddr, qdummy, nsel);
                        else panel_from_fast<false, SB - 1>(st, fr, Ldp, Dgp, wd, Dgp[SB - 1], Ldp[SB - 1][st.c16], qaddr, qdummy, nsel);
#ifdef GANQ_SOLVE_DEBUG
                        if (blockIdx.x == 0 && tid == 0) atomicAdd(&g_solve_dbg[3], __builtin_amdgcn_s_memtime() - t_in);
#endif
                        // every step of every row selected exactly one lane <=> each row's selections add up to the step count
                        nsel += dpp_u<0xB1>(nsel);
                        nsel += dpp_u<0x4E>(nsel);
                        nsel += dpp_u<0x141>(nsel);
                        nsel += dpp_u<0x140>(nsel);
                        bad = __ballot(nsel != (uint32_t)wd) != 0ull;
                        if (!bad && (!SOLVE_PF || ErrOut)) {  // (the indices stay in Qst for the ninth wave otherwise)
                            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                            __builtin_amdgcn_wave_barrier();
#pragma unroll
                            for (int k = 0; k < SKR; ++k) st.q[k] = qstage[c16 + 16 * k];
                        }
                    }
#ifdef GANQ_SOLVE_DEBUG
                if (lane == 0) atomicAdd(&g_solve_dbg[fast_ok ? (bad ? 1 : 0) : 2], 1ull);
#endif
                if (bad) {  // wave-uniform: a step left the range the thresholds are exact in (or the row never had one)
#pragma unroll
                        for (int k = 0; k < SKR; ++k) {
                            st.q[k] = 0;
                            st.e[k] = 0.0f;
                            st.r[k] = Rp[bP & 1][prow_in_tile][c16 + 16 * k];
                        }
                        if (wd == SB) {
                            panel_all<true>(st, Ld[bP & 1], Dg[bP & 1], wd, std::make_integer_sequence<int, SB>{});
                        } else {
                            panel_all<false>(st, Ld[bP & 1], Dg[bP & 1], wd, std::make_integer_sequence<int, SB>{});
                        }
                    }
#endif
                    GANQ_TRACE(0, s, 1);
#pragma unroll
                    for (int k = 0; k < SKR; ++k) {
                        const int col = c16 + 16 * k;
                        const float ev = (col < wd) ? st.e[k] : 0.0f;  // the (partial) top panel is zero beyond n
                        ErrPk[pk_idx[k]] = ev;
                        // the block lives in LDS (panels >= pbase, never read back from memory) or in the global scratch; with a
                        // helper workgroup every block also goes to the scratch, visible to the other CU
                        if (bP >= pbase) ErrL[(bP - pbase) * SBLK + pk_idx[k]] = ev;
                        if constexpr (SOLVE_PF) {
                            // the ninth wave takes the block and the indices to memory; a panel redone by the reductions has its
                            // indices in registers: filed where the fast path files them
#ifndef GANQ_SOLVE_NO_P
                            if (bad) qstage[c16 + 16 * k] = (uint8_t)min(st.q[k], (uint32_t)(V - 1));
#endif
                            if (ErrOut && col < wd && prow_ok) ErrOut[(int64_t)prow * n + j0 + col] = st.e[k];
                        } else {
                            if (duo) __hip_atomic_store(&errt[bP * SBLK + pk_idx[k]], ev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            else if (bP < pbase) errt[bP * SBLK + pk_idx[k]] = ev;
                            if (col < wd && prow_ok) {
                                Q[(int64_t)prow * n + j0 + col] = (uint8_t)min(st.q[k], (uint32_t)(V - 1));
                                if (ErrOut) ErrOut[(int64_t)prow * n + j0 + col] = st.e[k];
                            }
                        }
                    }
                } else {
                    const int j0 = (nb - 1) * SB;  // W of the first (possibly partial) panel
#pragma unroll
                    for (int k = 0; k < SKR; ++k) {
                        const int col = c16 + 16 * k;
                        wnext[k] = (j0 + col < n) ? W[(int64_t)prow * n + j0 + col] : 0.0f;
                    }
                }
            GANQ_TRACE(0, s, 2);
            if (!SOLVE_PF && duo) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's Err stores are acknowledged
            __syncthreads();  // panel bP solved (ErrPk, Qst in LDS); part 1 of panel bG done
            GANQ_TRACE(0, s, 3);
            // (without the ninth wave: the barrier waited for every wave's stores, the panel's block is in memory before the helper
            // hears of it)
            if (!SOLVE_PF && duo && tid == 0 && s >= 1)
                __hip_atomic_store(&ctrl[8 + DUO_TW * tile], tag | (uint32_t)s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();  // R of panel bG, its Ld / Dg ready; ErrPk free
            GANQ_TRACE(0, s, 4);
        }
        return;
    }
#ifndef GANQ_SPLIT_RING
#define GANQ_SPLIT_RING 4
#endif
    constexpr int RING = SPLIT ? GANQ_SPLIT_RING : GANQ_SOLVE_RING;
    // the operand ring lives across the steps: the first RING - 1 batches of the NEXT step's chain are requested before this
    // step's barriers (the blocks they need exist since long), so that a chain starts with its operands at hand instead of a
    // memory round trip
    constexpr bool PRIME = GANQ_SOLVE_PRIME != 0 && GANQ_MFMA_INPLACE != 0;
    f32x4v ra[RING][SKR], rb[RING][SKR];
    bool primed = false;
    f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
    // shared by both roles (the P waves run a chain too when they assist, see below)
    const __amdgpu_buffer_rsrc_t rsrcL = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Lr), 0, 0xffffffff, 0x00020000);
    // (a helper workgroup reads its own copy of the tile's Err blocks, see below)
    float* __restrict__ errg = helper ? ErrH + (hstage - 1) * errh_stride + (int64_t)tile * nb * SBLK : errt;
    const __amdgpu_buffer_rsrc_t rsrcE = __builtin_amdgcn_make_buffer_rsrc(errg, 0, 0xffffffff, 0x00020000);
    // zero records: every load through it is out of range and returns 0
    const __amdgpu_buffer_rsrc_t rsrcZ = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Lr), 0, 0, 0x00020000);
    const int voff = lane * 16;  // this lane's 16 bytes inside each quarter of a packed block
    // One chain segment: source panels phi, phi-1, .., plo (descending), A from LDS (panels >= pbase) or from
    // the global scratch.  One batch = one source panel = 16 MFMAs; operands are loaded two batches ahead into
    // three rotating register sets; every load is unconditional and the loop runs whole rounds of three
    // (batches past the end read B through rsrcZ: zeros leave the accumulator as it is), so that the waits
    // in the steady state count exactly the loads still allowed in flight.
    // a_in_lds: 1 = every A block of the segment is in LDS, 0 = none, 2 = decided per batch (split layout: ONE segment per
    // step, so the ring is filled once -- each fill is a full memory round trip before the first MFMA)
    auto chain = [&](auto a_in_lds, int phi, int plo, const int ct, bool have_head) {
        constexpr bool ALDS = decltype(a_in_lds)::value == 1;
        constexpr bool MIXED = decltype(a_in_lds)::value == 2;
        const int nbat = phi - plo + 1;
        if (nbat <= 0) return;
        auto& a = ra;
        auto& b = rb;
        auto ld = [&](int bi, f32x4v (&aa)[SKR], f32x4v (&bb)[SKR]) {
            const bool real = bi < nbat;
            const int ps = phi - min(bi, nbat - 1);  // source panel of the batch
            const uint32_t sB = (uint32_t)lr_block(ct, ps, NT, nb) * (uint32_t)SBLKB;
            const __amdgpu_buffer_rsrc_t rsB = real ? rsrcL : rsrcZ;
            const f32x4v* Al = reinterpret_cast<const f32x4v*>(ErrL + (ps - pbase) * SBLK + lane * 4);
#pragma unroll
            for (int j = 0; j < SKR; ++j) {
                bb[j] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(rsB, voff + 1024 * j, (int)sB, 0));
                if (ALDS || (MIXED && ps >= pbase)) aa[j] = Al[j * 64];
                else aa[j] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(rsrcE, voff + 1024 * j, ps * SBLKB, 0));
            }
        };
        auto mm = [&](const f32x4v (&aa)[SKR], const f32x4v (&bb)[SKR]) {
#pragma unroll
            for (int g = SKG - 1; g >= 0; --g) {
#if GANQ_MFMA_INPLACE
                // accumulate IN PLACE: the register allocator otherwise moves the accumulator between registers along
                // the chain, and a dependent MFMA whose destination differs from its SrcC loses the back-to-back path
                mfma_inplace(acc, aa[g >> 2][g & 3], bb[g >> 2][g & 3]);
#else
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(aa[g >> 2][g & 3], bb[g >> 2][g & 3], acc, 0, 0, 0);
#endif
            }
        };
        // GANQ_SOLVE_RING register sets in rotation: the operands of batch k + RING - 1 are requested while batch k is
        // multiplied (B comes from L2 / the Infinity Cache: the deeper the ring, the more of that latency is covered)
        if (!have_head) {  // wave-uniform
#pragma unroll
            for (int u = 0; u < RING - 1; ++u) ld(u, a[u], b[u]);
        }
        GANQ_PIN();
        // one stage = the loads of a later batch spread between the 16 MFMAs of batch k
        auto stage_sched = [&]() {
#pragma unroll
            for (int i = 0; i < SKR; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);             // MFMA
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);             // VMEM read (B)
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                if (ALDS) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // DS read (A)
                else __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);        // VMEM read (A)
            }
        };
#if GANQ_MFMA_INPLACE
        // the MFMAs are inline assembly (see mm), which the scheduler leaves where it is written together with the loads
        // around it: the interleaving is spelled out -- after every 4 MFMAs of batch k, one 16-byte piece of B and of A of
        // batch k + RING - 1
        auto mm_ld = [&](const f32x4v (&aa)[SKR], const f32x4v (&bb)[SKR], int bi, f32x4v (&an)[SKR], f32x4v (&bn)[SKR]) {
            const bool real = bi < nbat;
            const int ps = phi - min(bi, nbat - 1);
            const uint32_t sB = (uint32_t)lr_block(ct, ps, NT, nb) * (uint32_t)SBLKB;
            const __amdgpu_buffer_rsrc_t rsB = real ? rsrcL : rsrcZ;
            const f32x4v* Al = reinterpret_cast<const f32x4v*>(ErrL + (ps - pbase) * SBLK + lane * 4);
            const bool a_lds = ALDS || (MIXED && ps >= pbase);  // wave-uniform
#pragma unroll
            for (int j = SKR - 1; j >= 0; --j) {
#pragma unroll
                for (int g = 4 * j + 3; g >= 4 * j; --g) mfma_inplace(acc, aa[g >> 2][g & 3], bb[g >> 2][g & 3]);
                const int jl = SKR - 1 - j;
                bn[jl] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(rsB, voff + 1024 * jl, (int)sB, 0));
                if (a_lds) an[jl] = Al[jl * 64];
                else an[jl] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(rsrcE, voff + 1024 * jl, ps * SBLKB, 0));
            }
        };
        for (int bi = 0; bi < nbat; bi += RING) {  // whole rounds, no exits inside: a plain counted loop
#pragma unroll
            for (int u = 0; u < RING; ++u) {
                mm_ld(a[u], b[u], bi + u + RING - 1, a[(u + RING - 1) % RING], b[(u + RING - 1) % RING]);
                GANQ_PIN();
            }
        }
        (void)mm;
        (void)stage_sched;
#else
        for (int bi = 0; bi < nbat; bi += RING) {  // whole rounds, no exits inside: a plain counted loop
#pragma unroll
            for (int u = 0; u < RING; ++u) {
                ld(bi + u + RING - 1, a[(u + RING - 1) % RING], b[(u + RING - 1) % RING]);
                mm(a[u], b[u]);
                stage_sched();
                GANQ_PIN();
            }
        }
#endif
    };
    // ---- helper workgroup: the far part of every panel's chain (see "duo" above) ---------------------------------------
    if (helper) {
        if constexpr (!SPLIT) {
            const __amdgpu_buffer_rsrc_t rsrcM = __builtin_amdgcn_make_buffer_rsrc(errt, 0, 0xffffffff, 0x00020000);  // the tile's blocks
            // accumulators: out to the next stage (the tile, or the first helper), in from the second helper (first helper of a trio)
            const __amdgpu_buffer_rsrc_t rsrcF = __builtin_amdgcn_make_buffer_rsrc(
                Facc + (hstage - 1) * facc_stride + (int64_t)tile * (nb + 1) * SBLK, 0, 0xffffffff, 0x00020000);
            const __amdgpu_buffer_rsrc_t rsrcU =
                __builtin_amdgcn_make_buffer_rsrc(Facc + facc_stride + (int64_t)tile * (nb + 1) * SBLK, 0, 0xffffffff, 0x00020000);
            const uint32_t* solved = ctrl + 8 + DUO_TW * tile;
            uint32_t* ready = ctrl + 8 + DUO_TW * tile + 1 + 4 * (hstage - 1) + gw;
            const uint32_t* upstream = ctrl + 8 + DUO_TW * tile + 5 + gw;
            int have = nb;  // lowest panel copied so far
            for (int s = 1; s <= nb - 1; ++s) {
                const int c = s - 1;
                const int h = c - duo_near(c, pol);  // source panels nb-1 .. nb-h are the helpers'
                if (h <= 0) continue;
                const int h2 = nhelp == 2 ? h / 2 : 0;                    // ... nb-1 .. nb-h2 the second helper's
                const int phi = hstage == 2 || h2 == 0 ? nb - 1 : nb - 1 - h2;  // this workgroup's panels: phi .. need
                const int need = hstage == 2 ? nb - h2 : nb - h;
                if (phi < need) continue;  // (second helper of a trio at a step whose share is a single panel)
                if (need < have) {  // (uniform over the four waves)
                    // panel p was solved in step nb - p: wait until the tile has announced that step
                    int seen = 0;
                    if (lane == 0) {
                        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                        for (;;) {
                            const uint32_t v = __hip_atomic_load(solved, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if ((v ^ tag) < 1024u && (v & 1023u) >= (uint32_t)(nb - need)) {
                                seen = 1;
                                break;
                            }
                            // every spin is bounded: a tile that has not announced a panel for 16 x DUO_TIMEOUT is not coming
                            // (it gave up on this helper long before); the four waves read the same flag and leave together
                            if (__builtin_amdgcn_s_memrealtime() - t0 > 16 * DUO_TIMEOUT) break;
                            __builtin_amdgcn_s_sleep(8);
                        }
                    }
                    if (!__builtin_amdgcn_readfirstlane(seen)) return;
                    // copy the new blocks: 4 KB each, 16 bytes per thread; into LDS (panels >= pbase) or this workgroup's own
                    // scratch (plain stores and loads from here on: one CU, one L1)
                    for (int pp = have - 1; pp >= need; --pp) {
                        const f32x4v v = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(rsrcM, gtid * 16, pp * SBLKB, 16));
                        if (pp >= pbase) *reinterpret_cast<f32x4v*>(ErrL + (pp - pbase) * SBLK + gtid * 4) = v;
                        else *reinterpret_cast<f32x4v*>(errg + pp * SBLK + gtid * 4) = v;
                    }
                    have = need;
                    __syncthreads();
                }
                acc = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                if (hstage == 1 && h2 > 0) {  // first helper of a trio: continue from the second helper's accumulators
                    int ok = 0;
                    if (lane == 0) {
                        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                        for (;;) {
                            const uint32_t v = __hip_atomic_load(upstream, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if ((v ^ tag) < 1024u && (v & 1023u) >= (uint32_t)s) {
                                ok = 1;
                                break;
                            }
                            if (__builtin_amdgcn_s_memrealtime() - t0 > 16 * DUO_TIMEOUT) break;
                            __builtin_amdgcn_s_sleep(4);
                        }
                    }
                    if (!__builtin_amdgcn_readfirstlane(ok)) return;  // (the tile gives up on this wave's accumulators and works alone)
                    acc = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrcU, voff + 1024 * gw, s * SBLKB, 16));
                }
                const int ct = SKR * (nb - 1 - s) + gw;  // this wave's 16-wide tile of panel bG = nb - 1 - s
                const int plds = max(need, pbase);
                chain(std::integral_constant<int, 1>{}, phi, plds, ct, false);
                chain(std::integral_constant<int, 0>{}, min(phi, plds - 1), need, ct, false);
#if GANQ_MFMA_INPLACE
                asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#endif
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4raw, acc), rsrcF, voff + 1024 * gw, s * SBLKB, 16);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // acknowledged = visible to every CU
                if (lane == 0) __hip_atomic_store(ready, tag | (uint32_t)s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        return;
    }
    bool duo_live = duo;  // (per chain wave) the helper's accumulators are still being waited for
    for (int s = 0; s <= nb; ++s) {
        const int bG = nb - 1 - s;  // panel whose residual is produced in this step (none at s = nb)
        GANQ_TRACE(1, s, 0);
        acc = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        f32x4v bpre[SKR];  // (G) packed L block (source panel bG+1, this wave's tile): the B operands of part 2
        if (bG >= 0) {
            // ---- (G) part 1: the panels right of panel bG+1, descending --------------------------------------
            const int j0 = bG * SB;
            const int wd = min(SB, n - j0);
            const int ct = SKR * bG + gw;  // this wave's 16-wide tile of panel bG
            // prefetches for later in this step: the panel's own block of L, its diagonal, the B operands of part 2
            // (split layout: the P waves stage the panel's triangle, its diagonal and the part-2 operands instead -- the
            // chain waves need their registers for a deeper operand ring, and P has both time and registers to spare)
            constexpr int GT = 64 * SKR;  // threads of the role
            constexpr int NLP = SPLIT ? 1 : (SB * SB) / GT;
            float lpre[NLP];
            float dpre = 1.0f;
            if constexpr (!SPLIT) {
#pragma unroll
                for (int e = 0; e < NLP; ++e) {
                    const int idx = e * GT + gtid;
                    const int jj = idx / SB, col = idx % SB;
                    lpre[e] = (jj < wd && col < wd) ? L[(int64_t)(j0 + jj) * ldl + j0 + col] : 0.0f;
                }
                if (gtid < SB && gtid < wd) dpre = L[(int64_t)(j0 + gtid) * ldl + j0 + gtid];
            }
            if (!SPLIT && bG + 1 <= nb - 1) {
                const uint32_t sb = (uint32_t)lr_block(ct, bG + 1, NT, nb) * (uint32_t)SBLKB;
#pragma unroll
                for (int j = 0; j < SKR; ++j)
                    bpre[j] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(rsrcL, voff + 1024 * j, (int)sb, 0));
            }
            const int plds = max(bG + 2, pbase);  // lowest source panel of this chain whose Err block lives in LDS
#ifndef GANQ_SOLVE_NO_G  // timing experiment: results are meaningless without the residual chain
            // with a helper workgroup: the first h source panels of the chain are its work, the accumulators come from memory
            int top = nb - 1;
            if constexpr (!SPLIT) {
                if (duo_live) {
                    const int c = s - 1, h = c - duo_near(c, pol);
                    if (h > 0) {
                        int ok = 0;
                        if (lane == 0) {
                            const uint32_t* ready = ctrl + 8 + DUO_TW * tile + 1 + gw;
                            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                            for (;;) {
                                const uint32_t v = __hip_atomic_load(ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                if ((v ^ tag) < 1024u && (v & 1023u) >= (uint32_t)s) {
                                    ok = 1;
                                    break;
                                }
                                if (__builtin_amdgcn_s_memrealtime() - t0 > DUO_TIMEOUT) break;
                                __builtin_amdgcn_s_sleep(4);
                            }
                        }
                        ok = __builtin_amdgcn_readfirstlane(ok);
                        if (ok) {
                            const __amdgpu_buffer_rsrc_t rsrcF =
                                __builtin_amdgcn_make_buffer_rsrc(Facc + (int64_t)tile * (nb + 1) * SBLK, 0, 0xffffffff, 0x00020000);
                            acc = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrcF, voff + 1024 * gw, s * SBLKB, 16));
                            top = nb - 1 - h;
                        } else {
                            duo_live = false;  // no helper in sight: this wave computes its chains alone from here on
                        }
                    }
                }
            }
            if constexpr (false) {  // ONE mixed segment per step (A source decided per batch): measured slower, 23.5 vs 17.9 us for
                (void)plds;         // the last chains -- the per-batch branch costs the ring its load / MFMA interleave
                chain(std::integral_constant<int, 2>{}, nb - 1, bG + 2, ct, false);
            } else {
                chain(std::integral_constant<int, 1>{}, top, plds, ct, PRIME && primed);
                chain(std::integral_constant<int, 0>{}, min(top, plds - 1), bG + 2, ct, false);
            }
#if GANQ_MFMA_INPLACE
            // (inline assembly is invisible to the hazard recogniser: let the last MFMA retire before acc is read again)
            asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#endif
#endif
            GANQ_TRACE(1, s, 1);
            if constexpr (!SPLIT) {
#pragma unroll
                for (int e = 0; e < NLP; ++e) {
                    const int idx = e * GT + gtid;
                    const int jj = idx / SB, col = idx % SB;
                    reinterpret_cast<float*>(&Ld[bG & 1][jj][col & 15])[col >> 4] = lpre[e];
                }
                if (gtid < SB) Dg[bG & 1][gtid] = make_float2(dpre, 1.0f / dpre);
            }
        }
        GANQ_TRACE(1, s, 2);
        // (the chain waves exchange data through LDS only and store nothing to memory: their barriers wait for LDS, not for
        // the primed loads in flight)
        if constexpr (PRIME) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else __syncthreads();  // panel bP solved (ErrPk, ErrT visible); part 1 of panel bG done
        GANQ_TRACE(1, s, 3);
        // (after the barrier: the head may include the panel the P waves have just finished)
        primed = false;
        if constexpr (PRIME) {
            // head of the next step's first segment (panel bG - 1, A blocks in LDS): batches 0 .. RING - 2
            const int bGn = bG - 1, pldsn = max(bGn + 2, pbase);
            const int nbatn = nb - 1 - pldsn + 1;
            if (bGn >= 0 && nbatn > 0) {
                const int ctn = SKR * bGn + gw;
#pragma unroll
                for (int u = 0; u < RING - 1; ++u) {
                    const bool real = u < nbatn;
                    const int ps = nb - 1 - min(u, nbatn - 1);
                    const uint32_t sB = (uint32_t)lr_block(ctn, ps, NT, nb) * (uint32_t)SBLKB;
                    const __amdgpu_buffer_rsrc_t rsB = real ? rsrcL : rsrcZ;
                    const f32x4v* Al = reinterpret_cast<const f32x4v*>(ErrL + (ps - pbase) * SBLK + lane * 4);
#pragma unroll
                    for (int j = 0; j < SKR; ++j) {
                        rb[u][j] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(rsB, voff + 1024 * j, (int)sB, 0));
                        ra[u][j] = Al[j * 64];
                    }
                }
                primed = true;
            }
        }
        if (bG >= 0) {
            // ---- (G) part 2: the 64 columns of panel bG+1, descending; then publish R ------------------------
            if (bG + 1 <= nb - 1) {
                const f32x4v* Ap = reinterpret_cast<const f32x4v*>(ErrPk + lane * 4);
                f32x4v ap[SKR];
#pragma unroll
                for (int j = 0; j < SKR; ++j) ap[j] = Ap[j * 64];
                if constexpr (SPLIT) {
                    const f32x4v* Bp = reinterpret_cast<const f32x4v*>(Bp2 + gw * SBLK + lane * 4);
#pragma unroll
                    for (int j = 0; j < SKR; ++j) bpre[j] = Bp[j * 64];
                }
#pragma unroll
                for (int g = SKG - 1; g >= 0; --g)
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[g >> 2][g & 3], bpre[g >> 2][g & 3], acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) Rp[bG & 1][rsub * 4 + r][16 * gw + c16] = acc[r];
        }
        GANQ_TRACE(1, s, 4);
        if constexpr (PRIME) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else __syncthreads();  // R of panel bG, its Ld / Dg ready; ErrPk free
        GANQ_TRACE(1, s, 5);
    }
}

struct SolveLayout {
    int nb, NT;
    size_t errt_bytes, lr_bytes, facc_bytes, errh_bytes, ctrl_bytes, total;
};
static SolveLayout solve_layout(int64_t m, int64_t n) {
    SolveLayout lo;
    lo.nb = (int)((n + SB - 1) / SB);
    lo.NT = SKR * lo.nb;
    const int64_t tiles = (m + SR - 1) / SR;
    lo.errt_bytes = align_up((size_t)tiles * (size_t)lo.nb * SBLKB, 256);
    lo.lr_bytes = align_up((size_t)lo.nb * (size_t)lo.NT * SBLKB, 256);
    // helper workgroups (SB == 64 only): accumulators per (tile, step), the helpers' own copy of the Err blocks, flags
    const size_t dt = SPLIT ? 0 : (size_t)std::min<int64_t>(tiles, DUO_MAX_TILES);
    lo.facc_bytes = 2 * align_up(dt * (size_t)(lo.nb + 1) * SBLKB, 256);  // (x 2: one set per helper of a tile)
    lo.errh_bytes = 2 * align_up(dt * (size_t)lo.nb * SBLKB, 256);
    lo.ctrl_bytes = align_up((size_t)DUO_CTRL_WORDS * sizeof(uint32_t), 256);
    lo.total = lo.errt_bytes + lo.lr_bytes + lo.facc_bytes + lo.errh_bytes + lo.ctrl_bytes;
    return lo;
}

static int solve_check(const char* who, int64_t m, int64_t n, int V, int64_t ldl) {
    if (V < 2 || V > 16) return fail(-2, "%s: V=%d not supported (bits 2..4 are implemented; bits=8 is not)", who, V);
    if (m > INT32_MAX / 2 || n > INT32_MAX / 2) return fail(-1, "%s: shape too large", who);
    if (ldl < n) return fail(-1, "%s: ldl=%lld < n=%lld", who, (long long)ldl, (long long)n);
    // packed L: nb * 4 nb blocks of 4 KB behind 32-bit buffer offsets
    const int64_t nb = (n + SB - 1) / SB;
    if (nb * SKR * nb * SBLKB >= (1ll << 32)) return fail(-1, "%s: n=%lld exceeds the 4 GiB buffer window of the packed L", who, (long long)n);
    return 0;
}

// packed copy of L into the workspace (depends on L only: the loop driver does it once per layer)
int solve_s_pack_l(const float* L, int64_t ldl, int64_t m, int64_t n, void* workspace, hipStream_t stream) {
    const SolveLayout lo = solve_layout(m, n);
    float* Lr = reinterpret_cast<float*>(static_cast<char*>(workspace) + lo.errt_bytes);
    int rc = ganq_hip_selftest(stream);
    if (rc) return rc;
    ProfScope prof(KID_T_PREP, stream);  // per-layer preparation, reported with the T-update's
    uint32_t* ctrl = reinterpret_cast<uint32_t*>(static_cast<char*>(workspace) + lo.errt_bytes + lo.lr_bytes + lo.facc_bytes + lo.errh_bytes);
    hipLaunchKernelGGL(l_pack_kernel, dim3((unsigned)lo.NT, (unsigned)lo.nb), dim3(64), 0, stream, L, ldl, (int)n, lo.NT,
                       mfma_k_ascending(), Lr, ctrl, DUO_CTRL_WORDS);
    GANQ_LAUNCH_CHECK();
    return 0;
}

// the solve proper; the workspace already holds the packed L
int solve_s_launch(const float* W, const float* L, int64_t ldl, const float* T, int64_t m, int64_t n, int V, uint8_t* Q_out,
                   float* Err_out, void* workspace, hipStream_t stream, const int* rowlist, const int* nactive, bool allow_helpers) {
    const SolveLayout lo = solve_layout(m, n);
    float* errt = static_cast<float*>(workspace);
    const float* Lr = reinterpret_cast<const float*>(static_cast<char*>(workspace) + lo.errt_bytes);
    const int tiles = (int)((m + SR - 1) / SR);
    // the Err blocks of the top SOLVE_LDS_PANELS panels (the ones every later panel re-reads) stay in LDS -- as many of them
    // as the 160 KB of a CU leave next to the kernel's static buffers (asked of the runtime once: the static size moves with
    // the compiler and the build options, and the margin at SOLVE_LDS_PANELS is a few hundred bytes)
    static const int lds_panels = [] {
        hipFuncAttributes fa{};
        int fit = SOLVE_LDS_PANELS;
        if (hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(solve_s_kernel<true>)) == hipSuccess) {
            hipFuncAttributes fb{};
            size_t stat = fa.sharedSizeBytes;
            if (hipFuncGetAttributes(&fb, reinterpret_cast<const void*>(solve_s_kernel<false>)) == hipSuccess)
                stat = std::max(stat, fb.sharedSizeBytes);
            const size_t room = stat < 160u * 1024u ? 160u * 1024u - stat : 0u;
            fit = (int)std::min<size_t>((size_t)SOLVE_LDS_PANELS, room / SBLKB);
        }
        return fit;
    }();
    const int pbase = std::max(0, lo.nb - lds_panels);
    const size_t smem = (size_t)(lo.nb - pbase) * SBLKB;
    {
        int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(solve_s_kernel<true>), smem);
        if (!rc) rc = ensure_dynamic_lds(reinterpret_cast<const void*>(solve_s_kernel<false>), smem);
        if (rc) return rc;
    }
    // helper workgroups (see "duo" at the kernel): asked for when the launch could have at most half as many active tiles as
    // the chip has CUs and the chains are long enough to matter; the kernel decides by the rows really active
    char* wsb = static_cast<char*>(workspace);
    float* facc = reinterpret_cast<float*>(wsb + lo.errt_bytes + lo.lr_bytes);
    float* errh = reinterpret_cast<float*>(wsb + lo.errt_bytes + lo.lr_bytes + lo.facc_bytes);
    uint32_t* ctrl = reinterpret_cast<uint32_t*>(wsb + lo.errt_bytes + lo.lr_bytes + lo.facc_bytes + lo.errh_bytes);
    const int ncu = current_device_cus();
    int duo_pol = 0, trio_pol = 0, grid = tiles;
    const int64_t facc_stride = (int64_t)(lo.facc_bytes / 2 / sizeof(float)), errh_stride = (int64_t)(lo.errh_bytes / 2 / sizeof(float));
    if (!SPLIT && allow_helpers && opt_get(OPT_SOLVE_DUO) != 0 && ncu >= 16 && lo.nb <= 1000) {
        const int xa = (int)opt_get(OPT_SOLVE_DUO_XA), xb = (int)opt_get(OPT_SOLVE_DUO_XB), xmin = (int)opt_get(OPT_SOLVE_DUO_XMIN),
                  cmin = (int)opt_get(OPT_SOLVE_DUO_CMIN);
        const int pol = (xa & 255) | ((xb & 255) << 8) | ((std::max(xmin, 1) & 255) << 16) | ((std::max(cmin, 1) & 127) << 24);
        if (lo.nb - 2 >= std::max(cmin, 1)) {  // some chain is long enough to be shared
            duo_pol = opt_get(OPT_SOLVE_DUO) == 2 ? (int)((unsigned)pol | 0x80000000u) : pol;  // 2 (tests): mute helpers, see the kernel
            grid = std::max(tiles, std::min(16 * ((tiles + 7) / 8), ncu / 16 * 16));
            // two helpers per tile where a third of the chip holds the tiles: its own split policy (the tile keeps less)
            if (opt_get(OPT_SOLVE_TRIO) != 0 && ncu >= 24) {
                const int xa3 = (int)opt_get(OPT_SOLVE_TRIO_XA), xb3 = (int)opt_get(OPT_SOLVE_TRIO_XB);
                trio_pol = (xa3 & 255) | ((xb3 & 255) << 8) | ((std::max(xmin, 1) & 255) << 16) | ((std::max(cmin, 1) & 127) << 24);
                grid = std::max(grid, std::min(24 * ((tiles + 7) / 8), ncu / 24 * 24));
            }
        }
    }
    static std::atomic<uint32_t> epoch{0};
    const uint32_t tag = ((epoch.fetch_add(1) + 1u) & 0x3fffffu) << 10;  // per-launch tag of the flags (never 0)
    ProfScope prof(KID_SOLVE_S, stream);
    int fast = opt_get(OPT_SOLVE_VARIANT) == 1 ? 0 : 1;  // GANQ_SOLVE_VARIANT=1: reduction path only (A/B, tests)
    // bit 1: the prefetch wave.  It pays where one workgroup per CU walks a long L (measured: 2048 x 8192 3.26 -> 2.91 ms;
    // 4096 x 4096 unchanged; with several workgroups per CU, 14336 x 4096, it costs 5 %: they cover each other's misses)
    if (SOLVE_PF && tiles <= 256 && n > 4096) fast |= 2;
    if (mfma_k_ascending()) {
        hipLaunchKernelGGL(solve_s_kernel<true>, dim3(grid), dim3(SOLVE_THREADS), smem, stream, W, L, ldl, Lr, lo.NT, T, (int)m, (int)n,
                           V, Q_out, Err_out, errt, pbase, rowlist, nactive, fast, ctrl, facc, errh, tag, duo_pol, ncu, trio_pol, facc_stride, errh_stride);
    } else {
        hipLaunchKernelGGL(solve_s_kernel<false>, dim3(grid), dim3(SOLVE_THREADS), smem, stream, W, L, ldl, Lr, lo.NT, T, (int)m, (int)n,
                           V, Q_out, Err_out, errt, pbase, rowlist, nactive, fast, ctrl, facc, errh, tag, duo_pol, ncu, trio_pol, facc_stride, errh_stride);
    }
    GANQ_LAUNCH_CHECK();
    return 0;
}

}  // namespace ganq

using namespace ganq;

#ifdef GANQ_SOLVE_TRACE
extern "C" int ganq_debug_solve_trace(unsigned long long* out) {  // [2][320][6]
    GANQ_HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(ganq::g_solve_trace), 2 * 320 * 6 * sizeof(unsigned long long)));
    return 0;
}
#endif

#ifdef GANQ_SOLVE_DEBUG
extern "C" int ganq_debug_solve_counters(unsigned long long* out4) {
    GANQ_HIP_CHECK(hipMemcpyFromSymbol(out4, HIP_SYMBOL(ganq::g_solve_dbg), 4 * sizeof(unsigned long long)));
    unsigned long long z[4] = {0, 0, 0, 0};
    GANQ_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(ganq::g_solve_dbg), z, sizeof(z)));
    return 0;
}
#endif

extern "C" size_t ganq_solve_s_workspace_bytes(int64_t m, int64_t n, int V) {
    (void)V;
    if (m <= 0 || n <= 0) return 0;
    return solve_layout(m, n).total;
}

extern "C" int ganq_solve_s(const float* W, const float* L, int64_t ldl, const float* T, int64_t m, int64_t n,
                            int V, uint8_t* Q_out, float* Err_out, void* workspace, size_t workspace_bytes,
                            void* stream_) {
    if (m < 0 || n < 0) return fail(-1, "ganq_solve_s: negative shape m=%lld n=%lld", (long long)m, (long long)n);
    if (m == 0 || n == 0) return 0;
    int rc = solve_check("ganq_solve_s", m, n, V, ldl);
    if (rc) return rc;
    if (!W || !L || !T || !Q_out) return fail(-3, "ganq_solve_s: null pointer");
    const size_t need = ganq_solve_s_workspace_bytes(m, n, V);
    if (!workspace || workspace_bytes < need)
        return fail(-4, "ganq_solve_s: workspace %zu B < required %zu B", workspace_bytes, need);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    rc = solve_s_pack_l(L, ldl, m, n, workspace, stream);
    if (rc) return rc;
    return solve_s_launch(W, L, ldl, T, m, n, V, Q_out, Err_out, workspace, stream);
}

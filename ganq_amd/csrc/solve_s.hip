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
#include <type_traits>
#include <utility>

#include "common.h"

namespace ganq {

constexpr int SB = 64;   // panel width (columns)
constexpr int SR = 16;   // rows per workgroup
constexpr int SPF = 16;  // k-groups per prefetch batch
constexpr int SOLVE_LDS_COLS = 1792;  // columns of Err kept in LDS (112 KB next to the 46 KB of panel buffers)

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

struct PanelState {
    float r[4];     // running residual sums of this lane's 4 panel columns (c16 + 16k)
    float w[4];     // W[row][j0 + c16 + 16k]
    float e[4];     // err captured at this lane's columns
    uint32_t q[4];  // index captured at this lane's columns
    float tv;       // T[row][c16] (+inf beyond V)
    uint32_t c16;
};

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
    const uint32_t d = __builtin_bit_cast(uint32_t, eff - st.tv) & 0x7fffffffu;  // |eff - T[v]| as ordered bits
    const uint32_t dmin = row_min_u(d);
    const uint32_t cand = (d == dmin) ? st.c16 : 255u;
    const uint32_t idx = row_min_u(cand);  // first minimum
    const uint32_t tb = row_or_u((st.c16 == idx) ? __builtin_bit_cast(uint32_t, st.tv) : 0u);
    const float err = wj - __builtin_bit_cast(float, tb);
    st.r[0] = fmaf(err, lrow.x, st.r[0]);
    st.r[1] = fmaf(err, lrow.y, st.r[1]);
    st.r[2] = fmaf(err, lrow.z, st.r[2]);
    st.r[3] = fmaf(err, lrow.w, st.r[3]);
    if (st.c16 == (uint32_t)OWN) {
        st.q[KREG] = idx;
        st.e[KREG] = err;
    }
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
#ifdef GANQ_SOLVE_DEBUG
__device__ unsigned long long ss_dbg[8];
#define SS_T() __builtin_amdgcn_s_memtime()
#define SS_ADD(slot, t0, who) do { if (tid == (who)) atomicAdd(&ss_dbg[slot], __builtin_amdgcn_s_memtime() - (t0)); } while (0)
#else
#define SS_T() 0ull
#define SS_ADD(slot, t0, who) do {} while (0)
#endif
// keeps a batch of loads where it was written: the memory clobber stops IR-level load motion across stage boundaries,
// the sched_barrier stops the machine scheduler
#define GANQ_PIN() do { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
template <bool KASC, int DBG = 0>
__global__ __launch_bounds__(512) void solve_s_kernel(const float* __restrict__ W, const float* __restrict__ L,
                                                      int64_t ldl, const float* __restrict__ T, int m, int n, int V,
                                                      uint8_t* __restrict__ Q, float* __restrict__ ErrOut,
                                                      float* __restrict__ ErrT, int cbase) {
    __shared__ float4 Ld[2][SB][16];      // panel triangle of L, [jj][c16][k] <-> L[j0+jj][j0 + c16 + 16k]
    __shared__ float Rp[2][SR][SB + 4];   // residual panel handed from (G) to (P)
    __shared__ float2 Dg[2][SB];          // {L[j][j], 1 / L[j][j]} of the panel's columns
    __shared__ float ErrP[SB][SR];        // Err of the panel just solved, [col][row] (zero beyond the panel's width)
    extern __shared__ __align__(16) float ErrL[];  // [n - cbase][SR]: Err of the columns >= cbase, the A operand's hot part

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool roleG = wv >= 4;
    const int gw = wv & 3;
    const int gtid = tid & 255;  // thread index inside the role
    const int tile = blockIdx.x;
    const int rsub = lane >> 4;
    const int c16 = lane & 15;
    const int prow_in_tile = 4 * gw + rsub;  // row handled by this 16-lane group in phase (P)
    const int prow = min(tile * SR + prow_in_tile, m - 1);
    const bool prow_ok = tile * SR + prow_in_tile < m;
    float* __restrict__ errt = ErrT + (int64_t)tile * n * SR;

    PanelState st;
    st.c16 = (uint32_t)c16;
    st.tv = (c16 < V) ? T[(int64_t)prow * V + c16] : __builtin_inff();
    float wnext[4] = {0.f, 0.f, 0.f, 0.f};

    const int ksub = lane >> 4;
    const int kslot = KASC ? (3 - ksub) : ksub;  // column inside a k-group handled by this lane's MFMA slice
    const uint32_t laneA = (uint32_t)(kslot * SR + c16);

    const int nb = (n + SB - 1) / SB;
    for (int s = 0; s <= nb; ++s) {
        const int bP = nb - s;      // panel solved in this step (none at s = 0)
        const int bG = nb - 1 - s;  // panel whose residual is produced in this step (none at s = nb)
        f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
        float bpre[16];  // (G) L[(bG+1)*64 + 4g + kslot][colB], g = 0..15: the B operands of part 2
        [[maybe_unused]] const unsigned long long t_step = SS_T();
        if (!roleG) {
            // ---- (P) ---------------------------------------------------------------------------------------
            if (bP <= nb - 1) {
                const int j0 = bP * SB;
                const int wd = min(SB, n - j0);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    st.w[k] = wnext[k];
                    st.q[k] = 0;
                    st.e[k] = 0.0f;
                    st.r[k] = Rp[bP & 1][prow_in_tile][c16 + 16 * k];
                }
                if (bP >= 1) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) wnext[k] = W[(int64_t)prow * n + j0 - SB + c16 + 16 * k];  // full panel
                }
                if constexpr (DBG == 5) {
                    // developer experiment: no panel steps at all (G role alone)
                } else if (wd == SB) {
                    panel_all<true>(st, Ld[bP & 1], Dg[bP & 1], wd, std::make_integer_sequence<int, SB>{});
                } else {
                    panel_all<false>(st, Ld[bP & 1], Dg[bP & 1], wd, std::make_integer_sequence<int, SB>{});
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int col = c16 + 16 * k;
                    ErrP[col][prow_in_tile] = (col < wd) ? st.e[k] : 0.0f;
                    if (col < wd && j0 >= cbase) ErrL[(j0 - cbase + col) * SR + prow_in_tile] = st.e[k];
                    if (col < wd) {
                        errt[(int64_t)(j0 + col) * SR + prow_in_tile] = st.e[k];
                        if (prow_ok) {
                            Q[(int64_t)prow * n + j0 + col] = (uint8_t)min(st.q[k], (uint32_t)(V - 1));
                            if (ErrOut) ErrOut[(int64_t)prow * n + j0 + col] = st.e[k];
                        }
                    }
                }
            } else {
                const int j0 = (nb - 1) * SB;  // W of the first (possibly partial) panel
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int col = c16 + 16 * k;
                    wnext[k] = (j0 + col < n) ? W[(int64_t)prow * n + j0 + col] : 0.0f;
                }
            }
        } else if (bG >= 0) {
            // ---- (G) part 1: columns right of panel bG+1, descending ----------------------------------------
            const int j0 = bG * SB;
            const int wd = min(SB, n - j0);
            const int colB = j0 + 16 * gw + c16;  // < n whenever a product with it is used
            // prefetches for later in this step: the panel's own block of L, its diagonal, the B operands of part 2
            float lpre[(SB * SB) / 256];
#pragma unroll
            for (int e = 0; e < (SB * SB) / 256; ++e) {
                const int idx = e * 256 + gtid;
                const int jj = idx >> 6, col = idx & 63;
                lpre[e] = (jj < wd && col < wd) ? L[(int64_t)(j0 + jj) * ldl + j0 + col] : 0.0f;
            }
            float dpre = 1.0f;
            if (gtid < SB && gtid < wd) dpre = L[(int64_t)(j0 + gtid) * ldl + j0 + gtid];
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int u = j0 + SB + 4 * g + kslot;
                bpre[g] = (u < n && colB < n) ? L[(int64_t)u * ldl + colB] : 0.0f;
            }
            {
                const int gbot = (j0 + 2 * SB) >> 2;
                if ((n & 3) && (n >> 2) >= gbot) {  // ragged top group: columns >= n contribute nothing
                    const int u = 4 * (n >> 2) + kslot;
                    const bool ok = u < n;
                    const int uu = ok ? u : (n - 1);
                    const float av = errt[(int64_t)uu * SR + c16];
                    const float bv = L[(int64_t)uu * ldl + colB];
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ok ? av : 0.0f, ok ? bv : 0.0f, acc, 0, 0, 0);
                }
                const int gtop = (n >> 2) - 1;     // highest full group
                const uint32_t laneB = (uint32_t)kslot * (uint32_t)ldl + (uint32_t)colB;
                const uint32_t laneB4 = 4u * laneB;  // byte offset, < 2^32 (host check on ldl)
                const __amdgpu_buffer_rsrc_t rsrcL = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(L), 0, 0xffffffff, 0x00020000);
                const __amdgpu_buffer_rsrc_t rsrcE = __builtin_amdgcn_make_buffer_rsrc(errt, 0, 0xffffffff, 0x00020000);
                // zero records: every load through it is out of range and returns 0
                const __amdgpu_buffer_rsrc_t rsrcZ = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(L), 0, 0, 0x00020000);
                // One chain segment: groups ghi, ghi-1, .., glo (descending), A from LDS (columns >= cbase) or from the
                // global transposed scratch.  Batches of SPF groups, operands loaded two batches ahead into three
                // rotating register sets; every load is unconditional (indices clamped to the segment) so that the
                // waits in the steady-state loop count exactly the loads still allowed in flight.
                auto chain = [&](auto a_in_lds, int ghi, int glo) {
                    constexpr bool ALDS = decltype(a_in_lds)::value;
                    const int total = ghi - glo + 1;
                    if (total <= 0) return;
                    const int nbat = total / SPF;
                    const float* __restrict__ Al = ErrL + (int64_t)laneA - (int64_t)cbase * SR;
                    // buffer loads: (resource, per-lane byte offset in a VGPR, per-group byte offset in an SGPR) -- one
                    // scalar add and one VMEM instruction per operand, no per-lane 64-bit address arithmetic
                    const uint32_t bstride_b = (uint32_t)(16 * ldl);  // bytes between the B rows of consecutive groups
                    // Batches past the end (the loop below always runs whole rounds of three) multiply Err by zeros: their B
                    // operand comes through rsrcZ (zero records: out-of-range buffer loads return 0), which leaves the accumulator as it is.
                    auto ld = [&](int bi, float (&aa)[SPF], float (&bb)[SPF]) {
                        // lowest group of the batch; everything else is a compile-time multiple of a stride above it
                        const bool real = bi < nbat;
                        const int glow = ghi - min(bi, nbat - 1) * SPF - (SPF - 1);
                        uint32_t sB = real ? (uint32_t)glow * bstride_b : 0u;
                        const uint32_t stepB = real ? bstride_b : 0u;
                        const __amdgpu_buffer_rsrc_t rsB = real ? rsrcL : rsrcZ;
                        uint32_t sA = (uint32_t)glow * (uint32_t)(4 * SR * sizeof(float));
                        const float* __restrict__ Ap = Al + glow * (4 * SR);
#pragma unroll
                        for (int i = 0; i < SPF; ++i) {
                            if constexpr (DBG == 1) bb[i] = 1.0f + (float)bi;
                            else bb[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsB, (int)laneB4, (int)sB, 0));
                            sB += stepB;
                            if constexpr (DBG == 2) aa[i] = 1.0f + (float)bi;
                            else if constexpr (ALDS) aa[i] = Ap[i * (4 * SR)];
                            else aa[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrcE, (int)(4 * laneA), (int)sA, 0));
                            sA += 4 * SR * sizeof(float);
                        }
                    };
                    auto mm = [&](const float (&aa)[SPF], const float (&bb)[SPF]) {
#pragma unroll
                        for (int i = SPF - 1; i >= 0; --i) {
                            if constexpr (DBG == 3) acc[0] = fmaf(aa[i], bb[i], acc[0]);
                            else acc = __builtin_amdgcn_mfma_f32_16x16x4f32(aa[i], bb[i], acc, 0, 0, 0);
                        }
                    };
                    if (nbat > 0) {
                        float a0[SPF], b0[SPF], a1[SPF], b1[SPF], a2[SPF], b2[SPF];
                        ld(0, a0, b0);
                        ld(1, a1, b1);
                        GANQ_PIN();
                        // One stage = the loads of batch k+2 and the 16 MFMAs of batch k, interleaved one load (two when A
                        // also comes from memory) per MFMA: a lone wave issues a VMEM instruction every ~16 cycles and a
                        // dependent MFMA every ~40, so loads issued as a block in front of the MFMAs leave the matrix
                        // pipe idle for a third of the stage.
                        auto stage_sched = [&]() {
#pragma unroll
                            for (int i = 0; i < SPF / 2; ++i) {
                                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                   // MFMA
                                __builtin_amdgcn_sched_group_barrier(0x020, ALDS ? 1 : 2, 0);        // VMEM read
                                if (ALDS) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);         // DS read
                                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                                __builtin_amdgcn_sched_group_barrier(0x020, ALDS ? 1 : 2, 0);
                            }
                        };
                        for (int bi = 0; bi < nbat; bi += 3) {  // whole rounds, no exits inside: a plain counted loop
                            ld(bi + 2, a2, b2);
                            mm(a0, b0);
                            stage_sched();
                            GANQ_PIN();
                            ld(bi + 3, a0, b0);
                            mm(a1, b1);
                            stage_sched();
                            GANQ_PIN();
                            ld(bi + 4, a1, b1);
                            mm(a2, b2);
                            stage_sched();
                            GANQ_PIN();
                        }
                    }
                    for (int g = ghi - nbat * SPF; g >= glo; --g) {  // fewer than SPF groups left (ragged n only)
                        const float bv = (L + (int64_t)g * 4 * ldl)[laneB];
                        const float av = ALDS ? Al[g * (4 * SR)] : (errt + (int64_t)g * (4 * SR))[laneA];
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0);
                    }
                };
                const int glds = max(gbot, cbase >> 2);  // lowest group whose Err columns live in LDS
                chain(std::true_type{}, gtop, glds);
                chain(std::false_type{}, min(gtop, glds - 1), gbot);
            }
#pragma unroll
            for (int e = 0; e < (SB * SB) / 256; ++e) {
                const int idx = e * 256 + gtid;
                const int jj = idx >> 6, col = idx & 63;
                reinterpret_cast<float*>(&Ld[bG & 1][jj][col & 15])[col >> 4] = lpre[e];
            }
            if (gtid < SB) Dg[bG & 1][gtid] = make_float2(dpre, 1.0f / dpre);
        }
        SS_ADD(0, t_step, 0);    // P work
        SS_ADD(1, t_step, 256);  // G part 1
        __syncthreads();  // panel bP solved (ErrP, ErrT visible); part 1 of panel bG done
        SS_ADD(2, t_step, 0);    // step up to barrier A
        if (roleG && bG >= 0) {
            // ---- (G) part 2: the 64 columns of panel bG+1, descending; then publish R ------------------------
            if (bG + 1 <= nb - 1) {
#pragma unroll
                for (int g = 15; g >= 0; --g)
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ErrP[4 * g + kslot][c16], bpre[g], acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) Rp[bG & 1][rsub * 4 + r][16 * gw + c16] = acc[r];
        }
        __syncthreads();  // R of panel bG, its Ld / Dg ready; ErrP free
    }
}

}  // namespace ganq

using namespace ganq;

extern "C" size_t ganq_solve_s_workspace_bytes(int64_t m, int64_t n, int V) {
    (void)V;
    if (m <= 0 || n <= 0) return 0;
    const int64_t tiles = (m + SR - 1) / SR;
    return align_up((size_t)tiles * (size_t)n * SR * sizeof(float), 256);
}

extern "C" int ganq_solve_s(const float* W, const float* L, int64_t ldl, const float* T, int64_t m, int64_t n,
                            int V, uint8_t* Q_out, float* Err_out, void* workspace, size_t workspace_bytes,
                            void* stream_) {
    if (m < 0 || n < 0) return fail(-1, "ganq_solve_s: negative shape m=%lld n=%lld", (long long)m, (long long)n);
    if (m == 0 || n == 0) return 0;
    if (V < 2 || V > 16)
        return fail(-2, "ganq_solve_s: V=%d not supported (bits 2..4 are implemented; bits=8 is not)", V);
    if (m > INT32_MAX / 2 || n > INT32_MAX / 2) return fail(-1, "ganq_solve_s: shape too large");
    if (ldl < n) return fail(-1, "ganq_solve_s: ldl=%lld < n=%lld", (long long)ldl, (long long)n);
    if (ldl * n >= (1ll << 30)) return fail(-1, "ganq_solve_s: L of %lld x %lld floats exceeds the 4 GiB buffer window", (long long)n, (long long)ldl);
    if (!W || !L || !T || !Q_out) return fail(-3, "ganq_solve_s: null pointer");
    const size_t need = ganq_solve_s_workspace_bytes(m, n, V);
    if (!workspace || workspace_bytes < need)
        return fail(-4, "ganq_solve_s: workspace %zu B < required %zu B", workspace_bytes, need);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    int rc = ganq_hip_selftest(stream_);
    if (rc) return rc;
    const int tiles = (int)((m + SR - 1) / SR);
    float* errt = static_cast<float*>(workspace);
    // the top SOLVE_LDS_COLS columns of Err (the ones every later panel re-reads) stay in LDS
    const int cbase = (int)std::max<int64_t>(0, (n - SOLVE_LDS_COLS + SB - 1) / SB * SB);
    const size_t smem = (size_t)(n - cbase) * SR * sizeof(float);
    static size_t attr_smem = 0;
    if (smem > attr_smem) {
        GANQ_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(solve_s_kernel<true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        GANQ_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(solve_s_kernel<false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr_smem = smem;
    }
    ProfScope prof(KID_SOLVE_S, stream);
#ifdef GANQ_SOLVE_DEBUG
    if (const char* dm = getenv("GANQ_SOLVE_DBG")) {
        const int d = atoi(dm);
#define GANQ_DBG_LAUNCH(D) hipLaunchKernelGGL((solve_s_kernel<true, D>), dim3(tiles), dim3(512), smem, stream, W, L, ldl, T, (int)m, (int)n, V, Q_out, Err_out, errt, cbase)
        GANQ_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(solve_s_kernel<true, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        GANQ_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(solve_s_kernel<true, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        GANQ_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(solve_s_kernel<true, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        GANQ_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(solve_s_kernel<true, 5>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        if (d == 1) GANQ_DBG_LAUNCH(1);
        else if (d == 2) GANQ_DBG_LAUNCH(2);
        else if (d == 3) GANQ_DBG_LAUNCH(3);
        else if (d == 5) GANQ_DBG_LAUNCH(5);
        else GANQ_DBG_LAUNCH(0);
        GANQ_LAUNCH_CHECK();
        return 0;
    }
#endif
    if (mfma_k_ascending()) {
        hipLaunchKernelGGL(solve_s_kernel<true>, dim3(tiles), dim3(512), smem, stream, W, L, ldl, T, (int)m, (int)n, V,
                           Q_out, Err_out, errt, cbase);
    } else {
        hipLaunchKernelGGL(solve_s_kernel<false>, dim3(tiles), dim3(512), smem, stream, W, L, ldl, T, (int)m, (int)n, V,
                           Q_out, Err_out, errt, cbase);
    }
    GANQ_LAUNCH_CHECK();
    return 0;
}

#ifdef GANQ_SOLVE_DEBUG
extern "C" int ganq_debug_solve_cycles(unsigned long long* out8) {
    GANQ_HIP_CHECK(hipDeviceSynchronize());
    GANQ_HIP_CHECK(hipMemcpyFromSymbol(out8, HIP_SYMBOL(ganq::ss_dbg), 8 * sizeof(unsigned long long)));
    unsigned long long z[8] = {0};
    GANQ_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(ganq::ss_dbg), z, sizeof(z)));
    return 0;
}
#endif

// S-solve: the back-substitution assignment loop of GANQ (reference ganq.py:533-565; Metal
// kernel compute_s ganq.py:94-247).  Arithmetic contract: include/ganq_hip.h + oracle/ganq_oracle.c
// (ganq_oracle_solve_s) -- bit-exact indices.
//
// Decomposition (one workgroup = 16 rows of W for the whole solve, no inter-workgroup traffic):
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
#include <utility>

#include "common.h"

namespace ganq {

constexpr int SB = 64;   // panel width (columns)
constexpr int SR = 16;   // rows per workgroup
constexpr int SPF = 16;  // k-groups per prefetch batch

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
template <int JJ>
__device__ __forceinline__ void panel_step(PanelState& st, const float4 (*Ld)[16], const float2* Dg) {
    constexpr int KREG = JJ >> 4, OWN = JJ & 15;
    // r / L[j][j] as an exactly rounded quotient without the division sequence (Markstein): rinv = RN(1/L[j][j]) is
    // computed once per column with a true division; q0 = RN(r*rinv); e = fma(-q0, L, r) is exact; RN(q0 + e*rinv) is
    // the IEEE quotient (checked against the division by ganq_debug_div_check and, end to end, by the oracle tests)
    const float2 dg = Dg[JJ];  // {L[j][j], RN(1 / L[j][j])}, uniform LDS read
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
    const float4 lrow = Ld[JJ][st.c16];  // L[j][j0 + c16 + 16k], k = 0..3
    st.r[0] = fmaf(err, lrow.x, st.r[0]);
    st.r[1] = fmaf(err, lrow.y, st.r[1]);
    st.r[2] = fmaf(err, lrow.z, st.r[2]);
    st.r[3] = fmaf(err, lrow.w, st.r[3]);
    if (st.c16 == (uint32_t)OWN) {
        st.q[KREG] = idx;
        st.e[KREG] = err;
    }
}

template <bool FULL, int... I>
__device__ __forceinline__ void panel_all(PanelState& st, const float4 (*Ld)[16], const float2* Dg, int wd,
                                          std::integer_sequence<int, I...>) {
    // steps run from the panel's last column down to its first
    if constexpr (FULL) {
        (panel_step<SB - 1 - I>(st, Ld, Dg), ...);
    } else {
        ((SB - 1 - I < wd ? panel_step<SB - 1 - I>(st, Ld, Dg) : (void)0), ...);
    }
}

template <bool KASC>
__global__ __launch_bounds__(256) void solve_s_kernel(const float* __restrict__ W, const float* __restrict__ L,
                                                      int64_t ldl, const float* __restrict__ T, int m, int n, int V,
                                                      uint8_t* __restrict__ Q, float* __restrict__ ErrOut,
                                                      float* __restrict__ ErrT) {
    __shared__ float4 Ld[SB][16];      // panel triangle of L, [jj][c16][k] <-> L[j0+jj][j0 + c16 + 16k]
    __shared__ float Rp[SR][SB + 4];   // residual panel handed from (G) to (P)
    __shared__ float2 Dg[SB];          // {L[j][j], 1 / L[j][j]} of the panel's columns

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = tid >> 6;
    const int tile = blockIdx.x;
    const int rsub = lane >> 4;
    const int c16 = lane & 15;
    const int prow_in_tile = 4 * wv + rsub;  // row handled by this 16-lane group in phase (P)
    const int prow = min(tile * SR + prow_in_tile, m - 1);
    const bool prow_ok = tile * SR + prow_in_tile < m;
    float* __restrict__ errt = ErrT + (int64_t)tile * n * SR;

    PanelState st;
    st.c16 = (uint32_t)c16;
    st.tv = (c16 < V) ? T[(int64_t)prow * V + c16] : __builtin_inff();

    const int nb = (n + SB - 1) / SB;
    for (int b = nb - 1; b >= 0; --b) {
        const int j0 = b * SB;
        const int wd = min(SB, n - j0);

        // stage the panel's block of L into LDS
#pragma unroll
        for (int e = 0; e < (SB * SB) / 256; ++e) {
            const int idx = e * 256 + tid;
            const int jj = idx >> 6, col = idx & 63;
            float v = 0.0f;
            if (jj < wd && col < wd) v = L[(int64_t)(j0 + jj) * ldl + j0 + col];
            reinterpret_cast<float*>(&Ld[jj][col & 15])[col >> 4] = v;
        }
        if (tid < SB) {
            const float d = (tid < wd) ? L[(int64_t)(j0 + tid) * ldl + j0 + tid] : 1.0f;
            Dg[tid] = make_float2(d, 1.0f / d);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int col = c16 + 16 * k;
            st.w[k] = (col < wd) ? W[(int64_t)prow * n + j0 + col] : 0.0f;
            st.q[k] = 0;
            st.e[k] = 0.0f;
        }

        // ---- (G) residual GEMM: wave wv -> panel columns 16wv..16wv+15, all 16 rows -----------------
        // k-group g covers columns 4g..4g+3; inside one MFMA the slice order follows the probed k order so that
        // the chain always runs over columns in descending order.  Addresses are (uniform base) + (per-lane
        // 32-bit offset): scalar pointer arithmetic, one VMEM instruction per operand.
        f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
        {
            const int gbot = (j0 + SB) >> 2;
            const int ksub = lane >> 4;
            const int kslot = KASC ? (3 - ksub) : ksub;  // column inside the group handled by this lane's slice
            const int colB = j0 + 16 * wv + c16;         // < n whenever there is any group (only the last panel is partial)
            if ((n & 3) && (n >> 2) >= gbot) {           // ragged top group: columns >= n contribute nothing
                const int u = 4 * (n >> 2) + kslot;
                const bool ok = u < n;
                const int uu = ok ? u : (n - 1);
                const float av = errt[(int64_t)uu * SR + c16];
                const float bv = L[(int64_t)uu * ldl + colB];
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ok ? av : 0.0f, ok ? bv : 0.0f, acc, 0, 0, 0);
            }
            const int gtop = (n >> 2) - 1;               // highest full group
            const int nfull = gtop - gbot + 1;           // <= 0 for the last panel
            const uint32_t laneA = (uint32_t)(kslot * SR + c16);
            const uint32_t laneB = (uint32_t)kslot * (uint32_t)ldl + (uint32_t)colB;
            auto load_batch = [&](int g0, float (&a)[SPF], float (&bb)[SPF]) {  // groups g0, g0-1, .., g0-SPF+1
#pragma unroll
                for (int i = 0; i < SPF; ++i) {
                    const float* __restrict__ Ag = errt + (int64_t)(g0 - i) * (4 * SR);
                    const float* __restrict__ Bg = L + (int64_t)(g0 - i) * 4 * ldl;
                    a[i] = Ag[laneA];
                    bb[i] = Bg[laneB];
                }
            };
            float a0[SPF], b0[SPF], a1[SPF], b1[SPF];
            int g = gtop;
            int nbatch = nfull > 0 ? nfull / SPF : 0;
            if (nbatch > 0) {
                load_batch(g, a0, b0);
                while (true) {
                    if (nbatch > 1) load_batch(g - SPF, a1, b1);
#pragma unroll
                    for (int i = 0; i < SPF; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[i], b0[i], acc, 0, 0, 0);
                    g -= SPF;
                    if (--nbatch == 0) break;
                    if (nbatch > 1) load_batch(g - SPF, a0, b0);
#pragma unroll
                    for (int i = 0; i < SPF; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[i], b1[i], acc, 0, 0, 0);
                    g -= SPF;
                    if (--nbatch == 0) break;
                }
            }
            for (; g >= gbot; --g) {  // fewer than SPF groups left
                const float* __restrict__ Ag = errt + (int64_t)g * (4 * SR);
                const float* __restrict__ Bg = L + (int64_t)g * 4 * ldl;
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(Ag[laneA], Bg[laneB], acc, 0, 0, 0);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) Rp[rsub * 4 + r][16 * wv + c16] = acc[r];
        __syncthreads();

        // ---- (P) the panel's sequential steps ---------------------------------------------------------
#pragma unroll
        for (int k = 0; k < 4; ++k) st.r[k] = Rp[prow_in_tile][c16 + 16 * k];
        if (wd == SB) {
            panel_all<true>(st, Ld, Dg, wd, std::make_integer_sequence<int, SB>{});
        } else {
            panel_all<false>(st, Ld, Dg, wd, std::make_integer_sequence<int, SB>{});
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int col = c16 + 16 * k;
            if (col < wd) {
                errt[(int64_t)(j0 + col) * SR + prow_in_tile] = st.e[k];
                if (prow_ok) {
                    Q[(int64_t)prow * n + j0 + col] = (uint8_t)min(st.q[k], (uint32_t)(V - 1));
                    if (ErrOut) ErrOut[(int64_t)prow * n + j0 + col] = st.e[k];
                }
            }
        }
        __syncthreads();  // ErrT stores visible to the next panel's GEMM; Ld / Rp free for reuse
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
    if (!W || !L || !T || !Q_out) return fail(-3, "ganq_solve_s: null pointer");
    const size_t need = ganq_solve_s_workspace_bytes(m, n, V);
    if (!workspace || workspace_bytes < need)
        return fail(-4, "ganq_solve_s: workspace %zu B < required %zu B", workspace_bytes, need);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    int rc = ganq_hip_selftest(stream_);
    if (rc) return rc;
    const int tiles = (int)((m + SR - 1) / SR);
    float* errt = static_cast<float*>(workspace);
    ProfScope prof(KID_SOLVE_S, stream);
    if (mfma_k_ascending()) {
        hipLaunchKernelGGL(solve_s_kernel<true>, dim3(tiles), dim3(256), 0, stream, W, L, ldl, T, (int)m, (int)n, V,
                           Q_out, Err_out, errt);
    } else {
        hipLaunchKernelGGL(solve_s_kernel<false>, dim3(tiles), dim3(256), 0, stream, W, L, ldl, T, (int)m, (int)n, V,
                           Q_out, Err_out, errt);
    }
    GANQ_LAUNCH_CHECK();
    return 0;
}

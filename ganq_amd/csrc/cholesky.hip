// Lower Cholesky factorisation A = L L^T in fp32, in place, for the prologue inherited from GPTQ (reference
// gptq.py:280-309: torch.linalg.cholesky of the damped Hessian; the S-solve consumes L, the k-means weights and the
// per-weight losses consume the diagonal of the upper factor of H^-1).
//
// Blocked right-looking, block 128, three kernels per block column j (everything stays on the stream, no host sync):
//   chol_diag_kernel   one workgroup: the 128x128 diagonal block as 4x4 blocks of 32 -- each 32x32 diagonal block is
//                      factored and inverted by one wave in registers (IEEE sqrt / divide, v_readlane broadcasts),
//                      the rest of the block follows on the matrix cores;
//   chol_panel_kernel  panel:   L21 = A21 L11^-T       (block-wise substitution with the four 32x32 inverses)
//   chol_update_kernel update:  A22 -= L21 L21^T       (lower-triangular 128x128 tiles, K = 128 held in LDS)
// All products are C = A B^T on v_mfma_f32_32x32x2_f32.
// Pivots use v_rsq_f32 + one Newton step (accurate to fp32 rounding, not correctly rounded).
// A non-positive pivot sets *info (1-based column, like LAPACK) and poisons the factor with NaN; the host wrapper
// reads info once at the end.
#include <cstdlib>

#include <mutex>

#include "common.h"

namespace ganq {

constexpr int CB = 128;
constexpr int CP = CB + 1;  // LDS row pitch: lanes walking down a column hit distinct banks

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int XP = 33;  // pitch of the 32x32 diagonal inverses in LDS

__device__ __forceinline__ float rl(float v, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
__device__ __forceinline__ void wsync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// One wave, 32x32 blocks in LDS: acc (+/-)= A B^T, i.e. C[i][j] += sign * sum_k A[i][k] B[j][k].
// C layout of v_mfma_f32_32x32x2_f32: column = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
__device__ __forceinline__ void mm32(f32x16& acc, const float* A, int pa, const float* B, int pb, bool neg, int lane) {
    const int i32 = lane & 31, kk = lane >> 5;
#pragma unroll
    for (int k = 0; k < 32; k += 2) {
        float a = A[i32 * pa + k + kk];
        const float b = B[i32 * pb + k + kk];
        if (neg) a = -a;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
}
__device__ __forceinline__ void acc_load(f32x16& acc, const float* C, int pc, int lane) {
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = C[((e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)) * pc + (lane & 31)];
}
__device__ __forceinline__ void acc_store(const f32x16& acc, float* C, int pc, int lane) {
#pragma unroll
    for (int e = 0; e < 16; ++e) C[((e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)) * pc + (lane & 31)] = acc[e];
}

// 128 rows x 128 k of A (row0.., columns j..j+nb) -> LDS block, zero beyond nb / n.  All 16 row segments of a
// thread are requested before the first is stored: the block comes from L2 / HBM, one round trip instead of 16.
__device__ __forceinline__ void load_block(float (*dst)[CP], const float* __restrict__ A, int64_t lda, int row0, int nrows_valid,
                                           int j, int nb, int tid, bool lower_only) {
    float4 v[16];
    const int k4 = (tid & 31) * 4;
    const bool vec = ((lda & 3) == 0) && ((j & 3) == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int r = (tid >> 5) + 8 * e;
        float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < nrows_valid) {
            const float* p = A + (int64_t)(row0 + r) * lda + j + k4;
            if (vec && k4 + 3 < nb) {
                x = *reinterpret_cast<const float4*>(p);
            } else {
                if (k4 + 0 < nb) x.x = p[0];
                if (k4 + 1 < nb) x.y = p[1];
                if (k4 + 2 < nb) x.z = p[2];
                if (k4 + 3 < nb) x.w = p[3];
            }
        }
        v[e] = x;
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int r = (tid >> 5) + 8 * e;
        float w[4] = {v[e].x, v[e].y, v[e].z, v[e].w};
#pragma unroll
        for (int t = 0; t < 4; ++t) dst[r][k4 + t] = (lower_only && k4 + t > r) ? 0.0f : w[t];
    }
}

// Diagonal 128x128 block as 4x4 blocks of 32.  Per block column kb: wave 0 factors the 32x32 diagonal block with one
// row per lane in registers (pivot / column broadcasts are v_readlane, no LDS round trips) and inverts it in the same
// instructions (one column of the inverse per lane of the upper half wave); the blocks below are multiplied by that inverse and the trailing blocks
// updated on the matrix cores.  Only the four 32x32 inverses leave the kernel: the panel kernel solves block-wise.
// jprev >= 0 (look-ahead driver): the block has not yet received the PREVIOUS step's update -- it is applied here, S -= P P^T with
// P = A[j.., jprev .. jprev + 128) (that step's first panel block), so that this kernel depends on the previous panel kernel only
// and the previous step's whole trailing update runs beside it on the second stream.
__global__ __launch_bounds__(256) void chol_diag_kernel(float* __restrict__ A, int64_t lda, int j, int nb, int jprev,
                                                        float* __restrict__ Xout, int* __restrict__ info) {
    extern __shared__ __align__(16) float sm[];
    float(*S)[CP] = reinterpret_cast<float(*)[CP]>(sm);  // the block, lower part
    float* Xd = sm + CB * CP;                              // [4][32][XP]
    float(*Pp)[CP] = reinterpret_cast<float(*)[CP]>(sm + CB * CP + 4 * 32 * XP);  // (jprev >= 0 only: the launch then has the LDS for it)
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (jprev >= 0) load_block(Pp, A, lda, j, nb, jprev, CB, tid, false);
    // (all 16 row segments of a thread in flight at once: the element-wise loop this replaces was 64 dependent round trips to
    // L2 -- a third of the kernel's 64 us in round 3)
    load_block(S, A, lda, j, nb, j, nb, tid, true);
    __syncthreads();
    if (jprev >= 0) {  // the ten 32 x 32 blocks of the lower triangle, K = 128, dealt over the four waves
        int p = 0;
        for (int i = 0; i < 4; ++i)
            for (int jb = 0; jb <= i; ++jb, ++p) {
                if ((p & 3) != wv) continue;
                f32x16 acc;
                acc_load(acc, &S[32 * i][32 * jb], CP, lane);
                for (int k = 0; k < 4; ++k) mm32(acc, &Pp[32 * i][32 * k], CP, &Pp[32 * jb][32 * k], CP, true, lane);
                acc_store(acc, &S[32 * i][32 * jb], CP, lane);
            }
        __syncthreads();
    }
    if (tid >= nb && tid < CB) S[tid][tid] = 1.0f;  // a short last block is padded with the identity
    __syncthreads();
    for (int kb = 0; kb < 4; ++kb) {
        if (wv == 0) {
            // Lanes 0..31: row r of the block, factored right-looking.  Lanes 32..63: column r of the block's INVERSE, by the
            // right-looking form of the forward substitution -- s[c2] -= L[c2][c] x[c] -- whose multipliers L[c2][c] are exactly
            // the broadcasts the factor's own update a[c2] -= L[c2][c] L[r][c] needs: one v_readlane + one fma per (c, c2) serve
            // both halves, and the inverse costs no instruction of its own (it was a second 496-broadcast chain behind the factor).
            const int r = lane & 31;
            const bool inv = lane >= 32;
            float z[32];
#pragma unroll
            for (int c = 0; c < 32; ++c) z[c] = inv ? (r == c ? 1.0f : 0.0f) : S[32 * kb + r][32 * kb + c];
            int bad = 0;
#pragma unroll
            for (int c = 0; c < 32; ++c) {
                const float d = rl(z[c], c);  // the pivot: lane c of the factor half
                if (!(d > 0.0f) && bad == 0) bad = 32 * kb + c + 1;
                // reciprocal square root + one Newton step (full fp32 accuracy) instead of an IEEE sqrt and 32 IEEE
                // divisions per column: the factor does not need correctly rounded pivots, only accurate ones, and
                // these two sequences were two thirds of the instruction count of this loop
                float y = __builtin_amdgcn_rsqf(d);  // NaN for a negative pivot: the factor is visibly unusable
                y = y * fmaf(-0.5f * d * y, y, 1.5f);
                const float l = (!inv && r == c) ? d * y : z[c] * y;  // factor: L[r][c]; inverse: x[c] = s[c] / L[c][c]
                z[c] = l;
#pragma unroll
                for (int c2 = c + 1; c2 < 32; ++c2) z[c2] = fmaf(-l, rl(l, c2), z[c2]);  // rl(l, c2) = L[c2][c], from the factor half
            }
            if (bad && lane == 0) atomicCAS(info, 0, j + bad);
#pragma unroll
            for (int c = 0; c < 32; ++c) {
                if (!inv) S[32 * kb + r][32 * kb + c] = (c <= r) ? z[c] : 0.0f;
                else Xd[(kb * 32 + c) * XP + r] = z[c];  // X[row c][col r]
            }
        }
        __syncthreads();
        if (kb + 1 + wv <= 3) {  // panel: S[i][kb] <- S[i][kb] X_kb^T
            const int i = kb + 1 + wv;
            f32x16 acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
            mm32(acc, &S[32 * i][32 * kb], CP, Xd + kb * 32 * XP, XP, false, lane);
            wsync();
            acc_store(acc, &S[32 * i][32 * kb], CP, lane);
        }
        __syncthreads();
        int p = 0;
        for (int i = kb + 1; i < 4; ++i)
            for (int jb = kb + 1; jb <= i; ++jb, ++p) {
                if ((p & 3) != wv) continue;
                f32x16 acc;
                acc_load(acc, &S[32 * i][32 * jb], CP, lane);
                mm32(acc, &S[32 * i][32 * kb], CP, &S[32 * jb][32 * kb], CP, true, lane);
                acc_store(acc, &S[32 * i][32 * jb], CP, lane);
            }
        __syncthreads();
    }
    for (int i = tid; i < CB * CB; i += 256) {
        const int r = i / CB, c = i % CB;
        if (r < nb && c < nb) A[(int64_t)(j + r) * lda + j + c] = (c <= r) ? S[r][c] : 0.0f;
    }
    for (int i = tid; i < 4 * 32 * 32; i += 256) Xout[i] = Xd[(i >> 5) * XP + (i & 31)];
}

// Panel: rows R0 + 128*blockIdx.x ..: L21 = A21 L11^-T, block-wise forward substitution with the 32x32 inverses:
//   L21[:, kb] = (A21[:, kb] - sum_{k < kb} L21[:, k] L11[kb][k]^T) X_kb^T.  Every wave owns 32 rows; no block barriers.
__global__ __launch_bounds__(256) void chol_panel_kernel(float* __restrict__ A, int64_t lda, int n, int j, int nb,
                                                         const float* __restrict__ Xin) {
    extern __shared__ __align__(16) float sm[];
    float(*Ab)[CP] = reinterpret_cast<float(*)[CP]>(sm);
    float(*L11)[CP] = reinterpret_cast<float(*)[CP]>(sm + CB * CP);
    float* Xd = sm + 2 * CB * CP;  // [4][32][XP]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int r0 = j + nb + CB * blockIdx.x;
    load_block(Ab, A, lda, r0, min(CB, n - r0), j, nb, tid, false);
    load_block(L11, A, lda, j, nb, j, nb, tid, true);
    for (int i = tid; i < 4 * 32 * 32; i += 256) Xd[(i >> 5) * XP + (i & 31)] = Xin[i];
    __syncthreads();
    for (int kb = 0; kb < 4; ++kb) {
        f32x16 acc;
        acc_load(acc, &Ab[32 * wv][32 * kb], CP, lane);
        for (int k = 0; k < kb; ++k) mm32(acc, &Ab[32 * wv][32 * k], CP, &L11[32 * kb][32 * k], CP, true, lane);
        wsync();
        acc_store(acc, &Ab[32 * wv][32 * kb], CP, lane);
        wsync();
        f32x16 out;
#pragma unroll
        for (int e = 0; e < 16; ++e) out[e] = 0.0f;
        mm32(out, &Ab[32 * wv][32 * kb], CP, Xd + kb * 32 * XP, XP, false, lane);
        wsync();
        acc_store(out, &Ab[32 * wv][32 * kb], CP, lane);
        wsync();
    }
    __syncthreads();
    for (int i = tid; i < CB * CB; i += 256) {
        const int r = i / CB, c = i % CB;
        if (r0 + r < n && c < nb) A[(int64_t)(r0 + r) * lda + j + c] = Ab[r][c];
    }
}

// Update: tile (bi >= bj) of the trailing matrix at R0 = j + nb: A22[bi][bj] -= L21[bi] L21[bj]^T, K = nb <= 128 held
// entirely in LDS, 4 waves x (64x64).
// part 0: every tile; part 1: only the first block column (what the next diagonal block and panel read); part 2: the
// tiles right of it; part 3: the first block column below its diagonal tile -- the look-ahead driver runs parts 3 and 2 on a second
// stream next to the next step's diagonal / panel kernels.
__global__ __launch_bounds__(256) void chol_update_kernel(float* __restrict__ A, int64_t lda, int n, int j, int nb, int part) {
    extern __shared__ __align__(16) float sm[];
    float(*As)[CP] = reinterpret_cast<float(*)[CP]>(sm);
    float(*Bs)[CP] = reinterpret_cast<float(*)[CP]>(sm + CB * CP);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int R0 = j + nb;  // first row / column of the trailing part
    // linear index -> (bi, bj) with bj <= bi
    const int t = blockIdx.x;
    int bi, bj;
    if (part == 1 || part == 3) {  // 3: the first block column WITHOUT its diagonal tile (the next diagonal kernel applies that one itself)
        bi = t + (part == 3 ? 1 : 0);
        bj = 0;
    } else {
        bi = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
        while ((bi + 1) * (bi + 2) / 2 <= t) ++bi;
        while (bi * (bi + 1) / 2 > t) --bi;
        bj = t - bi * (bi + 1) / 2;
        if (part == 2) {  // the triangle without its first column
            ++bi;
            ++bj;
        }
    }
    const int ra0 = R0 + CB * bi, rb0 = R0 + CB * bj;
    // stage the operands: 128 rows x 128 k each (k beyond nb and rows beyond n are zero)
    load_block(As, A, lda, ra0, max(0, min(CB, n - ra0)), j, nb, tid, false);
    load_block(Bs, A, lda, rb0, max(0, min(CB, n - rb0)), j, nb, tid, false);
    __syncthreads();
    const int wm = (wv >> 1) * 64, wn = (wv & 1) * 64;
    const int i32 = lane & 31, kk = lane >> 5;
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.0f;
    for (int k = 0; k < CB; k += 2) {
        float av[2], bv[2];
#pragma unroll
        for (int a = 0; a < 2; ++a) av[a] = As[wm + 32 * a + i32][k + kk];
#pragma unroll
        for (int b = 0; b < 2; ++b) bv[b] = Bs[wn + 32 * b + i32][k + kk];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a], bv[b], acc[a][b], 0, 0, 0);
    }
    // C layout of the 32x32 MFMA: column = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
    // Read-modify-write of the tile: ALL 64 old values of a lane are requested before the first store (as a loop of `*p -= acc`
    // the compiler keeps every load behind the previous store -- it cannot know they do not alias -- and the tile's end was 64
    // round trips to L2: 33 us for a launch of one round of tiles whose matrix work is 7 us).
    float oldv[2][2][16];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = ra0 + wm + 32 * a + (e & 3) + 8 * (e >> 2) + 4 * kk;
                const int col = rb0 + wn + 32 * b + i32;
                const bool ok = row < n && col < n && col <= row;
                oldv[a][b][e] = ok ? __builtin_nontemporal_load(A + (int64_t)row * lda + col) : 0.0f;
            }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = ra0 + wm + 32 * a + (e & 3) + 8 * (e >> 2) + 4 * kk;
                const int col = rb0 + wn + 32 * b + i32;
                if (row < n && col < n && col <= row) A[(int64_t)row * lda + col] = oldv[a][b][e] - acc[a][b][e];
            }
}

__global__ __launch_bounds__(256) void chol_zero_upper_kernel(float* __restrict__ A, int64_t lda, int n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)n * n) return;
    const int r = (int)(i / n), c = (int)(i % n);
    if (c > r) A[(int64_t)r * lda + c] = 0.0f;
}

}  // namespace ganq

using namespace ganq;

extern "C" size_t ganq_cholesky_workspace_bytes(int64_t n) {
    if (n <= 0) return 0;
    return align_up((size_t)CB * CB * sizeof(float), 256) + 256;
}

namespace {
struct Lookahead {
    hipStream_t side;
    hipEvent_t panel_done, col_done, rest_done[2];
};
// one helper stream + two events per (device, caller stream), created on first use: two factorisations may run at
// the same time on different streams (the prologue does that).  GANQ_CHOL_LOOKAHEAD=0 turns the second stream off.
Lookahead* lookahead_for(hipStream_t main) {
    struct Slot {
        int dev;
        hipStream_t main;
        Lookahead la;
    };
    static Slot slots[32];
    static int used = 0;
    static std::mutex mu;
    const bool off = opt_get(OPT_CHOL_LOOKAHEAD) == 0;
    int dev = 0;
    if (off || hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    for (int i = 0; i < used; ++i)
        if (slots[i].dev == dev && slots[i].main == main) return &slots[i].la;
    if (used == 32) return nullptr;  // more caller streams than slots: plain single-stream factorisation
    Lookahead la{};
    if (hipStreamCreateWithFlags(&la.side, hipStreamNonBlocking) != hipSuccess) return nullptr;
    if (hipEventCreateWithFlags(&la.panel_done, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&la.col_done, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&la.rest_done[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&la.rest_done[1], hipEventDisableTiming) != hipSuccess)
        return nullptr;
    slots[used] = Slot{dev, main, la};
    return &slots[used++].la;
}
}  // namespace

extern "C" int ganq_cholesky(float* A, int64_t n, int64_t lda, int32_t* info_out, void* workspace, size_t workspace_bytes,
                             void* stream_) {
    if (n < 0) return fail(-1, "ganq_cholesky: negative n");
    if (n == 0) return 0;
    if (lda < n) return fail(-1, "ganq_cholesky: lda=%lld < n=%lld", (long long)lda, (long long)n);
    if (n > INT32_MAX / 2) return fail(-1, "ganq_cholesky: n too large");
    if (!A || !info_out) return fail(-3, "ganq_cholesky: null pointer");
    const size_t need = ganq_cholesky_workspace_bytes(n);
    if (!workspace || workspace_bytes < need)
        return fail(-4, "ganq_cholesky: workspace %zu B < required %zu B", workspace_bytes, need);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    float* X = static_cast<float*>(workspace);
    const size_t smem = 2 * (size_t)CB * CP * sizeof(float);
    const size_t smem_diag = ((size_t)CB * CP + 4 * 32 * XP) * sizeof(float);
    const size_t smem_panel = (2 * (size_t)CB * CP + 4 * 32 * XP) * sizeof(float);
    {
        int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(chol_diag_kernel), smem_diag);
        if (!rc) rc = ensure_dynamic_lds(reinterpret_cast<const void*>(chol_panel_kernel), smem_panel);
        if (!rc) rc = ensure_dynamic_lds(reinterpret_cast<const void*>(chol_update_kernel), smem);
        if (rc) return rc;
    }
    ProfScope prof(KID_CHOLESKY, stream);
    GANQ_HIP_CHECK(hipMemsetAsync(info_out, 0, sizeof(int32_t), stream));
    // Look-ahead (n >= 4096): the chain diagonal -> panel -> diagonal -> ... runs on the caller's stream and touches the trailing
    // matrix only through the panel: the diagonal kernel of step s + 1 applies step s's update to its own block itself (from the first
    // panel block), the panel kernel of step s + 1 waits for the first block column of step s's update, and the whole update of step
    // s -- first column, then the rest -- runs on a second stream beside them.  Round 3 kept the first block column on the chain
    // (diag 64 + panel 21 + column 26 us per step and the launch gaps between them: 4.05 ms at n = 4096).
    Lookahead* la = lookahead_for(stream);
    const bool two = la != nullptr && n >= 32 * CB;
    if (two) {
        const int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(chol_diag_kernel), smem_diag + (size_t)CB * CP * sizeof(float));
        if (rc) return rc;
    }
    int step = 0;
    for (int64_t j = 0; j < n; j += CB, ++step) {
        const int nb = (int)std::min<int64_t>(CB, n - j);
        const int64_t rem = n - j - nb;
        const int nblk = (int)((rem + CB - 1) / CB);
        if (!two) {
            hipLaunchKernelGGL(chol_diag_kernel, dim3(1), dim3(256), smem_diag, stream, A, lda, (int)j, nb, -1, X, info_out);
            if (rem > 0) {
                hipLaunchKernelGGL(chol_panel_kernel, dim3(nblk), dim3(256), smem_panel, stream, A, lda, (int)n, (int)j, nb, X);
                hipLaunchKernelGGL(chol_update_kernel, dim3(nblk * (nblk + 1) / 2), dim3(256), smem, stream, A, lda, (int)n, (int)j, nb, 0);
            }
            continue;
        }
        // this step's diagonal block was last written by the rest-update of step - 2 (second stream)
        if (step >= 2) GANQ_HIP_CHECK(hipStreamWaitEvent(stream, la->rest_done[step & 1], 0));
        hipLaunchKernelGGL(chol_diag_kernel, dim3(1), dim3(256), smem_diag + (step > 0 ? (size_t)CB * CP * sizeof(float) : 0), stream, A, lda, (int)j, nb,
                           step > 0 ? (int)(j - CB) : -1, X, info_out);
        if (rem > 0) {
            // the panel's rows were last written by the first block column of the previous step's update (second stream)
            if (step >= 1) GANQ_HIP_CHECK(hipStreamWaitEvent(stream, la->col_done, 0));
            hipLaunchKernelGGL(chol_panel_kernel, dim3(nblk), dim3(256), smem_panel, stream, A, lda, (int)n, (int)j, nb, X);
            GANQ_HIP_CHECK(hipEventRecord(la->panel_done, stream));
            GANQ_HIP_CHECK(hipStreamWaitEvent(la->side, la->panel_done, 0));
            if (nblk > 1)
                hipLaunchKernelGGL(chol_update_kernel, dim3(nblk - 1), dim3(256), smem, la->side, A, lda, (int)n, (int)j, nb, 3);
            GANQ_HIP_CHECK(hipEventRecord(la->col_done, la->side));
            if (nblk > 1)
                hipLaunchKernelGGL(chol_update_kernel, dim3((nblk - 1) * nblk / 2), dim3(256), smem, la->side, A, lda, (int)n, (int)j, nb, 2);
            GANQ_HIP_CHECK(hipEventRecord(la->rest_done[step & 1], la->side));
        }
    }
    if (two) {  // (the last steps' updates: both parities)
        GANQ_HIP_CHECK(hipStreamWaitEvent(stream, la->rest_done[0], 0));
        GANQ_HIP_CHECK(hipStreamWaitEvent(stream, la->rest_done[1], 0));
        GANQ_HIP_CHECK(hipStreamWaitEvent(stream, la->col_done, 0));
    }
    hipLaunchKernelGGL(chol_zero_upper_kernel, dim3((unsigned)((n * n + 255) / 256)), dim3(256), 0, stream, A, lda, (int)n);
    GANQ_LAUNCH_CHECK();
    return 0;
}

// T-update: closed-form codebook update of GANQ (reference ganq.py:570-591, CPU/gelsd branch)
//     A_i = S_i H S_i^T,  b_i = S_i (W H)_i^T,  T_i = minimum-norm lstsq(A_i, b_i)
// without ever materialising the one-hot tensor S [m,V,n] (1 GiB at 4096^2, ganq.py:505).
//
// H is symmetric: A_i = M_i + M_i^T + diag_a( sum_{u in a} H[u,u] ) with
//     M_i[a][b] = sum_{u > v} [Q_iu == a][Q_iv == b] H[u,v]      (strict lower triangle only).
//
// Kernels
//   sort_codes_kernel   per (row, 256-row tile of H): counting sort of the tile's columns u by code
//                       -> LDS byte offsets of the tile rows grouped by code (ascending u inside a code),
//                       every code segment padded to a multiple of 8 with the offset of an all-zero row
//   sht_accum_kernel    workgroup = (128-wide v chunk c, 32 rows of W).  Tiles of H [256 u x 128 v]
//                       that reach below the diagonal are staged through LDS (entries with u <= v zeroed);
//                       each wave owns 2 rows and keeps Yl[row][code][2 v per lane] in registers:
//                       for code a: for u in tile with Q_iu == a: Yl[a] += H[u, chunk]   (ds_read_b64 + v_pk_add_f32)
//                       The H tile is shared by 32 rows, the accumulator index is static (code-major
//                       loops over the sorted lists, 8 independent LDS reads per batch).  Epilogue:
//                       M_c = Yl @ onehot(Q_i[chunk])^T on the fp32 matrix cores (ordered, deterministic).
//   solve_kernel        16 lanes per row: reduce the chunks in fixed order, add M^T and the diagonal term,
//                       build b, eigen-decompose the 16x16 system with round-robin Jacobi in fp64 and
//                       form the minimum-norm solution with the gelsd cut-off (rcond * |lambda|_max).
#include "common.h"

namespace ganq {

constexpr int UT = 256;                 // rows of H per tile
constexpr int VC = 128;                 // columns of H per chunk (2 per lane)
constexpr int TW = 16;                  // waves per workgroup in sht_accum
constexpr int RW = 2;                   // rows of W per wave
constexpr int TR = TW * RW;             // rows of W per workgroup (32)
constexpr int NB = 8;                   // sorted-list batch: every code segment is padded to a multiple of NB
constexpr int LIST = NB * 64;           // sorted-list slots per (row, tile): [NB][64] uint32, entry e at [e % NB][e / NB]
constexpr uint32_t ZERO_OFF = UT * VC * 4;  // LDS byte offset of the all-zero row that padding entries point to
constexpr size_t ACCUM_TILE_BYTES = (size_t)(UT + 1) * VC * sizeof(float);
constexpr size_t ACCUM_EPI_BYTES = (size_t)TW * 16 * (VC + 4) * sizeof(float);
constexpr size_t ACCUM_SMEM = ACCUM_TILE_BYTES > ACCUM_EPI_BYTES ? ACCUM_TILE_BYTES : ACCUM_EPI_BYTES;

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sort_codes_kernel(const uint8_t* __restrict__ Q, int m, int n, int ntile,
                                                         uint32_t* __restrict__ sorted_off,
                                                         uint8_t* __restrict__ seg_start) {
    // one wave per (row, tile): counting sort of the tile's 256 columns by code, ascending column inside a code
    const int lane = threadIdx.x & 63;
    const int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= (int64_t)m * ntile) return;
    const int row = (int)(item / ntile), t = (int)(item % ntile);
    const int u0 = t * UT;
    const uint8_t* q = Q + (int64_t)row * n;
    uint32_t qv[UT / 64];
#pragma unroll
    for (int j = 0; j < UT / 64; ++j) {
        const int u = u0 + 64 * j + lane;
        qv[j] = u < n ? q[u] : 255u;
    }
    uint32_t* out = sorted_off + item * LIST;
    uint8_t* seg = seg_start + item * 32;
    const uint64_t lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    uint32_t start = 0;  // in entries, always a multiple of NB
#pragma unroll
    for (uint32_t a = 0; a < 16; ++a) {
        uint32_t pos = start;
#pragma unroll
        for (int j = 0; j < UT / 64; ++j) {
            const uint64_t mk = __ballot(qv[j] == a);
            if (qv[j] == a) {
                const uint32_t e = pos + __popcll(mk & lt);
                out[(e % NB) * 64 + e / NB] = (uint32_t)(64 * j + lane) * (VC * 4);
            }
            pos += __popcll(mk);
        }
        if (lane == 0) seg[a] = (uint8_t)(start / NB);
        start = (pos + NB - 1) / NB * NB;
        if ((uint32_t)lane < start - pos) {  // pad the segment's last batch with the all-zero row
            const uint32_t e = pos + lane;
            out[(e % NB) * 64 + e / NB] = ZERO_OFF;
        }
    }
    if (lane == 0) seg[16] = (uint8_t)(start / NB);  // <= (256 + 16*7)/8 = 46
}

// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void accum_batches(float2& acc, const char* lds_lane, const uint32_t (&off)[NB], int s,
                                              int e) {
    for (int b = s; b < e; ++b) {
        float2 h[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const uint32_t o = (uint32_t)__builtin_amdgcn_readlane((int)off[j], b);
            h[j] = *reinterpret_cast<const float2*>(lds_lane + o);
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {  // ascending column order
            acc.x += h[j].x;
            acc.y += h[j].y;
        }
    }
}

__global__ __launch_bounds__(TW * 64) void sht_accum_kernel(const float* __restrict__ H, const uint8_t* __restrict__ Q,
                                                           const uint32_t* __restrict__ sorted_off,
                                                           const uint8_t* __restrict__ seg_start, int m, int n,
                                                           int ntile, float* __restrict__ Mws, int kasc) {
    extern __shared__ __align__(16) char smem[];  // [UT + 1][VC] floats (last row = zeros)
    float(*Ht)[VC] = reinterpret_cast<float(*)[VC]>(smem);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = tid >> 6;
    const int c = blockIdx.x;               // v chunk
    const int rg = blockIdx.y;              // row group
    const int v0 = c * VC;
    const int row_base = rg * TR + wv * RW;  // this wave's rows
    const int t_first = (c * VC) / UT;       // first tile that reaches below the chunk's diagonal

    float2 acc[RW][16];
#pragma unroll
    for (int r = 0; r < RW; ++r)
#pragma unroll
        for (int a = 0; a < 16; ++a) acc[r][a] = make_float2(0.f, 0.f);

    // global -> register -> LDS staging of one H tile: 256 x 128 floats = 8192 float4, 8 per thread
    constexpr int NST = (UT * VC / 4) / (TW * 64);
    float4 stage[NST];
    auto gload = [&](int t) {
#pragma unroll
        for (int e = 0; e < NST; ++e) {
            const int idx = e * (TW * 64) + tid;
            const int ul = idx >> 5, v4 = (idx & 31) * 4;
            const int u = t * UT + ul, v = v0 + v4;
            float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
            if (u < n) {
                const float* p = H + (int64_t)u * n + v;
                if (v + 3 < n && ((reinterpret_cast<uintptr_t>(p) & 15) == 0)) {
                    x = *reinterpret_cast<const float4*>(p);
                } else {
                    if (v + 0 < n) x.x = p[0];
                    if (v + 1 < n) x.y = p[1];
                    if (v + 2 < n) x.z = p[2];
                    if (v + 3 < n) x.w = p[3];
                }
                // keep the strict lower triangle u > v only (matters on the tile that crosses the diagonal)
                if (!(u > v + 0)) x.x = 0.f;
                if (!(u > v + 1)) x.y = 0.f;
                if (!(u > v + 2)) x.z = 0.f;
                if (!(u > v + 3)) x.w = 0.f;
            }
            stage[e] = x;
        }
    };
    auto sstore = [&]() {
#pragma unroll
        for (int e = 0; e < NST; ++e) {
            const int idx = e * (TW * 64) + tid;
            const int ul = idx >> 5, v4 = (idx & 31) * 4;
            *reinterpret_cast<float4*>(&Ht[ul][v4]) = stage[e];
        }
    };

    if (tid < VC / 4) *reinterpret_cast<float4*>(&Ht[UT][tid * 4]) = make_float4(0.f, 0.f, 0.f, 0.f);
    gload(t_first);
    sstore();
    __syncthreads();
    const char* lds_lane = reinterpret_cast<const char*>(&Ht[0][0]) + lane * 8;
    for (int t = t_first; t < ntile; ++t) {
        if (t + 1 < ntile) gload(t + 1);
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            const int row = min(row_base + r, m - 1);
            const int64_t item = (int64_t)row * ntile + t;
            uint32_t off[NB];
#pragma unroll
            for (int j = 0; j < NB; ++j) off[j] = sorted_off[item * LIST + j * 64 + lane];
            // segment starts (in batches): 17 bytes, uniform
            const uint4 s4 = *reinterpret_cast<const uint4*>(seg_start + item * 32);
            const uint32_t s16 = seg_start[item * 32 + 16];
            const uint32_t w[4] = {(uint32_t)__builtin_amdgcn_readfirstlane((int)s4.x),
                                   (uint32_t)__builtin_amdgcn_readfirstlane((int)s4.y),
                                   (uint32_t)__builtin_amdgcn_readfirstlane((int)s4.z),
                                   (uint32_t)__builtin_amdgcn_readfirstlane((int)s4.w)};
            int st[17];
#pragma unroll
            for (int a = 0; a < 16; ++a) st[a] = (int)((w[a >> 2] >> (8 * (a & 3))) & 0xffu);
            st[16] = (int)__builtin_amdgcn_readfirstlane((int)s16);
#pragma unroll
            for (int a = 0; a < 16; ++a) accum_batches(acc[r][a], lds_lane, off, st[a], st[a + 1]);
        }
        __syncthreads();  // every wave is done reading the tile
        if (t + 1 < ntile) {
            sstore();
            __syncthreads();
        }
    }

    // ---- epilogue: M_c[row] = Yl[16 codes x 128 v] @ onehot(Q[row, chunk])^T  (16x16x4 fp32 MFMA) ----
    // per-wave scratch Ys[16][VC + 4] in LDS (the H tile is free after the last barrier)
    float(*Ys)[VC + 4] = reinterpret_cast<float(*)[VC + 4]>(smem + (size_t)wv * 16 * (VC + 4) * sizeof(float));
    const int ksub = lane >> 4, c16 = lane & 15;
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        const int row = row_base + r;
        const int rowc = min(row, m - 1);
#pragma unroll
        for (int a = 0; a < 16; ++a) *reinterpret_cast<float2*>(&Ys[a][2 * lane]) = acc[r][a];
        __builtin_amdgcn_wave_barrier();
        f32x4 d = {0.f, 0.f, 0.f, 0.f};
        for (int g = 0; g < VC / 4; ++g) {
            const int vl = kasc ? (4 * g + ksub) : (4 * g + 3 - ksub);  // accumulate v ascending
            const int v = v0 + vl;
            const float av = Ys[c16][vl];
            const uint32_t qq = (v < n) ? Q[(int64_t)rowc * n + v] : 255u;
            const float bv = (qq == (uint32_t)c16) ? 1.0f : 0.0f;
            d = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, d, 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();
        if (row < m) {
            float* out = Mws + ((int64_t)c * m + row) * 256;
#pragma unroll
            for (int i = 0; i < 4; ++i) out[(ksub * 4 + i) * 16 + c16] = d[i];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// 16 lanes per row, 4 rows per wave, 1 wave per workgroup.
constexpr int JS = 17;  // padded leading dimension of the fp64 16x16 matrices in LDS

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

__global__ __launch_bounds__(64) void solve_kernel(const float* __restrict__ Mws, int nchunk, const float* __restrict__ H,
                                                   const float* __restrict__ WH, const uint8_t* __restrict__ Q, int m,
                                                   int n, int V, double rcond, float* __restrict__ T_out,
                                                   float* __restrict__ A_out, float* __restrict__ b_out) {
    __shared__ double As[4][16][JS];
    __shared__ double Es[4][16][JS];
    __shared__ double CS[4][8][2];
    __shared__ double Coef[4][16];

    const int lane = threadIdx.x & 63;
    const int rs = lane >> 4, l = lane & 15;
    const int row = blockIdx.x * 4 + rs;
    const int rowc = min(row, m - 1);
    double(*A)[JS] = As[rs];
    double(*E)[JS] = Es[rs];

    // ---- assemble: lane l owns column l of M (summed over chunks in ascending order) ----
    {
        float col[16];
#pragma unroll
        for (int a = 0; a < 16; ++a) col[a] = 0.f;
        for (int c = 0; c < nchunk; ++c) {
            const float* src = Mws + ((int64_t)c * m + rowc) * 256;
#pragma unroll
            for (int a = 0; a < 16; ++a) col[a] += src[a * 16 + l];
        }
#pragma unroll
        for (int a = 0; a < 16; ++a) A[a][l] = (double)col[a];
    }
    // lane l == code l: diagonal term and right-hand side, ascending u
    double dsum = 0.0, bsum = 0.0;
    {
        const uint8_t* q = Q + (int64_t)rowc * n;
        const float* wh = WH + (int64_t)rowc * n;
        for (int u = 0; u < n; ++u) {
            const bool hit = (q[u] == (uint8_t)l);
            const double hd = (double)H[(int64_t)u * n + u];
            const double wv = (double)wh[u];
            dsum += hit ? hd : 0.0;
            bsum += hit ? wv : 0.0;
        }
    }
    __builtin_amdgcn_wave_barrier();
    // A = M + M^T + diag, rounded to fp32 (the reference holds A and b in fp32), exactly symmetric
    double colA[16];
#pragma unroll
    for (int a = 0; a < 16; ++a) {
        double x = A[a][l] + A[l][a];
        if (a == l) x += dsum;
        colA[a] = (double)(float)x;
    }
    const double bl = (double)(float)bsum;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int a = 0; a < 16; ++a) {
        A[a][l] = colA[a];
        E[a][l] = (a == l) ? 1.0 : 0.0;
    }
    if (row < m) {
        if (A_out && l < V)
            for (int a = 0; a < V; ++a) A_out[((int64_t)row * V + a) * V + l] = (float)colA[a];
        if (b_out && l < V) b_out[(int64_t)row * V + l] = (float)bl;
    }
    __builtin_amdgcn_wave_barrier();

    // ---- round-robin Jacobi: 15 rounds of 8 disjoint rotations per sweep ----
    const int t = l >> 1;  // pair handled (redundantly) by lanes 2t, 2t+1
    for (int sweep = 0; sweep < 30; ++sweep) {
        double off = 0.0, dg = 0.0;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const double x = A[l][j];
            if (j == l) dg += x * x; else off += x * x;
        }
        off = row16_sum(off);
        dg = row16_sum(dg);
        const bool done = (off <= 1e-30 * dg) || (off == 0.0);
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
                const double theta = (aqq - app) / (2.0 * apq);
                const double tt = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                cc = 1.0 / sqrt(tt * tt + 1.0);
                ss = tt * cc;
            }
            if ((l & 1) == 0) {
                CS[rs][t][0] = cc;
                CS[rs][t][1] = ss;
            }
            __builtin_amdgcn_wave_barrier();
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
            __builtin_amdgcn_wave_barrier();
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
            __builtin_amdgcn_wave_barrier();
        }
    }

    // ---- minimum-norm solution: x = sum_k [|lam_k| > rcond * lam_max] (e_k . b / lam_k) e_k ----
    const double lam = A[l][l];
    const double lmax = row16_max(fabs(lam));
    Coef[rs][l] = bl;
    __builtin_amdgcn_wave_barrier();
    double proj = 0.0;
#pragma unroll
    for (int a = 0; a < 16; ++a) proj += E[a][l] * Coef[rs][a];
    __builtin_amdgcn_wave_barrier();
    const bool keep = fabs(lam) > rcond * lmax;
    Coef[rs][l] = keep ? proj / lam : 0.0;
    __builtin_amdgcn_wave_barrier();
    double x = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) x += E[l][k] * Coef[rs][k];
    if (row < m && l < V) T_out[(int64_t)row * V + l] = (float)x;
}

}  // namespace ganq

using namespace ganq;

struct UpdateTLayout {
    int64_t ntile, nchunk;
    size_t off_sorted, off_seg, off_mws, total;
};

static UpdateTLayout update_t_layout(int64_t m, int64_t n) {
    UpdateTLayout lo;
    lo.ntile = (n + UT - 1) / UT;
    lo.nchunk = (n + VC - 1) / VC;
    size_t off = 0;
    lo.off_sorted = off;
    off = align_up(off + (size_t)m * (size_t)lo.ntile * LIST * sizeof(uint32_t), 256);
    lo.off_seg = off;
    off = align_up(off + (size_t)m * (size_t)lo.ntile * 32, 256);
    lo.off_mws = off;
    off = align_up(off + (size_t)lo.nchunk * (size_t)m * 256 * sizeof(float), 256);
    lo.total = off;
    return lo;
}

extern "C" size_t ganq_update_t_workspace_bytes(int64_t m, int64_t n, int V) {
    (void)V;
    if (m <= 0 || n <= 0) return 0;
    return update_t_layout(m, n).total;
}

extern "C" int ganq_update_t(const float* WH, const float* H, const uint8_t* Q, int64_t m, int64_t n, int V,
                             double rcond, float* T_out, float* A_out, float* b_out, void* workspace,
                             size_t workspace_bytes, void* stream_) {
    if (m < 0 || n < 0) return fail(-1, "ganq_update_t: negative shape");
    if (m == 0 || n == 0) return 0;
    if (V < 2 || V > 16) return fail(-2, "ganq_update_t: V=%d not supported (bits 2..4 are implemented)", V);
    if (m > INT32_MAX / 2 || n > INT32_MAX / 2) return fail(-1, "ganq_update_t: shape too large");
    if (!WH || !H || !Q || !T_out) return fail(-3, "ganq_update_t: null pointer");
    const UpdateTLayout lo = update_t_layout(m, n);
    const int64_t ntile = lo.ntile, nchunk = lo.nchunk;
    if (!workspace || workspace_bytes < lo.total)
        return fail(-4, "ganq_update_t: workspace %zu B < required %zu B", workspace_bytes, lo.total);
    if (rcond < 0) rcond = 1.1920928955078125e-07 * (double)V;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    int rc = ganq_hip_selftest(stream_);
    if (rc) return rc;
    char* ws = static_cast<char*>(workspace);
    uint32_t* sorted_off = reinterpret_cast<uint32_t*>(ws + lo.off_sorted);
    uint8_t* seg = reinterpret_cast<uint8_t*>(ws + lo.off_seg);
    float* mws = reinterpret_cast<float*>(ws + lo.off_mws);

    const int64_t items = m * ntile;
    {
        ProfScope prof(KID_SORT_CODES, stream);
        hipLaunchKernelGGL(sort_codes_kernel, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, stream, Q, (int)m, (int)n,
                           (int)ntile, sorted_off, seg);
    }
    GANQ_LAUNCH_CHECK();

    static bool attr_set = false;
    const size_t smem = ACCUM_SMEM;
    if (!attr_set) {
        GANQ_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(sht_accum_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr_set = true;
    }
    const dim3 grid((unsigned)nchunk, (unsigned)((m + TR - 1) / TR));
    {
        ProfScope prof(KID_SHT_ACCUM, stream);
        hipLaunchKernelGGL(sht_accum_kernel, grid, dim3(TW * 64), smem, stream, H, Q, sorted_off, seg, (int)m, (int)n,
                           (int)ntile, mws, mfma_k_ascending());
    }
    GANQ_LAUNCH_CHECK();
    {
        ProfScope prof(KID_T_SOLVE, stream);
        hipLaunchKernelGGL(solve_kernel, dim3((unsigned)((m + 3) / 4)), dim3(64), 0, stream, mws, (int)nchunk, H, WH, Q,
                           (int)m, (int)n, V, rcond, T_out, A_out, b_out);
    }
    GANQ_LAUNCH_CHECK();
    return 0;
}

// Codebook initialisation: optimal weighted 1-D k-means per row (reference ganq.py:423-438 + kmeans_fit :27-30,
// which calls the un-vendored kmeans1d package; algorithm restated in oracle/ganq_oracle.c ganq_oracle_kmeans_row).
//
// One workgroup per row (persistent over rows):
//   1. bitonic sort of (value, column) keys in LDS;
//   2. weighted prefix sums cw, cwx, cwxx in fp64 (two-level: chunks of 16 summed left to right, chunk totals
//      scanned left to right -- the same association order as the oracle);
//   3. dynamic programme  D[k][i] = min_j D[k-1][j-1] + cost(j..i),  cost = sum w x^2 - (sum w x)^2 / sum w.
//      The argmin is monotone in i, so each layer is solved position-by-position in "bit-reversed" levels:
//      i = n-1 first (full scan by the whole workgroup), then i = hs-1 + t*2hs for hs = P/2, P/4, .., 1, each
//      bounded by the argmins of its two already-solved neighbours i-hs and i+hs; nodes of a level are independent
//      (groups of up to 64 lanes scan one node and reduce with leftmost-minimum tie-break);
//   4. backtrack, centroids = weighted means (ascending).
#include "common.h"

namespace ganq {

struct KmPre {
    double cw, cwx, cwxx, dprev;  // prefix sums at index j, and D[k-1][j-1]
};

constexpr int KM_CHUNK = 16;

__device__ __forceinline__ double km_cost(const KmPre& pj, double cw_i1, double cwx_i1, double cwxx_i1) {
    const double w = cw_i1 - pj.cw;
    const double wx = cwx_i1 - pj.cwx;
    const double wxx = cwxx_i1 - pj.cwxx;
    if (!(w > 0.0)) return 0.0;
    const double c = wxx - (wx * wx) / w;
    return c > 0.0 ? c : 0.0;
}

__device__ __forceinline__ void km_better(double& bc, int& bj, double c, int j) {
    if (c < bc || (c == bc && j < bj)) {
        bc = c;
        bj = j;
    }
}

__global__ __launch_bounds__(256) void kmeans_kernel(const float* __restrict__ W, const double* __restrict__ col_weight,
                                                     int m, int n, int V, int P, float* __restrict__ T0,
                                                     char* __restrict__ ws, size_t ws_stride) {
    extern __shared__ __align__(16) char km_smem[];
    uint64_t* keys = reinterpret_cast<uint64_t*>(km_smem);  // [P]
    __shared__ double red_c[4];
    __shared__ int red_j[4];
    __shared__ double chunk_tot[3][1024];  // chunk totals (n <= 16384 -> <= 1024 chunks)

    const int tid = threadIdx.x;
    char* my = ws + (size_t)blockIdx.x * ws_stride;
    KmPre* pre = reinterpret_cast<KmPre*>(my);                                               // [n+1]
    double* dcur = reinterpret_cast<double*>(my + align_up((size_t)(n + 1) * sizeof(KmPre), 256));  // [n]
    int* arg = reinterpret_cast<int*>(reinterpret_cast<char*>(dcur) + align_up((size_t)n * sizeof(double), 256));  // [V][n]

    for (int row = blockIdx.x; row < m; row += gridDim.x) {
        // ---- 1. sort ------------------------------------------------------------------------------
        for (int i = tid; i < P; i += 256) {
            uint64_t key = ~0ull;
            if (i < n) {
                uint32_t b = __builtin_bit_cast(uint32_t, W[(int64_t)row * n + i]);
                b ^= (b >> 31) ? 0xffffffffu : 0x80000000u;
                key = ((uint64_t)b << 32) | (uint32_t)i;
            }
            keys[i] = key;
        }
        __syncthreads();
        for (int k = 2; k <= P; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int i = tid; i < P; i += 256) {
                    const int partner = i ^ j;
                    if (partner > i) {
                        const bool asc = (i & k) == 0;
                        const uint64_t a = keys[i], b = keys[partner];
                        if ((a > b) == asc) {
                            keys[i] = b;
                            keys[partner] = a;
                        }
                    }
                }
                __syncthreads();
            }
        }
        auto val_at = [&](int u) -> double {
            uint32_t b = (uint32_t)(keys[u] >> 32);
            b ^= (b >> 31) ? 0x80000000u : 0xffffffffu;
            return (double)__builtin_bit_cast(float, b);
        };
        auto wt_at = [&](int u) -> double { return col_weight ? col_weight[(uint32_t)keys[u]] : 1.0; };

        // ---- 2. prefix sums (chunks of 16, then chunk totals, both left to right) ----------------------
        const int nchunk = (n + KM_CHUNK - 1) / KM_CHUNK;
        for (int c = tid; c < nchunk; c += 256) {
            double a = 0.0, b = 0.0, d = 0.0;
            const int hi = min(n, (c + 1) * KM_CHUNK);
            for (int u = c * KM_CHUNK; u < hi; ++u) {
                const double x = val_at(u), w = wt_at(u);
                a += w;
                b += w * x;
                d += w * x * x;
            }
            chunk_tot[0][c] = a;
            chunk_tot[1][c] = b;
            chunk_tot[2][c] = d;
        }
        __syncthreads();
        if (tid < 3) {
            double run = 0.0;
            for (int c = 0; c < nchunk; ++c) {
                const double t = chunk_tot[tid][c];
                chunk_tot[tid][c] = run;  // exclusive
                run += t;
            }
        }
        __syncthreads();
        for (int c = tid; c < nchunk; c += 256) {
            double a = 0.0, b = 0.0, d = 0.0;
            const double oa = chunk_tot[0][c], ob = chunk_tot[1][c], od = chunk_tot[2][c];
            const int hi = min(n, (c + 1) * KM_CHUNK);
            for (int u = c * KM_CHUNK; u < hi; ++u) {
                pre[u].cw = oa + a;
                pre[u].cwx = ob + b;
                pre[u].cwxx = od + d;
                const double x = val_at(u), w = wt_at(u);
                a += w;
                b += w * x;
                d += w * x * x;
            }
            if (hi == n) {
                pre[n].cw = oa + a;
                pre[n].cwx = ob + b;
                pre[n].cwxx = od + d;
                pre[n].dprev = 0.0;
            }
        }
        __syncthreads();

        // ---- 3. DP ----------------------------------------------------------------------------------------
        // layer 0: D[0][i] = cost(0..i)
        for (int i = tid; i < n; i += 256) {
            const KmPre p0 = pre[0];
            const KmPre pi = pre[i + 1];
            dcur[i] = km_cost(p0, pi.cw, pi.cwx, pi.cwxx);
            arg[i] = 0;
        }
        __syncthreads();
        for (int k = 1; k < V; ++k) {
            // D[k-1] -> pre[j].dprev = D[k-1][j-1]
            for (int j = tid; j <= n; j += 256) pre[j].dprev = (j == 0) ? 0.0 : dcur[j - 1];
            __syncthreads();
            int* a = arg + (size_t)k * n;
            // position n-1: full scan by the whole workgroup
            {
                const int i = n - 1;
                const KmPre pi = pre[i + 1];
                double bc = INFINITY;
                int bj = 0x7fffffff;
                for (int j = tid; j <= i; j += 256) {
                    const KmPre pj = pre[j];
                    km_better(bc, bj, pj.dprev + km_cost(pj, pi.cw, pi.cwx, pi.cwxx), j);
                }
                for (int off = 32; off > 0; off >>= 1) {
                    const double oc = __shfl_xor(bc, off);
                    const int oj = __shfl_xor(bj, off);
                    km_better(bc, bj, oc, oj);
                }
                if ((tid & 63) == 0) {
                    red_c[tid >> 6] = bc;
                    red_j[tid >> 6] = bj;
                }
                __syncthreads();
                if (tid == 0) {
                    for (int w = 1; w < 4; ++w) km_better(bc, bj, red_c[w], red_j[w]);
                    dcur[i] = bc;
                    a[i] = bj;
                }
                __syncthreads();
            }
            if (k == V - 1) break;  // only D[V-1][n-1] is needed from the last layer
            for (int hs = P >> 1; hs >= 1; hs >>= 1) {
                // nodes i = hs-1 + t*2hs, i < n-1
                const int cnt = (n - 1 > hs - 1) ? ((n - 1 - (hs - 1) + 2 * hs - 1) / (2 * hs)) : 0;
                if (cnt == 0) continue;
                int G = 1;
                while (G < 64 && G * 2 * cnt <= 256) G <<= 1;
                const int groups = 256 / G;
                const int lg = tid & (G - 1);
                for (int t = tid / G; t < cnt; t += groups) {
                    const int i = hs - 1 + t * 2 * hs;
                    const int lo = (i - hs >= 0) ? a[i - hs] : 0;
                    const int right = (i + hs < n) ? (i + hs) : (n - 1);
                    const int hi = max(lo, min(i, a[right]));
                    const KmPre pi = pre[i + 1];
                    double bc = INFINITY;
                    int bj = 0x7fffffff;
                    for (int j = lo + lg; j <= hi; j += G) {
                        const KmPre pj = pre[j];
                        km_better(bc, bj, pj.dprev + km_cost(pj, pi.cw, pi.cwx, pi.cwxx), j);
                    }
                    for (int off = G >> 1; off > 0; off >>= 1) {
                        const double oc = __shfl_xor(bc, off);
                        const int oj = __shfl_xor(bj, off);
                        km_better(bc, bj, oc, oj);
                    }
                    if (lg == 0) {
                        dcur[i] = bc;
                        a[i] = bj;
                    }
                }
                __syncthreads();
            }
        }

        // ---- 4. backtrack + centroids ------------------------------------------------------------------------
        if (tid == 0) {
            int end = n - 1;
            float* out = T0 + (int64_t)row * V;
            for (int k = V - 1; k >= 0; --k) {
                int start = (end >= 0) ? arg[(size_t)k * n + end] : 0;
                if (k == 0) start = 0;
                if (end >= start && end >= 0) {
                    const double sw = pre[end + 1].cw - pre[start].cw;
                    const double swx = pre[end + 1].cwx - pre[start].cwx;
                    out[k] = (float)(sw > 0.0 ? swx / sw : val_at(start));
                } else {
                    out[k] = (k + 1 < V) ? out[k + 1] : (float)val_at(n - 1);
                }
                end = start - 1;
            }
        }
        __syncthreads();
    }
}

static size_t kmeans_stride(int64_t n, int V) {
    return align_up((size_t)(n + 1) * sizeof(KmPre), 256) + align_up((size_t)n * sizeof(double), 256) +
           align_up((size_t)V * (size_t)n * sizeof(int), 256);
}
static int kmeans_grid(int64_t m) { return (int)std::min<int64_t>(m, 1024); }

}  // namespace ganq

using namespace ganq;

extern "C" size_t ganq_kmeans_workspace_bytes(int64_t m, int64_t n, int V) {
    if (m <= 0 || n <= 0 || V <= 0) return 0;
    return (size_t)kmeans_grid(m) * kmeans_stride(n, V);
}

extern "C" int ganq_kmeans_init(const float* W, const double* col_weight, int64_t m, int64_t n, int V, float* T0,
                                void* workspace, size_t workspace_bytes, void* stream_) {
    if (m < 0 || n < 0) return fail(-1, "ganq_kmeans_init: negative shape");
    if (m == 0) return 0;
    if (n < 1 || V < 1 || V > 256) return fail(-2, "ganq_kmeans_init: bad n=%lld / V=%d", (long long)n, V);
    if (n > 16384) return fail(-2, "ganq_kmeans_init: n=%lld > 16384 not supported (LDS sort)", (long long)n);
    if (!W || !T0) return fail(-3, "ganq_kmeans_init: null pointer");
    const size_t need = ganq_kmeans_workspace_bytes(m, n, V);
    if (!workspace || workspace_bytes < need)
        return fail(-4, "ganq_kmeans_init: workspace %zu B < required %zu B", workspace_bytes, need);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    int P = 1;
    while (P < n) P <<= 1;
    const size_t smem = (size_t)P * sizeof(uint64_t);
    static size_t attr_smem = 0;
    if (smem > attr_smem) {
        GANQ_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kmeans_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr_smem = smem;
    }
    ProfScope prof(KID_KMEANS, stream);
    hipLaunchKernelGGL(kmeans_kernel, dim3(kmeans_grid(m)), dim3(256), smem, stream, W, col_weight, (int)m, (int)n, V, P, T0,
                       static_cast<char*>(workspace), kmeans_stride(n, V));
    GANQ_LAUNCH_CHECK();
    return 0;
}

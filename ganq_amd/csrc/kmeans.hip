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
//      (groups of up to 64 lanes scan one node and reduce with leftmost-minimum tie-break).  Round 3, resident kernel: the
//      argmin is also monotone in the number of clusters, opt[k-1][i] <= opt[k][i] (the cost is Monge), so the previous
//      layer's argmin at the SAME position is a second lower bound -- it cuts the long ranges of the top levels, where a
//      level costs its latency, by 3-10x and the candidates of a row by a fifth;
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

// the same order for a scan that visits its candidates with increasing j (one lane's share of a range): a later candidate
// only wins with a strictly smaller cost, so the leftmost minimum stays
__device__ __forceinline__ void km_better_asc(double& bc, int& bj, double c, int j) {
    if (c < bc) {
        bc = c;
        bj = j;
    }
}

// ---- LDS-resident variant (n up to ~4.6k): the prefix sums and D[k-1] live in LDS as four fp64 arrays, the argmins of
// the current layer as u16, one 1024-thread workgroup per row (persistent).  Same arithmetic, association order and
// tie-breaks as kmeans_kernel; only where the operands live differs -- the global-memory variant spends its time waiting
// on one dependent L2 round trip per candidate.
constexpr int KL_THREADS = 1024;
#ifndef KM_SPAN_HS
#define KM_SPAN_HS 512  // windowed kernel: from this node spacing down a layer may switch to solving its levels span by span (sweep: tools/dev/kmeans_span_sweep.py)
#endif
#ifndef KM_LONG
#define KM_LONG 16  // measured on MI355X (4096x4096, V=16): 4 -> 30 ms, 8 -> 23 ms, 16 -> 19 ms, 48 -> 22 ms
#endif
#ifdef GANQ_KMEANS_DEBUG
__device__ unsigned long long km_dbg[32];
#define KM_STAMP(slot) do { __syncthreads(); if (threadIdx.x == 0) { unsigned long long now_ = __builtin_amdgcn_s_memtime(); atomicAdd(&km_dbg[slot], now_ - last_); last_ = now_; } } while (0)
#else
#define KM_STAMP(slot) do {} while (0)
#endif

// Barrier for data exchanged through LDS only.  __syncthreads() also waits for this wave's outstanding GLOBAL stores
// (s_waitcnt vmcnt(0)); the levels of a DP layer hand each other nothing but the argmins in LDS, while every node also
// stores D[k][i] and its argmin to global memory (read again at the next layer / the backtrack, behind a full
// barrier) -- waiting for those stores at each of the ~170 level barriers per row exposed a store round trip each time.
__device__ __forceinline__ void km_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ double km_cost4(double cwj, double cwxj, double cwxxj, double cw_i1, double cwx_i1, double cwxx_i1) {
    const double w = cw_i1 - cwj;
    const double wx = cwx_i1 - cwxj;
    const double wxx = cwxx_i1 - cwxxj;
    if (!(w > 0.0)) return 0.0;
    const double c = wxx - (wx * wx) / w;
    return c > 0.0 ? c : 0.0;
}

// Nodes t0..t1 of one level, G <= 64 lanes per node (G a power of two; tid/G indexes the node within a pass).
// wcw.. are the operand arrays indexed by j - base, icw.. the same sums indexable by i + 1.
// The argmin jumps at cluster boundaries, so a few nodes of a level have ranges hundreds of candidates long while
// most have two or three: a node longer than KM_LONG*G candidates is left to the whole wave (64 lanes) right after the
// pass, instead of stalling the 63 other lanes of its wave.  The minimum with its (cost, j) tie-break is
// order-independent, so who scans what does not change the result.
__device__ __forceinline__ void km_level_nodes(const double* wcw, const double* wcwx, const double* wcwxx, const double* wdp,
                                               int base, const double* icw, const double* icwx, const double* icwxx,
                                               uint16_t* acur, double* dcur, uint16_t* ag, int t0, int t1, int hs, int n, int G,
                                               const uint16_t* aprev = nullptr) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int lg = tid & (G - 1);
    const int groups = KL_THREADS / G;
    for (int tw = t0 + (tid >> 6) * (64 / G); tw <= t1; tw += groups) {  // uniform per wave
        const int t = tw + lane / G;
        const bool valid = t <= t1;
        int i = 0, lo = 0, hi = -1;
        if (valid) {
            i = hs - 1 + t * 2 * hs;
            lo = (i - hs >= 0) ? (int)acur[i - hs] : 0;
            if (aprev) lo = max(lo, (int)aprev[i]);  // opt[k-1][i] <= opt[k][i]
            const int right = (i + hs < n) ? (i + hs) : (n - 1);
            hi = max(lo, min(i, (int)acur[right]));
        }
        const bool is_long = valid && (hi - lo + 1 > KM_LONG * G);
        double bc = INFINITY;
        int bj = 0x7fffffff;
        if (valid && !is_long) {
            const double ci = icw[i + 1], cxi = icwx[i + 1], cxxi = icwxx[i + 1];
            for (int j = lo + lg; j <= hi; j += G) {
                const int idx = j - base;
                km_better_asc(bc, bj, wdp[idx] + km_cost4(wcw[idx], wcwx[idx], wcwxx[idx], ci, cxi, cxxi), j);
            }
        }
        for (int off = G >> 1; off > 0; off >>= 1) {
            const double oc = __shfl_xor(bc, off);
            const int oj = __shfl_xor(bj, off);
            km_better(bc, bj, oc, oj);
        }
        if (valid && !is_long && lg == 0) {
            dcur[i] = bc;
            ag[i] = (uint16_t)bj;
            acur[i] = (uint16_t)bj;
        }
        uint64_t todo = __ballot(is_long && lg == 0);
        while (todo) {
            const int src = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            const int li = __builtin_amdgcn_readlane(i, src), llo = __builtin_amdgcn_readlane(lo, src),
                      lhi = __builtin_amdgcn_readlane(hi, src);
            const double ci = icw[li + 1], cxi = icwx[li + 1], cxxi = icwxx[li + 1];
            double c2 = INFINITY;
            int j2 = 0x7fffffff;
            for (int j = llo + lane; j <= lhi; j += 64) {
                const int idx = j - base;
                km_better_asc(c2, j2, wdp[idx] + km_cost4(wcw[idx], wcwx[idx], wcwxx[idx], ci, cxi, cxxi), j);
            }
            for (int off = 32; off > 0; off >>= 1) {
                const double oc = __shfl_xor(c2, off);
                const int oj = __shfl_xor(j2, off);
                km_better(c2, j2, oc, oj);
            }
            if (lane == 0) {
                dcur[li] = c2;
                ag[li] = (uint16_t)j2;
                acur[li] = (uint16_t)j2;
            }
        }
    }
}

// Balanced variant for the levels with many nodes (resident kernel).  A node's candidate range is short on average
// (~3) but heavy-tailed: wherever the argmin jumps, one node of a wave has tens to hundreds of candidates and its
// 63 neighbours wait.  Here every node takes its first KB_CAP candidates itself; what is left of the long ranges is
// queued (one 64-bit atomic hands out the queue slot and the offset of the range in a flat item space) and then
// spread evenly over all 1024 threads.  The per-node minimum over the spread items is an LDS atomic min on the cost
// bits (non-negative doubles order like their bit patterns), the leftmost-argmin tie-break a second atomic min on j
// among the items that reached that cost -- the same (cost, j) order as a sequential scan.
#ifndef KB_CAP
#define KB_CAP 12              // candidates every node evaluates itself, unrolled (4096x4096, V = 16 on MI355X: 2 -> 23.0 ms,
                               // 3 -> 20.6, 4 -> 19.4, 6 -> 17.7, 8 -> 17.1, 12 -> 16.9, 16 -> 17.3: the independent cost
                               // evaluations of one node overlap, a queued range costs a pass over the queue)
#endif
#ifndef KB_MIN_NODES
#define KB_MIN_NODES 512       // levels with at least this many nodes take the balanced path, the others G lanes per node
                               // (4096x4096, V = 16: 128 -> 16.0 ms, 256 / 512 -> 15.5, 1024 -> 16.1, 2048 -> 17.2, never -> 18.0)
#endif
#ifndef KB_RND
#define KB_RND 2               // candidates per round of a node's own scan (the rounds stop with the longest range of the wave;
                               // 17.4 -> 16.0 ms against always KB_CAP candidates; 1 / 4 per round: the same within 1 %)
#endif
#ifndef KB_CAP_1
#define KB_CAP_1 12            // ... of the level hs = 1 (measured, 4096x4096: 12 -> 17.0 ms, 4 -> 17.6, 3 -> 18.7, 2 -> 20.6)
#endif
#ifndef KB_CAP_2
#define KB_CAP_2 12            // hs = 2
#endif
#ifndef KB_CAP_4
#define KB_CAP_4 12            // hs = 4
#endif
struct KmQueue {               // LDS scratch of the balanced levels: `cap` queued ranges per level (more: finished in place)
    unsigned int* count;
    unsigned long long* seedc;              // best of the node's own first KB_CAP candidates
    unsigned int* seedj;
    unsigned short *t, *jstart, *cnt;
    int cap;
};
__host__ __device__ inline size_t km_queue_bytes(int cap) { return 16 + (size_t)cap * (8 + 4 + 2 + 2 + 2); }
__host__ inline int km_queue_cap(int64_t n) { return (int)std::min<int64_t>(1024, std::max<int64_t>(128, (n / 4 + 63) / 64 * 64)); }
__device__ __forceinline__ KmQueue km_queue_at(char* base, int cap) {
    KmQueue q;
    q.count = reinterpret_cast<unsigned int*>(base);
    q.seedc = reinterpret_cast<unsigned long long*>(base + 16);
    q.seedj = reinterpret_cast<unsigned int*>(base + 16 + (size_t)cap * 8);
    q.t = reinterpret_cast<unsigned short*>(base + 16 + (size_t)cap * 12);
    q.jstart = q.t + cap;
    q.cnt = q.jstart + cap;
    q.cap = cap;
    return q;
}

// One level with many nodes.  Most nodes have two or three candidates, a few (where the argmin jumps at a cluster
// boundary) have hundreds.  Pass A: every node evaluates its first KB_CAP candidates, unrolled -- no lane waits for a
// neighbour's longer loop; what is left of a range is queued.  Pass B: the queued ranges are dealt to 32 groups of 32
// lanes, 32 candidates per step, (cost, leftmost j) reduced inside the group.  The minimum with its tie-break is
// order-independent, so who scans what does not change the result.
// CAP = candidates every node evaluates itself; a compile-time parameter per level (KB_CAP_1 / _2 / _4 for the last three
// levels).  Smaller caps on the last levels -- whose ranges one would expect to be 1-2 candidates long -- were measured and
// lose: more ranges end up in the queue pass (defaults stay at 12 everywhere).
template <int CAP>
__device__ __forceinline__ void km_level_balanced(const double* cw, const double* cwx, const double* cwxx, const double* dprev,
                                                  uint16_t* acur, double* dcur, uint16_t* ag, int cnt, int hs, int n, const KmQueue& q,
                                                  const uint16_t* aprev) {
    const int tid = threadIdx.x;
    for (int t = tid; t < cnt; t += KL_THREADS) {
        const int i = hs - 1 + t * 2 * hs;
        const int lo = max((i - hs >= 0) ? (int)acur[i - hs] : 0, (int)aprev[i]);
        const int right = (i + hs < n) ? (i + hs) : (n - 1);
        const int hi = max(lo, min(i, (int)acur[right]));
        const double ci = cw[i + 1], cxi = cwx[i + 1], cxxi = cwxx[i + 1];
        double bc = INFINITY;
        int bj = 0x7fffffff;
        // two candidates per round, and no more rounds than the longest range of this wave needs (hs = 1: 3 candidates on
        // average, CAP only for the few nodes next to a jump of the argmin)
        const int need = min(hi - lo + 1, CAP);
#pragma unroll
        for (int c = 0; c < CAP; c += KB_RND) {
            if (!__any(c < need)) break;  // wave-uniform
#pragma unroll
            for (int d = 0; d < KB_RND && c + d < CAP; ++d) {
                const int j = min(lo + c + d, hi);  // past the range: the last candidate again (same cost: no effect)
                km_better_asc(bc, bj, dprev[j] + km_cost4(cw[j], cwx[j], cwxx[j], ci, cxi, cxxi), j);
            }
        }
        const int rest = hi - (lo + CAP - 1);
        bool queued = false;
        if (rest > 0) {
            const unsigned int idx = atomicAdd(q.count, 1u);
            if (idx < (unsigned int)q.cap) {
                q.t[idx] = (unsigned short)t;
                q.jstart[idx] = (unsigned short)(lo + CAP);
                q.cnt[idx] = (unsigned short)min(rest, 65535);
                q.seedc[idx] = (unsigned long long)__double_as_longlong(bc);
                q.seedj[idx] = (unsigned int)bj;
                queued = true;
            } else {
                for (int j = lo + CAP; j <= hi; ++j)
                    km_better_asc(bc, bj, dprev[j] + km_cost4(cw[j], cwx[j], cwxx[j], ci, cxi, cxxi), j);
            }
        }
        if (!queued) {
            dcur[i] = bc;
            ag[i] = (uint16_t)bj;
            acur[i] = (uint16_t)bj;
        }
    }
    km_lds_barrier();
    const int nq = min((int)*q.count, q.cap);
    const int grp = tid >> 5, l32 = tid & 31;
    for (int e = grp; e < nq; e += KL_THREADS / 32) {
        const int i = hs - 1 + (int)q.t[e] * 2 * hs;
        const int j0 = (int)q.jstart[e], len = (int)q.cnt[e];
        const double ci = cw[i + 1], cxi = cwx[i + 1], cxxi = cwxx[i + 1];
        double bc = INFINITY;
        int bj = 0x7fffffff;
        for (int d = l32; d < len; d += 32) {
            const int j = j0 + d;
            km_better_asc(bc, bj, dprev[j] + km_cost4(cw[j], cwx[j], cwxx[j], ci, cxi, cxxi), j);
        }
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) {
            const double oc = __shfl_xor(bc, off);
            const int oj = __shfl_xor(bj, off);
            km_better(bc, bj, oc, oj);
        }
        if (l32 == 0) {
            km_better(bc, bj, __longlong_as_double((long long)q.seedc[e]), (int)q.seedj[e]);  // own candidates lie left: they win ties
            dcur[i] = bc;
            ag[i] = (uint16_t)bj;
            acur[i] = (uint16_t)bj;
        }
    }
    km_lds_barrier();
    if (tid == 0) *q.count = 0;
    km_lds_barrier();
}

// Radix-4 level of the resident kernel (the TOP of a DP layer): the positions S4 - 1 + t S4 are solved (S4 = 4 S), this level
// solves the three positions in between, i = S - 1 + t S with t % 4 != 3, each bounded by the argmins of the nearest solved
// positions on its left and right.  Half as many levels as the binary scheme for 1.5 times the candidates -- up here a level
// costs its latency (barriers, reductions), not its work.  G lanes per node (a power of two): up to 64 reduce by shuffles,
// more go through the per-wave partials in LDS.  Same (cost, leftmost j) order as everywhere.
// Radix-4 steps run while the new spacing is at least KM_R4_MIN (0x7fffffff: binary levels only).  Measured, 4096x4096: binary
// only 15.5 ms, down to spacing 64: 15.0, 16: 15.3, 4: 18.6 (further down a level is bound by its work, which grows by half);
// 1024x2048 with two workgroups per CU: 1.96 / 1.78 / 1.75 / 1.69 ms.
#ifndef KM_R4_MIN
#define KM_R4_MIN (MINW == 8 ? 4 : 64)
#endif
__device__ __forceinline__ void km_level_r4(const double* cw, const double* cwx, const double* cwxx, const double* dprev, uint16_t* acur,
                                            double* dcur, uint16_t* ag, int S, int n, double* red_c, int* red_j, const uint16_t* aprev) {
    const int tid = threadIdx.x;
    const int T = (n - 1) / S;        // positions i = S - 1 + t S < n - 1
    const int cntU = T - T / 4;       // ... that are not solved yet
    if (cntU <= 0) return;            // uniform
    int G = 1;
    while (G < KL_THREADS && G * 2 * cntU <= KL_THREADS) G <<= 1;
    const int lg = tid & (G - 1);
    const int per_pass = KL_THREADS / G;
    for (int u0 = 0; u0 < cntU; u0 += per_pass) {  // one pass unless there are more nodes than threads (never at the top)
        const int u = u0 + tid / G;
        const bool valid = u < cntU;
        const int t = (u / 3) * 4 + u % 3;
        const int i = S - 1 + t * S;
        double bc = INFINITY;
        int bj = 0x7fffffff;
        if (valid) {
            const int left = i - ((t & 3) + 1) * S;
            const int right = min(i + (3 - (t & 3)) * S, n - 1);
            const int lo = max(left >= 0 ? (int)acur[left] : 0, (int)aprev[i]);
            const int hi = max(lo, min(i, (int)acur[right]));
            const double ci = cw[i + 1], cxi = cwx[i + 1], cxxi = cwxx[i + 1];
            for (int j = lo + lg; j <= hi; j += G)
                km_better_asc(bc, bj, dprev[j] + km_cost4(cw[j], cwx[j], cwxx[j], ci, cxi, cxxi), j);
        }
        for (int off = min(G, 64) >> 1; off > 0; off >>= 1) {
            const double oc = __shfl_xor(bc, off);
            const int oj = __shfl_xor(bj, off);
            km_better(bc, bj, oc, oj);
        }
        if (G > 64) {
            if ((tid & 63) == 0) {
                red_c[tid >> 6] = bc;
                red_j[tid >> 6] = bj;
            }
            km_lds_barrier();
            if (lg == 0 && valid) {
                const int w0 = tid >> 6;
                for (int w = 1; w < G / 64; ++w) km_better(bc, bj, red_c[w0 + w], red_j[w0 + w]);
            }
        }
        if (lg == 0 && valid) {
            dcur[i] = bc;
            ag[i] = (uint16_t)bj;
            acur[i] = (uint16_t)bj;
        }
        km_lds_barrier();
    }
}

// Radix-16 level at the very top of a layer (round 3): with the cross-layer bound a position needs no solved neighbour on
// its left -- [opt[k-1][i], min(i, opt[k][n-1])] is short enough (a few hundred candidates from the third layer on) -- so the
// 15 positions S - 1 + t S, S = P / 16, are solved in ONE level, one wave each, instead of two radix-4 levels.
#ifndef KM_TOP16
#define KM_TOP16 1
#endif
__device__ __forceinline__ void km_level_top16(const double* cw, const double* cwx, const double* cwxx, const double* dprev, uint16_t* acur,
                                               const uint16_t* aprev, double* dcur, uint16_t* ag, int S, int n) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int T = (n - 1) / S;  // positions i = S - 1 + t S < n - 1 (at most 15: one wave each)
    if (wv < T) {
        const int i = S - 1 + wv * S;
        const int lo = (int)aprev[i];
        const int hi = max(lo, min(i, (int)acur[n - 1]));
        const double ci = cw[i + 1], cxi = cwx[i + 1], cxxi = cwxx[i + 1];
        double bc = INFINITY;
        int bj = 0x7fffffff;
        for (int j = lo + lane; j <= hi; j += 64)
            km_better_asc(bc, bj, dprev[j] + km_cost4(cw[j], cwx[j], cwxx[j], ci, cxi, cxxi), j);
        for (int off = 32; off > 0; off >>= 1) {
            const double oc = __shfl_xor(bc, off);
            const int oj = __shfl_xor(bj, off);
            km_better(bc, bj, oc, oj);
        }
        if (lane == 0) {
            dcur[i] = bc;
            ag[i] = (uint16_t)bj;
            acur[i] = (uint16_t)bj;
        }
    }
    km_lds_barrier();
}

// Bitonic sort of a row's (value, column) keys, result in keys[0 .. P) (LDS), P = the power of two >= n, padding keys = ~0.
// Thread t holds the keys t, t + 1024, .. in REGISTERS: a pass whose partner distance j is >= 1024 exchanges inside the thread, a pass with
// j < 64 inside the wave (two 32-bit lane permutes per key, no barrier), and only the passes with 64 <= j < 1024 -- 18 of the 78 at
// P = 4096 -- go through LDS and the workgroup's barrier.  (Rounds 1-3 ran every pass as compare-exchanges on the LDS array: 135 k
// cycles per row at n = 4096, 7 % of the kernel.)  The keys are distinct (the column is part of the key), so the sorted order is the
// one and only: the same bits as before.
template <int E>
__device__ __forceinline__ void km_sort_regs(const float* __restrict__ wrow, int n, int P, uint64_t* keys, int tid) {
    uint64_t key[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int i = tid + KL_THREADS * e;
        uint64_t kv = ~0ull;
        if (i < n) {
            uint32_t b = __builtin_bit_cast(uint32_t, wrow[i]);
            b ^= (b >> 31) ? 0xffffffffu : 0x80000000u;
            kv = ((uint64_t)b << 32) | (uint32_t)i;
        }
        key[e] = kv;
    }
    auto keep = [](uint64_t a, uint64_t p, bool lower, bool asc) { return (lower == asc) ? (a < p ? a : p) : (a < p ? p : a); };
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j >= KL_THREADS) {  // inside the thread
                const int je = j / KL_THREADS;
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    if ((e & je) != 0 || (e | je) >= E) continue;
                    const int i = tid + KL_THREADS * e;
                    const bool asc = (i & k) == 0;
                    const uint64_t a = key[e], b = key[e | je];
                    const bool sw = (a > b) == asc;
                    key[e] = sw ? b : a;
                    key[e | je] = sw ? a : b;
                }
            } else if (j >= 64) {   // across waves: through LDS
#pragma unroll
                for (int e = 0; e < E; ++e) keys[tid + KL_THREADS * e] = key[e];
                __syncthreads();
                uint64_t pk[E];
#pragma unroll
                for (int e = 0; e < E; ++e) pk[e] = keys[(tid + KL_THREADS * e) ^ j];
                __syncthreads();
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const int i = tid + KL_THREADS * e;
                    key[e] = keep(key[e], pk[e], (i & j) == 0, (i & k) == 0);
                }
            } else {                // inside the wave
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const int i = tid + KL_THREADS * e;
                    const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)key[e], j), hi = (uint32_t)__shfl_xor((int)(uint32_t)(key[e] >> 32), j);
                    key[e] = keep(key[e], ((uint64_t)hi << 32) | lo, (i & j) == 0, (i & k) == 0);
                }
            }
        }
    }
#pragma unroll
    for (int e = 0; e < E; ++e) keys[tid + KL_THREADS * e] = key[e];
    __syncthreads();
}

// the LDS version (rows shorter than the workgroup, and rows of more than 4096 keys: 8 or 16 keys per thread in registers cost the
// windowed kernel more than the LDS passes -- n = 11008: 97.5 against 84.7 ms per 4096 rows)
__device__ __forceinline__ void km_sort_lds(const float* __restrict__ wrow, int n, int P, uint64_t* keys, int tid) {
    for (int i = tid; i < P; i += KL_THREADS) {
        uint64_t key = ~0ull;
        if (i < n) {
            uint32_t b = __builtin_bit_cast(uint32_t, wrow[i]);
            b ^= (b >> 31) ? 0xffffffffu : 0x80000000u;
            key = ((uint64_t)b << 32) | (uint32_t)i;
        }
        keys[i] = key;
    }
    __syncthreads();
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < P; i += KL_THREADS) {
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
            // a pass with j < 64 pairs elements of the same wave (thread t holds t, t + 1024, ..): it only has to wait for
            // that wave's own LDS traffic; the workgroup meets before the next pass that reaches across waves
            // (a pass that reached across waves itself must also be complete for everybody before anyone goes on)
            const int jn = j > 1 ? (j >> 1) : k;  // distance of the next pass
            if (j >= 64 || jn >= 64 || (j == 1 && k == P)) __syncthreads();
            else {
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
}

__device__ __forceinline__ void km_sort_row(const float* __restrict__ wrow, int n, int P, uint64_t* keys, int tid) {
    switch (P / KL_THREADS) {
        case 1: km_sort_regs<1>(wrow, n, P, keys, tid); break;
        case 2: km_sort_regs<2>(wrow, n, P, keys, tid); break;
        case 4: km_sort_regs<4>(wrow, n, P, keys, tid); break;
        default: km_sort_lds(wrow, n, P, keys, tid); break;
    }
}

// MINW = waves per SIMD the register allocation must allow: 4 (one workgroup per CU) or 8 (two, when two rows' arrays
// fit the LDS together: n <= 2.3 k)
template <int MINW>
__global__ __launch_bounds__(KL_THREADS, MINW) void kmeans_lds_kernel(const float* __restrict__ W, const double* __restrict__ col_weight,
                                                                      int m, int n, int V, int P, int qcap, float* __restrict__ T0,
                                                                      char* __restrict__ ws, size_t ws_stride) {
    extern __shared__ __align__(16) char km_smem[];
    const int n1 = n + 1;
    double* cw = reinterpret_cast<double*>(km_smem);
    double* cwx = cw + n1;
    double* cwxx = cwx + n1;
    double* dprev = cwxx + n1;
    uint16_t* acur = reinterpret_cast<uint16_t*>(dprev + n1);  // [n] argmins of the layer being solved
    uint16_t* aprev = acur + n;                                 // [n] ... of the layer before (they swap per layer)
    const KmQueue kq = km_queue_at(km_smem + align_up(4 * (size_t)n1 * sizeof(double) + 2 * (size_t)n * sizeof(uint16_t), 16), qcap);
    uint64_t* keys = reinterpret_cast<uint64_t*>(km_smem);     // [P] during the sort only (8P <= 16(n+1): over cw, cwx)
    double* ctot = dprev;                                       // [3][nchunk] during the prefix sums only
    __shared__ double red_c[KL_THREADS / 64];
    __shared__ int red_j[KL_THREADS / 64];

    const int tid = threadIdx.x;
    char* my = ws + (size_t)blockIdx.x * ws_stride;
    double* xs = reinterpret_cast<double*>(my);  // [n] sorted values
    double* wts = xs + n;                        // [n] their weights
    double* dcur = wts + n;                      // [n] D[k] of the layer being solved
    uint16_t* arg = reinterpret_cast<uint16_t*>(dcur + n);  // [V][n] argmins for the backtrack (16 bits: n <= 16384; int32 until round 4 -- half the bytes the kernel writes)
    const int nchunk = (n + KM_CHUNK - 1) / KM_CHUNK;

    for (int row = blockIdx.x; row < m; row += gridDim.x) {
#ifdef GANQ_KMEANS_DEBUG
        unsigned long long last_ = __builtin_amdgcn_s_memtime();
#endif
        // ---- 1. sort ------------------------------------------------------------------------------
        km_sort_row(W + (int64_t)row * n, n, P, keys, tid);
        for (int u = tid; u < n; u += KL_THREADS) {
            const uint64_t key = keys[u];
            uint32_t b = (uint32_t)(key >> 32);
            b ^= (b >> 31) ? 0x80000000u : 0xffffffffu;
            xs[u] = (double)__builtin_bit_cast(float, b);
            wts[u] = col_weight ? col_weight[(uint32_t)key] : 1.0;
        }
        __syncthreads();

        KM_STAMP(0);
        // ---- 2. prefix sums (chunks of 16, then chunk totals, both left to right) ----------------------
        for (int c = tid; c < nchunk; c += KL_THREADS) {
            double a = 0.0, b = 0.0, d = 0.0;
            const int hi = min(n, (c + 1) * KM_CHUNK);
            for (int u = c * KM_CHUNK; u < hi; ++u) {
                const double x = xs[u], w = wts[u];
                a += w;
                b += w * x;
                d += w * x * x;
            }
            ctot[c] = a;
            ctot[nchunk + c] = b;
            ctot[2 * nchunk + c] = d;
        }
        __syncthreads();
        if ((tid & 63) == 0 && (tid >> 6) < 3) {  // three waves, one lane each
            double* ct = ctot + (tid >> 6) * nchunk;
            double run = 0.0;
            for (int c = 0; c < nchunk; ++c) {
                const double t = ct[c];
                ct[c] = run;  // exclusive
                run += t;
            }
        }
        __syncthreads();
        for (int c = tid; c < nchunk; c += KL_THREADS) {
            double a = 0.0, b = 0.0, d = 0.0;
            const double oa = ctot[c], ob = ctot[nchunk + c], od = ctot[2 * nchunk + c];
            const int hi = min(n, (c + 1) * KM_CHUNK);
            for (int u = c * KM_CHUNK; u < hi; ++u) {
                cw[u] = oa + a;
                cwx[u] = ob + b;
                cwxx[u] = od + d;
                const double x = xs[u], w = wts[u];
                a += w;
                b += w * x;
                d += w * x * x;
            }
            if (hi == n) {
                cw[n] = oa + a;
                cwx[n] = ob + b;
                cwxx[n] = od + d;
            }
        }
        __syncthreads();

        KM_STAMP(1);
        if (tid == 0) *kq.count = 0;
        // ---- 3. DP ----------------------------------------------------------------------------------------
        for (int i = tid; i < n; i += KL_THREADS) {
            dcur[i] = km_cost4(cw[0], cwx[0], cwxx[0], cw[i + 1], cwx[i + 1], cwxx[i + 1]);
            arg[i] = 0;
            acur[i] = 0;  // layer 0: one cluster, every argmin is 0
        }
        __syncthreads();
        for (int k = 1; k < V; ++k) {
            __syncthreads();  // full barrier: D[k-1] (global) of every thread is complete
            {   // the finished layer's argmins become the lower bounds of this one
                uint16_t* tmp = aprev;
                aprev = acur;
                acur = tmp;
            }
            for (int j = tid; j <= n; j += KL_THREADS) dprev[j] = (j == 0) ? 0.0 : dcur[j - 1];
            km_lds_barrier();
            KM_STAMP(2);
            uint16_t* ag = arg + (size_t)k * n;
            {   // position n-1: full scan by the whole workgroup
                const int i = n - 1;
                const double ci = cw[i + 1], cxi = cwx[i + 1], cxxi = cwxx[i + 1];
                double bc = INFINITY;
                int bj = 0x7fffffff;
                for (int j = (int)aprev[i] + tid; j <= i; j += KL_THREADS)
                    km_better_asc(bc, bj, dprev[j] + km_cost4(cw[j], cwx[j], cwxx[j], ci, cxi, cxxi), j);
                for (int off = 32; off > 0; off >>= 1) {
                    const double oc = __shfl_xor(bc, off);
                    const int oj = __shfl_xor(bj, off);
                    km_better(bc, bj, oc, oj);
                }
                if ((tid & 63) == 0) {
                    red_c[tid >> 6] = bc;
                    red_j[tid >> 6] = bj;
                }
                km_lds_barrier();
                if (tid == 0) {
                    for (int w = 1; w < KL_THREADS / 64; ++w) km_better(bc, bj, red_c[w], red_j[w]);
                    dcur[i] = bc;
                    ag[i] = (uint16_t)bj;
                    acur[i] = (uint16_t)bj;
                }
                km_lds_barrier();
            }
            KM_STAMP(3);
            if (k == V - 1) break;  // only D[V-1][n-1] is needed from the last layer
            int sp = P;  // spacing of the solved positions (sp - 1 + t sp, and n - 1)
            if (KM_TOP16 && P >= 1024 && (P >> 4) >= KM_R4_MIN) {  // (n - 1) / (P / 16) <= 15 positions, 16 waves
                sp = P >> 4;
                km_level_top16(cw, cwx, cwxx, dprev, acur, aprev, dcur, ag, sp, n);
                KM_STAMP(4 + (31 - __builtin_clz(sp)));
            }
            while ((sp >> 2) >= KM_R4_MIN) {  // radix-4 steps at the top
                sp >>= 2;
                km_level_r4(cw, cwx, cwxx, dprev, acur, dcur, ag, sp, n, red_c, red_j, aprev);
                KM_STAMP(4 + (31 - __builtin_clz(sp)));
            }
            for (int hs = sp >> 1; hs >= 1; hs >>= 1) {
                const int cnt = (n - 1 > hs - 1) ? ((n - 1 - (hs - 1) + 2 * hs - 1) / (2 * hs)) : 0;
                if (cnt == 0) continue;
                int G = 1;
                while (G < KL_THREADS && G * 2 * cnt <= KL_THREADS) G <<= 1;
                const int lg = tid & (G - 1);
                if (cnt >= KB_MIN_NODES) {
                    if (hs == 1) km_level_balanced<KB_CAP_1>(cw, cwx, cwxx, dprev, acur, dcur, ag, cnt, hs, n, kq, aprev);
                    else if (hs == 2) km_level_balanced<KB_CAP_2>(cw, cwx, cwxx, dprev, acur, dcur, ag, cnt, hs, n, kq, aprev);
                    else if (hs == 4) km_level_balanced<KB_CAP_4>(cw, cwx, cwxx, dprev, acur, dcur, ag, cnt, hs, n, kq, aprev);
                    else km_level_balanced<KB_CAP>(cw, cwx, cwxx, dprev, acur, dcur, ag, cnt, hs, n, kq, aprev);
                } else if (G <= 64) {
                    km_level_nodes(cw, cwx, cwxx, dprev, 0, cw, cwx, cwxx, acur, dcur, ag, 0, cnt - 1, hs, n, G, aprev);
                    km_lds_barrier();
                } else {
                    // few nodes: several waves per node, partial minima through LDS
                    const int t = tid / G;
                    const int i = hs - 1 + t * 2 * hs;
                    double bc = INFINITY;
                    int bj = 0x7fffffff;
                    if (t < cnt) {
                        const int lo = max((i - hs >= 0) ? (int)acur[i - hs] : 0, (int)aprev[i]);
                        const int right = (i + hs < n) ? (i + hs) : (n - 1);
                        const int hi = max(lo, min(i, (int)acur[right]));
                        const double ci = cw[i + 1], cxi = cwx[i + 1], cxxi = cwxx[i + 1];
                        for (int j = lo + lg; j <= hi; j += G)
                            km_better_asc(bc, bj, dprev[j] + km_cost4(cw[j], cwx[j], cwxx[j], ci, cxi, cxxi), j);
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
                    km_lds_barrier();
                    if (lg == 0 && t < cnt) {
                        const int w0 = tid >> 6;
                        for (int w = 1; w < G / 64; ++w) km_better(bc, bj, red_c[w0 + w], red_j[w0 + w]);
                        dcur[i] = bc;
                        ag[i] = (uint16_t)bj;
                        acur[i] = (uint16_t)bj;
                    }
                    km_lds_barrier();
                }
                KM_STAMP(4 + (31 - __builtin_clz(hs)));
            }
        }

        __syncthreads();  // the argmins of every layer (global) are complete before the backtrack reads them
        // ---- 4. backtrack + centroids ------------------------------------------------------------------------
        if (tid == 0) {
            int end = n - 1;
            float* out = T0 + (int64_t)row * V;
            for (int k = V - 1; k >= 0; --k) {
                int start = (end >= 0) ? arg[(size_t)k * n + end] : 0;
                if (k == 0) start = 0;
                if (end >= start && end >= 0) {
                    const double sw = cw[end + 1] - cw[start];
                    const double swx = cwx[end + 1] - cwx[start];
                    out[k] = (float)(sw > 0.0 ? swx / sw : xs[start]);
                } else {
                    out[k] = (k + 1 < V) ? out[k + 1] : (float)xs[n - 1];
                }
                end = start - 1;
            }
        }
        KM_STAMP(20);
        __syncthreads();
    }
}

// ---- windowed variant (any n <= 16384): prefix sums and D[k] live in global memory (L2); every level walks its nodes
// in segments whose candidate window [lo(first), hi(last)] fits the LDS window, stages the four operand arrays of
// that window, and then works exactly like the resident variant.  A node whose own range is longer than the window
// (top levels, and the full scan at i = n-1) is scanned by the whole workgroup piece by piece.
template <int MINW>
__global__ __launch_bounds__(KL_THREADS, MINW) void kmeans_win_kernel(const float* __restrict__ W, const double* __restrict__ col_weight,
                                                                int m, int n, int V, int P, int Wcap, float* __restrict__ T0,
                                                                char* __restrict__ ws, size_t ws_stride, int span_hs) {
    extern __shared__ __align__(16) char km_smem[];
    double* wcw = reinterpret_cast<double*>(km_smem);  // window copies, index j - base
    double* wcwx = wcw + Wcap;
    double* wcwxx = wcwx + Wcap;
    double* wdp = wcwxx + Wcap;
    uint16_t* acur = reinterpret_cast<uint16_t*>(wdp + Wcap);  // [n] argmins of the layer being solved
    uint16_t* aprev = acur + n;                                 // [n] ... of the layer before: the cross-layer lower bound
    uint64_t* keys = reinterpret_cast<uint64_t*>(km_smem);     // [P] during the sort only (may run over acur / aprev)
    double* ctot = reinterpret_cast<double*>(km_smem);         // [3][nchunk] during the prefix sums only
    __shared__ double red_c[KL_THREADS / 64];
    __shared__ int red_j[KL_THREADS / 64];

    const int tid = threadIdx.x;
    const int n1 = n + 1;
    char* my = ws + (size_t)blockIdx.x * ws_stride;
    double* xs = reinterpret_cast<double*>(my);  // [n] sorted values
    double* wts = xs + n;                        // [n] their weights
    double* cw = wts + n;                        // [n+1] prefix sums
    double* cwx = cw + n1;
    double* cwxx = cwx + n1;
    double* dbuf0 = cwxx + n1;                   // [n] D of even layers
    double* dbuf1 = dbuf0 + n;                   // [n] D of odd layers
    uint16_t* arg = reinterpret_cast<uint16_t*>(dbuf1 + n);  // [V][n]
    const int nchunk = (n + KM_CHUNK - 1) / KM_CHUNK;

    for (int row = blockIdx.x; row < m; row += gridDim.x) {
        // ---- 1. sort ------------------------------------------------------------------------------
        km_sort_row(W + (int64_t)row * n, n, P, keys, tid);
        for (int u = tid; u < n; u += KL_THREADS) {
            const uint64_t key = keys[u];
            uint32_t b = (uint32_t)(key >> 32);
            b ^= (b >> 31) ? 0x80000000u : 0xffffffffu;
            xs[u] = (double)__builtin_bit_cast(float, b);
            wts[u] = col_weight ? col_weight[(uint32_t)key] : 1.0;
        }
        __syncthreads();

        // ---- 2. prefix sums (chunks of 16, then chunk totals, both left to right) ----------------------
        for (int c = tid; c < nchunk; c += KL_THREADS) {
            double a = 0.0, b = 0.0, d = 0.0;
            const int hi = min(n, (c + 1) * KM_CHUNK);
            for (int u = c * KM_CHUNK; u < hi; ++u) {
                const double x = xs[u], w = wts[u];
                a += w;
                b += w * x;
                d += w * x * x;
            }
            ctot[c] = a;
            ctot[nchunk + c] = b;
            ctot[2 * nchunk + c] = d;
        }
        __syncthreads();
        if ((tid & 63) == 0 && (tid >> 6) < 3) {
            double* ct = ctot + (tid >> 6) * nchunk;
            double run = 0.0;
            for (int c = 0; c < nchunk; ++c) {
                const double t = ct[c];
                ct[c] = run;  // exclusive
                run += t;
            }
        }
        __syncthreads();
        for (int c = tid; c < nchunk; c += KL_THREADS) {
            double a = 0.0, b = 0.0, d = 0.0;
            const double oa = ctot[c], ob = ctot[nchunk + c], od = ctot[2 * nchunk + c];
            const int hi = min(n, (c + 1) * KM_CHUNK);
            for (int u = c * KM_CHUNK; u < hi; ++u) {
                cw[u] = oa + a;
                cwx[u] = ob + b;
                cwxx[u] = od + d;
                const double x = xs[u], w = wts[u];
                a += w;
                b += w * x;
                d += w * x * x;
            }
            if (hi == n) {
                cw[n] = oa + a;
                cwx[n] = ob + b;
                cwxx[n] = od + d;
            }
        }
        __syncthreads();

        // ---- 3. DP ----------------------------------------------------------------------------------------
        {
            const double c0 = cw[0], cx0 = cwx[0], cxx0 = cwxx[0];
            for (int i = tid; i < n; i += KL_THREADS) {
                dbuf0[i] = km_cost4(c0, cx0, cxx0, cw[i + 1], cwx[i + 1], cwxx[i + 1]);
                arg[i] = 0;
                acur[i] = 0;  // layer 0: one cluster, every argmin is 0
            }
        }
        __syncthreads();
        for (int k = 1; k < V; ++k) {
            {   // opt[k-1][i] <= opt[k][i]: the finished layer's argmins bound this one's ranges from below -- up here that
                // also shrinks the windows that have to be staged (the top levels' ranges by 3-10x)
                uint16_t* tmp = aprev;
                aprev = acur;
                acur = tmp;
            }
            const double* dprev_g = (k & 1) ? dbuf0 : dbuf1;  // D[k-1]
            double* dcur = (k & 1) ? dbuf1 : dbuf0;
            uint16_t* ag = arg + (size_t)k * n;
            auto stage = [&](int base, int len) {  // window := [base, base+len)
                for (int idx = tid; idx < len; idx += KL_THREADS) {
                    const int j = base + idx;
                    wcw[idx] = cw[j];
                    wcwx[idx] = cwx[j];
                    wcwxx[idx] = cwxx[j];
                    wdp[idx] = (j == 0) ? 0.0 : dprev_g[j - 1];
                }
            };
            // one node scanned by the whole workgroup, window piece by window piece
            auto solve_wide = [&](int i, int lo, int hi) {
                const double ci = cw[i + 1], cxi = cwx[i + 1], cxxi = cwxx[i + 1];
                double bc = INFINITY;
                int bj = 0x7fffffff;
                for (int pb = lo; pb <= hi; pb += Wcap) {
                    const int len = min(Wcap, hi - pb + 1);
                    stage(pb, len);
                    __syncthreads();
                    for (int idx = tid; idx < len; idx += KL_THREADS)
                        km_better_asc(bc, bj, wdp[idx] + km_cost4(wcw[idx], wcwx[idx], wcwxx[idx], ci, cxi, cxxi), pb + idx);
                    __syncthreads();
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
                    for (int w = 1; w < KL_THREADS / 64; ++w) km_better(bc, bj, red_c[w], red_j[w]);
                    dcur[i] = bc;
                    ag[i] = (uint16_t)bj;
                    acur[i] = (uint16_t)bj;
                }
                __syncthreads();
            };
            solve_wide(n - 1, (int)aprev[n - 1], n - 1);
            if (k == V - 1) break;  // only D[V-1][n-1] is needed from the last layer
            // the nodes t0 .. t1 of level hs, whose candidates all lie in the staged window [base, ..)
            auto run_nodes = [&](int hs, int t0, int t1, int base) {
                auto node_lo = [&](int t) {
                    const int i = hs - 1 + t * 2 * hs;
                    return max((i - hs >= 0) ? (int)acur[i - hs] : 0, (int)aprev[i]);
                };
                auto node_hi = [&](int t) {
                    const int i = hs - 1 + t * 2 * hs;
                    const int right = (i + hs < n) ? (i + hs) : (n - 1);
                    return max(node_lo(t), min(i, (int)acur[right]));
                };
                const int segcnt = t1 - t0 + 1;
                int G = 1;
                while (G < KL_THREADS && G * 2 * segcnt <= KL_THREADS) G <<= 1;
                const int lg = tid & (G - 1);
                if (G <= 64) {
                    km_level_nodes(wcw, wcwx, wcwxx, wdp, base, cw, cwx, cwxx, acur, dcur, ag, t0, t1, hs, n, G, aprev);
                    __syncthreads();
                } else {
                    const int t = t0 + tid / G;
                    const int i = hs - 1 + t * 2 * hs;
                    double bc = INFINITY;
                    int bj = 0x7fffffff;
                    if (t <= t1) {
                        const int lo = node_lo(t), hi = node_hi(t);
                        const double ci = cw[i + 1], cxi = cwx[i + 1], cxxi = cwxx[i + 1];
                        for (int j = lo + lg; j <= hi; j += G) {
                            const int idx = j - base;
                            km_better_asc(bc, bj, wdp[idx] + km_cost4(wcw[idx], wcwx[idx], wcwxx[idx], ci, cxi, cxxi), j);
                        }
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
                    if (lg == 0 && t <= t1) {
                        const int w0 = tid >> 6;
                        for (int w = 1; w < G / 64; ++w) km_better(bc, bj, red_c[w0 + w], red_j[w0 + w]);
                        dcur[i] = bc;
                        ag[i] = (uint16_t)bj;
                        acur[i] = (uint16_t)bj;
                    }
                    __syncthreads();
                }
            };
            // The upper levels: one level at a time, every level walks its nodes in segments whose candidate window
            // fits the LDS window (the top levels' ranges are long).
            // The lower levels (round 4): SPAN by span.  Behind a level of spacing 2 hs the positions D - 1, 2 D - 1, .. (D = 2 hs) and
            // n - 1 are solved; every node between two of them has its candidates between THEIR argmins (monotone), so a run of
            // consecutive intervals whose window [max(opt(left end), lower bound of the first position), opt(right end)] fits is
            // staged ONCE and all the levels inside it are solved from that copy.  The switch happens at the first level (from
            // spacing `span_hs` down) at which every single interval fits the window -- per layer: 256-1024 at n = 8192-14336.
            // Rounds 1-3 staged per level: thirteen windows' worth of four arrays per layer at n = 8192 instead of five,
            // ~600 segment set-ups per row.  Same candidates, same (cost, leftmost j) order: the same bits.
            for (int hs = P >> 1; hs >= 1; hs >>= 1) {
                if (hs <= span_hs) {  // from the first level on whose every interval fits the window: span by span
                    const int D = 2 * hs;
                    const int nint = (n - 1 + D - 1) / D;  // intervals (u D - 1, min((u + 1) D - 1, n - 1)), u = 0 .. nint - 1
                    auto span_lo = [&](int u) {            // lowest candidate of any node inside interval u
                        const int iL = u * D - 1;
                        return max(iL >= 0 ? (int)acur[iL] : 0, (int)aprev[iL + 1]);
                    };
                    auto span_hi = [&](int u) { return (int)acur[min((u + 1) * D - 1, n - 1)]; };
                    bool fits = true;
                    for (int u = 0; u < nint && fits; ++u) fits = span_hi(u) - span_lo(u) + 1 <= Wcap;
                    if (fits) {
                        int u0 = 0;
                        while (u0 < nint) {
                            const int base = span_lo(u0);
                            int u1 = u0;
                            while (u1 + 1 < nint && span_hi(u1 + 1) - base + 1 <= Wcap) ++u1;
                            stage(base, span_hi(u1) - base + 1);
                            __syncthreads();
                            const int iR = min((u1 + 1) * D - 1, n - 1);
                            for (int h = hs; h >= 1; h >>= 1) {
                                const int cnt_h = (n - 1 > h - 1) ? ((n - 1 - (h - 1) + 2 * h - 1) / (2 * h)) : 0;
                                const int ta = u0 * (D / (2 * h));
                                const int tb = min(cnt_h - 1, (iR - h) / (2 * h));  // positions h - 1 + 2 h t < iR
                                if (iR - h >= 0 && tb >= ta) run_nodes(h, ta, tb, base);
                            }
                            u0 = u1 + 1;
                        }
                        break;  // every level from hs down is done
                    }
                }
                const int cnt = (n - 1 > hs - 1) ? ((n - 1 - (hs - 1) + 2 * hs - 1) / (2 * hs)) : 0;
                auto node_lo = [&](int t) {
                    const int i = hs - 1 + t * 2 * hs;
                    return max((i - hs >= 0) ? (int)acur[i - hs] : 0, (int)aprev[i]);
                };
                auto node_hi = [&](int t) {
                    const int i = hs - 1 + t * 2 * hs;
                    const int right = (i + hs < n) ? (i + hs) : (n - 1);
                    return max(node_lo(t), min(i, (int)acur[right]));
                };
                int t0 = 0;
                while (t0 < cnt) {  // every quantity below is uniform over the workgroup
                    const int base = node_lo(t0);
                    if (node_hi(t0) - base + 1 > Wcap) {
                        solve_wide(hs - 1 + t0 * 2 * hs, base, node_hi(t0));
                        ++t0;
                        continue;
                    }
                    int t1 = t0;  // last node of the segment: largest t with hi(t) < base + Wcap (hi is non-decreasing)
                    for (int lo_t = t0, hi_t = cnt - 1; lo_t <= hi_t;) {
                        const int mid = (lo_t + hi_t) >> 1;
                        if (node_hi(mid) - base + 1 <= Wcap) {
                            t1 = mid;
                            lo_t = mid + 1;
                        } else {
                            hi_t = mid - 1;
                        }
                    }
                    stage(base, node_hi(t1) - base + 1);
                    __syncthreads();
                    run_nodes(hs, t0, t1, base);
                    t0 = t1 + 1;
                }
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
                    const double sw = cw[end + 1] - cw[start];
                    const double swx = cwx[end + 1] - cwx[start];
                    out[k] = (float)(sw > 0.0 ? swx / sw : xs[start]);
                } else {
                    out[k] = (k + 1 < V) ? out[k + 1] : (float)xs[n - 1];
                }
                end = start - 1;
            }
        }
        __syncthreads();
    }
}

struct KmPlan {
    bool lds;
    int grid, P, Wcap, qcap = 0, per_cu = 1;
    size_t smem, stride;
};
constexpr size_t KM_LDS_BUDGET = 160 * 1024 - 1024;  // dynamic part; the rest covers the static reduction slots
static KmPlan kmeans_plan(int64_t m, int64_t n, int V) {
    KmPlan p;
    p.P = 1;
    while (p.P < n) p.P <<= 1;
    p.Wcap = 0;
    const size_t acur_bytes = align_up(2 * (size_t)n * sizeof(uint16_t), 16);  // the layer's argmins and the previous layer's
    p.qcap = km_queue_cap(n);
    // the second argmin array (round 3) is paid for by a shorter queue: first to keep TWO rows per CU where the arrays allow
    // it at all (n <= 2.3 k), otherwise to stay inside the LDS with one
    const size_t arrays = align_up(4 * (size_t)(n + 1) * sizeof(double) + 2 * (size_t)n * sizeof(uint16_t), 16);
    const size_t half_budget = (KM_LDS_BUDGET + 1024) / 2 - 1024;
    const size_t target = arrays + km_queue_bytes(128) <= half_budget ? half_budget : KM_LDS_BUDGET;
    while (arrays + km_queue_bytes(p.qcap) > target && p.qcap > 128) p.qcap -= 64;
    const size_t lds_bytes = arrays + km_queue_bytes(p.qcap);
    p.lds = lds_bytes <= KM_LDS_BUDGET;
    const int forced = (int)opt_get(OPT_KMEANS_WCAP);  // testing: force the windowed kernel with a small window
    if (forced > 0) p.lds = false;
    if (p.lds) {
        p.smem = lds_bytes;
        const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(2, (KM_LDS_BUDGET + 1024) / (lds_bytes + 1024)));
        p.per_cu = per_cu;
        p.grid = (int)std::min<int64_t>(m, 256 * per_cu);
        p.stride = align_up(3 * (size_t)n * sizeof(double) + (size_t)V * (size_t)n * sizeof(uint16_t), 256);
    } else {
        p.Wcap = (int)((KM_LDS_BUDGET - acur_bytes) / (4 * sizeof(double))) & ~63;
        // two workgroups per CU with half the window each, as long as the sort keys of a row (8 B x P) fit half the
        // LDS: a small gain here (2048x8192: 33.1 -> 32.1 ms) -- unlike the resident kernel (1.4x from the second
        // workgroup) this one is bound by staging the windows, not by latency
        const size_t half = (KM_LDS_BUDGET + 1024) / 2 - 1024;
        if ((size_t)p.P * sizeof(uint64_t) <= half && half > acur_bytes + 4 * 512 * sizeof(double) && m > 256) {
            p.per_cu = 2;
            p.Wcap = (int)((half - acur_bytes) / (4 * sizeof(double))) & ~63;
        }
        if (forced > 0) p.Wcap = std::min(p.Wcap, forced);
        p.smem = std::max((size_t)p.P * sizeof(uint64_t), 4 * (size_t)p.Wcap * sizeof(double) + acur_bytes);
        p.grid = (int)std::min<int64_t>(m, 256 * p.per_cu);
        p.stride = align_up((4 * (size_t)n + 3 * (size_t)(n + 1)) * sizeof(double) + (size_t)V * (size_t)n * sizeof(uint16_t), 256);
    }
    return p;
}

}  // namespace ganq

using namespace ganq;

extern "C" size_t ganq_kmeans_workspace_bytes(int64_t m, int64_t n, int V) {
    if (m <= 0 || n <= 0 || V <= 0) return 0;
    const KmPlan p = kmeans_plan(m, n, V);
    return (size_t)p.grid * p.stride;
}

extern "C" int ganq_kmeans_init(const float* W, const double* col_weight, int64_t m, int64_t n, int V, float* T0,
                                void* workspace, size_t workspace_bytes, void* stream_) {
    if (m < 0 || n < 0) return fail(-1, "ganq_kmeans_init: negative shape");
    if (m == 0) return 0;
    if (n < 1 || V < 1 || V > 256) return fail(-2, "ganq_kmeans_init: bad n=%lld / V=%d", (long long)n, V);
    if (n > 16384) return fail(-2, "ganq_kmeans_init: n=%lld > 16384 not supported (LDS sort)", (long long)n);
    if (!W || !T0) return fail(-3, "ganq_kmeans_init: null pointer");
    const KmPlan p = kmeans_plan(m, n, V);
    const size_t need = (size_t)p.grid * p.stride;
    if (!workspace || workspace_bytes < need)
        return fail(-4, "ganq_kmeans_init: workspace %zu B < required %zu B", workspace_bytes, need);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    {
        int rc;
        if (p.lds) {
            rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kmeans_lds_kernel<4>), p.smem);
            if (!rc) rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kmeans_lds_kernel<8>), p.smem);
        } else {
            rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kmeans_win_kernel<4>), p.smem);
            if (!rc) rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kmeans_win_kernel<8>), p.smem);
        }
        if (rc) return rc;
    }
    ProfScope prof(KID_KMEANS, stream);
    // (windowed kernel) levels with this spacing and below are solved span by span; GANQ_KMEANS_SPAN (developer): another power of two, 0 = never
    int span_hs = (int)opt_get(OPT_KMEANS_SPAN);
    if (span_hs < 0) span_hs = KM_SPAN_HS;
    if (p.lds && p.per_cu >= 2)
        hipLaunchKernelGGL(kmeans_lds_kernel<8>, dim3(p.grid), dim3(KL_THREADS), p.smem, stream, W, col_weight, (int)m, (int)n, V,
                           p.P, p.qcap, T0, static_cast<char*>(workspace), p.stride);
    else if (p.lds)
        hipLaunchKernelGGL(kmeans_lds_kernel<4>, dim3(p.grid), dim3(KL_THREADS), p.smem, stream, W, col_weight, (int)m, (int)n, V,
                           p.P, p.qcap, T0, static_cast<char*>(workspace), p.stride);
    else
        if (p.per_cu >= 2)
            hipLaunchKernelGGL(kmeans_win_kernel<8>, dim3(p.grid), dim3(KL_THREADS), p.smem, stream, W, col_weight, (int)m, (int)n,
                               V, p.P, p.Wcap, T0, static_cast<char*>(workspace), p.stride, span_hs);
        else
            hipLaunchKernelGGL(kmeans_win_kernel<4>, dim3(p.grid), dim3(KL_THREADS), p.smem, stream, W, col_weight, (int)m, (int)n,
                               V, p.P, p.Wcap, T0, static_cast<char*>(workspace), p.stride, span_hs);
    GANQ_LAUNCH_CHECK();
    return 0;
}

#ifdef GANQ_KMEANS_DEBUG
extern "C" int ganq_debug_kmeans_cycles(unsigned long long* out32) {
    GANQ_HIP_CHECK(hipDeviceSynchronize());
    GANQ_HIP_CHECK(hipMemcpyFromSymbol(out32, HIP_SYMBOL(ganq::km_dbg), 32 * sizeof(unsigned long long)));
    unsigned long long z[32] = {0};
    GANQ_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(ganq::km_dbg), z, sizeof(z)));
    return 0;
}
#endif

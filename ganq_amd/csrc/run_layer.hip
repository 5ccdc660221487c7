// The whole alternating optimisation of one layer (reference ganq.py:516-634):
//   K x ( S-solve -> T-update -> loss ) with best-of-K selection, enqueued on one stream with no host
//   synchronisation: the "is this iteration the best so far" decision and the conditional copies run on
//   the device.
#include <cmath>
#include <mutex>

#include "common.h"
#include "solve_s.h"
#include "update_t.h"

namespace ganq {

__global__ void best_init_kernel(double* best, int32_t* best_k, int32_t* flag) {
    best[0] = INFINITY;  // best[k & 1] is read by iteration k, best[(k + 1) & 1] written
    *best_k = -1;
    *flag = 0;
}

// ganq.py:621-626 in one launch: dist = sum of the per-row losses (fixed order), if dist < best: best = (dist, T)
// (strict <, NaN never wins).  Every workgroup forms the same sum and takes the same decision; the running best is
// double-buffered (read [k & 1], write [(k + 1) & 1]) so that no workgroup can see this launch's update.
__global__ __launch_bounds__(256) void select_copy_kernel(const double* __restrict__ loss_rows, int m, int k, double* best,
                                                          int32_t* best_k, int32_t* flag, double* __restrict__ dists,
                                                          const uint32_t* __restrict__ src, uint32_t* __restrict__ dst,
                                                          int64_t words) {
    __shared__ double sh[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < m; i += 256) s += loss_rows[i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
        __syncthreads();
    }
    const double d = sh[0], b = best[k & 1];
    const bool win = d < b;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        dists[k] = d;
        best[(k + 1) & 1] = win ? d : b;
        if (win) *best_k = k;
        *flag = win ? 1 : 0;
    }
    if (!win) return;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < words; i += stride) dst[i] = src[i];
}

__global__ __launch_bounds__(256) void copy_if_kernel(const int32_t* __restrict__ flag, const uint32_t* __restrict__ src,
                                                      uint32_t* __restrict__ dst, int64_t words,
                                                      const uint8_t* __restrict__ src_tail, uint8_t* __restrict__ dst_tail,
                                                      int tail) {
    if (*flag == 0) return;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < words; i += stride) dst[i] = src[i];
    if (blockIdx.x == 0 && (int)threadIdx.x < tail) dst_tail[threadIdx.x] = src_tail[threadIdx.x];
}

// "no iteration won" (all distances NaN): hand back the last codebook instead of leaving T_best unwritten
__global__ __launch_bounds__(256) void copy_if_none_kernel(const int32_t* __restrict__ best_k, const float* __restrict__ src,
                                                           float* __restrict__ dst, int64_t count) {
    if (*best_k >= 0) return;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += stride) dst[i] = src[i];
}

static int launch_copy_if(const int32_t* flag, const void* src, void* dst, size_t bytes, hipStream_t stream) {
    const int64_t words = (int64_t)(bytes / 4);
    const int tail = (int)(bytes % 4);
    const int blocks = (int)std::min<int64_t>(1024, std::max<int64_t>(1, (words + 255) / 256));
    hipLaunchKernelGGL(copy_if_kernel, dim3(blocks), dim3(256), 0, stream, flag, static_cast<const uint32_t*>(src),
                       static_cast<uint32_t*>(dst), words, static_cast<const uint8_t*>(src) + words * 4,
                       static_cast<uint8_t*>(dst) + words * 4, tail);
    GANQ_LAUNCH_CHECK();
    return 0;
}

// One helper stream + two events per (device, caller stream), created on first use (the pattern of cholesky.hip): the
// iteration-invariant preparation of the T-update (fixed-point planes of H, W @ H) needs nothing of the first S-solve and
// the first S-solve nothing of it, so the two CAN be enqueued side by side: GANQ_PREP_OVERLAP=1.  Measured without a gain at
// 4096 x 4096 (see runtime.hip), so the default keeps one stream.
struct PrepSide {
    hipStream_t side;
    hipEvent_t fork, join;
};
static PrepSide* prep_side_for(hipStream_t main) {
    struct Slot {
        int dev;
        hipStream_t main;
        PrepSide ps;
    };
    static Slot slots[32];
    static int used = 0;
    static std::mutex mu;
    int dev = 0;
    if (opt_get(OPT_PREP_OVERLAP) == 0 || hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    for (int i = 0; i < used; ++i)
        if (slots[i].dev == dev && slots[i].main == main) return &slots[i].ps;
    if (used == 32) return nullptr;  // more caller streams than slots: plain single-stream sequence
    PrepSide ps{};
    if (hipStreamCreateWithFlags(&ps.side, hipStreamNonBlocking) != hipSuccess) return nullptr;
    if (hipEventCreateWithFlags(&ps.fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ps.join, hipEventDisableTiming) != hipSuccess)
        return nullptr;
    slots[used] = Slot{dev, main, ps};
    return &slots[used++].ps;
}

struct RunLayout {
    size_t off_t0, off_t1, off_q, off_solve, off_upd, off_best, off_flag, total;
    size_t solve_bytes;
    TLayout t;
};

static RunLayout run_layout(int64_t m, int64_t n, int V) {
    RunLayout lo;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        size_t o = off;
        off = align_up(off + bytes, 256);
        return o;
    };
    lo.solve_bytes = ganq_solve_s_workspace_bytes(m, n, V);
    lo.t = t_layout(m, n, true);
    lo.off_t0 = take((size_t)m * V * sizeof(float));
    lo.off_t1 = take((size_t)m * V * sizeof(float));
    lo.off_q = take((size_t)m * n);
    lo.off_solve = take(lo.solve_bytes);
    lo.off_upd = take(lo.t.total);
    lo.off_best = take(2 * sizeof(double));
    lo.off_flag = take(sizeof(int32_t));
    lo.total = off;
    return lo;
}

}  // namespace ganq

using namespace ganq;

extern "C" size_t ganq_run_layer_workspace_bytes(int64_t m, int64_t n, int V) {
    if (m <= 0 || n <= 0) return 0;
    return run_layout(m, n, V).total;
}

// shared body of ganq_run_layer / ganq_run_layer_rows.  T_all [K,m,V], loss_rows_all [K,m], Q_all [K,m,n]: optional
// per-iteration records (device; null = not wanted)
static int run_layer_impl(const float* W, const float* H, const float* L, int64_t ldl, const float* T0, int64_t m,
                          int64_t n, int V, int K, uint32_t flags, double rcond, float* T_best, uint8_t* Q_out,
                          double* dists, int32_t* best_k, float* T_all, double* loss_rows_all, uint8_t* Q_all,
                          void* workspace, size_t workspace_bytes, void* stream_) {
    if (m < 0 || n < 0 || K < 0) return fail(-1, "ganq_run_layer: negative shape / K");
    if (m == 0 || n == 0 || K == 0) return 0;
    if (V < 2 || V > 16) return fail(-2, "ganq_run_layer: V=%d not supported (bits 2..4 are implemented)", V);
    if (!W || !H || !L || !T0 || !T_best || !Q_out || !dists || !best_k) return fail(-3, "ganq_run_layer: null pointer");
    if (ldl < n) return fail(-1, "ganq_run_layer: ldl=%lld < n=%lld", (long long)ldl, (long long)n);
    if (n > 16384) return fail(-1, "ganq_run_layer: n=%lld > 16384 not supported", (long long)n);
    const RunLayout lo = run_layout(m, n, V);
    if (!workspace || workspace_bytes < lo.total)
        return fail(-4, "ganq_run_layer: workspace %zu B < required %zu B", workspace_bytes, lo.total);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    int rc = ganq_hip_selftest(stream_);
    if (rc) return rc;
    char* ws = static_cast<char*>(workspace);
    float* Tc = reinterpret_cast<float*>(ws + lo.off_t0);
    float* Tn = reinterpret_cast<float*>(ws + lo.off_t1);
    uint8_t* Qc = reinterpret_cast<uint8_t*>(ws + lo.off_q);
    double* best = reinterpret_cast<double*>(ws + lo.off_best);
    int32_t* flag = reinterpret_cast<int32_t*>(ws + lo.off_flag);
    const bool alias = (flags & GANQ_FLAG_ALIAS_Q) != 0;
    // with the reference's aliasing the returned indices are simply those of the last iteration
    uint8_t* Qwork = alias ? Q_out : Qc;

    if (rcond < 0) rcond = 1.1920928955078125e-07 * (double)V;
    // iteration-invariant part of the T-update / loss: fixed-point planes of H, W @ H and w^T H w in fp64 (ganq.py:590);
    // on the helper stream, beside the first S-solve (joined before the first T-update)
    PrepSide* ps = profile_enabled(KID_T_PREP) ? nullptr : prep_side_for(stream);  // per-kernel timing wants one stream
    if (ps) {
        GANQ_HIP_CHECK(hipEventRecord(ps->fork, stream));
        GANQ_HIP_CHECK(hipStreamWaitEvent(ps->side, ps->fork, 0));
    }
    rc = t_prepare(W, H, m, n, lo.t, ws + lo.off_upd, true, ps ? ps->side : stream);
    if (rc) return rc;
    if (ps) GANQ_HIP_CHECK(hipEventRecord(ps->join, ps->side));
    GANQ_HIP_CHECK(hipMemcpyAsync(Tc, T0, (size_t)m * V * sizeof(float), hipMemcpyDeviceToDevice, stream));
    hipLaunchKernelGGL(best_init_kernel, dim3(1), dim3(1), 0, stream, best, best_k, flag);
    GANQ_LAUNCH_CHECK();

    rc = solve_s_pack_l(L, ldl, m, n, ws + lo.off_solve, stream);  // L is the same in every iteration
    if (rc) return rc;
    // rows whose indices did not change in an iteration are fixed points (same indices -> same bucket sums -> same
    // codebook -> same indices): from the next S-solve on they are skipped
    const int* rowlist = nullptr;
    const int* nactive = nullptr;
    for (int k = 0; k < K; ++k) {
        rc = solve_s_launch(W, L, ldl, Tc, m, n, V, Qwork, nullptr, ws + lo.off_solve, stream, rowlist, nactive,
                            (flags & GANQ_FLAG_NO_HELPERS) == 0);
        if (rc) return rc;
        if (k == 0 && ps) GANQ_HIP_CHECK(hipStreamWaitEvent(stream, ps->join, 0));
        // new codebook and, from the same A and b, the loss of (new codebook, these indices)  (ganq.py:589-591, :621-622)
        // from the second iteration on only the rows that changed are re-solved, in place in the current codebooks (the others are
        // fixed points); the list doubles as the next S-solve's
        const bool in_place = k >= 1 && t_rows_listable(lo.t);
        rc = t_iterate(Qwork, m, n, V, rcond, lo.t, ws + lo.off_upd, nullptr, in_place ? Tc : Tn, nullptr, nullptr, 1, nullptr, k, stream, in_place);
        if (rc) return rc;
        if (k >= 1 && k + 1 < K) {
            rc = t_active_rows(m, lo.t, ws + lo.off_upd, &rowlist, &nactive, stream, in_place);
            if (rc) return rc;
        }
        if (!in_place) std::swap(Tc, Tn);
        // per-iteration records for a caller that takes the best-of-K decision itself (row-sharded layers, tests)
        if (T_all)
            GANQ_HIP_CHECK(hipMemcpyAsync(T_all + (size_t)k * m * V, Tc, (size_t)m * V * sizeof(float), hipMemcpyDeviceToDevice, stream));
        if (loss_rows_all)
            GANQ_HIP_CHECK(hipMemcpyAsync(loss_rows_all + (size_t)k * m, t_loss_rows(lo.t, ws + lo.off_upd), (size_t)m * sizeof(double),
                                          hipMemcpyDeviceToDevice, stream));
        if (Q_all)
            GANQ_HIP_CHECK(hipMemcpyAsync(Q_all + (size_t)k * m * n, Qwork, (size_t)m * n, hipMemcpyDeviceToDevice, stream));
        {
            const int64_t words = m * V;  // fp32 codebook
            const int blocks = (int)std::min<int64_t>(64, std::max<int64_t>(1, (words + 255) / 256));
            hipLaunchKernelGGL(select_copy_kernel, dim3(blocks), dim3(256), 0, stream, t_loss_rows(lo.t, ws + lo.off_upd), (int)m, k,
                               best, best_k, flag, dists, reinterpret_cast<const uint32_t*>(Tc), reinterpret_cast<uint32_t*>(T_best),
                               words);
            GANQ_LAUNCH_CHECK();
        }
        if (!alias) {
            if (k == 0) {  // defined contents even if no iteration ever wins (all distances NaN)
                GANQ_HIP_CHECK(hipMemcpyAsync(Q_out, Qwork, (size_t)m * n, hipMemcpyDeviceToDevice, stream));
            } else {
                rc = launch_copy_if(flag, Qwork, Q_out, (size_t)m * n, stream);
                if (rc) return rc;
            }
        }
    }
    hipLaunchKernelGGL(copy_if_none_kernel, dim3(64), dim3(256), 0, stream, best_k, Tc, T_best, (int64_t)m * V);
    GANQ_LAUNCH_CHECK();
    return 0;
}

extern "C" int ganq_run_layer(const float* W, const float* H, const float* L, int64_t ldl, const float* T0, int64_t m,
                              int64_t n, int V, int K, uint32_t flags, double rcond, float* T_best, uint8_t* Q_out,
                              double* dists, int32_t* best_k, void* workspace, size_t workspace_bytes, void* stream_) {
    return run_layer_impl(W, H, L, ldl, T0, m, n, V, K, flags, rcond, T_best, Q_out, dists, best_k, nullptr, nullptr, nullptr,
                          workspace, workspace_bytes, stream_);
}

// The same fused loop on a SLICE of the rows of a layer (rows are independent in every stage; only the best-of-K decision
// is global), with the per-iteration records the owner of the whole layer needs to take that decision:
// T_all [K,m,V] codebooks, loss_rows_all [K,m] per-row losses, Q_all [K,m,n] indices (each optional).  dists / best_k /
// T_best / Q_out describe the slice alone.  See ganq_select_best for the global decision.
extern "C" int ganq_run_layer_rows(const float* W, const float* H, const float* L, int64_t ldl, const float* T0, int64_t m,
                                   int64_t n, int V, int K, uint32_t flags, double rcond, float* T_best, uint8_t* Q_out,
                                   double* dists, int32_t* best_k, float* T_all, double* loss_rows_all, uint8_t* Q_all,
                                   void* workspace, size_t workspace_bytes, void* stream_) {
    return run_layer_impl(W, H, L, ldl, T0, m, n, V, K, flags, rcond, T_best, Q_out, dists, best_k, T_all, loss_rows_all, Q_all,
                          workspace, workspace_bytes, stream_);
}

namespace ganq {
// dist_k = sum of the K x m per-row losses in the loop's own order (256 strided partial sums, binary tree), best_k =
// first k with the smallest dist (strict <, NaN never wins): bit-identical to the decisions of select_copy_kernel
__global__ __launch_bounds__(256) void select_best_kernel(const double* __restrict__ loss_rows_all, int m, int K,
                                                          double* __restrict__ dists, int32_t* __restrict__ best_k) {
    __shared__ double sh[256];
    double best = INFINITY;
    int bk = -1;
    for (int k = 0; k < K; ++k) {
        double s = 0.0;
        for (int i = threadIdx.x; i < m; i += 256) s += loss_rows_all[(int64_t)k * m + i];
        sh[threadIdx.x] = s;
        __syncthreads();
        for (int w = 128; w > 0; w >>= 1) {
            if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
            __syncthreads();
        }
        const double d = sh[0];
        __syncthreads();
        if (threadIdx.x == 0) dists[k] = d;
        if (d < best) {
            best = d;
            bk = k;
        }
    }
    if (threadIdx.x == 0) *best_k = bk;
}
}  // namespace ganq

// ganq.py:621-626 over the gathered per-row losses of all row slices of a layer: loss_rows_all [K, m] (m = all rows, in
// row order) -> dists [K] fp64, best_k int32 (device).  Same summation order as the single-call loop, so a row-sharded
// layer takes bit-for-bit the decision the unsharded one takes.
extern "C" int ganq_select_best(const double* loss_rows_all, int64_t m, int K, double* dists, int32_t* best_k, void* stream_) {
    if (m < 0 || K < 0) return fail(-1, "ganq_select_best: negative shape");
    if (!dists || !best_k || (!loss_rows_all && m > 0 && K > 0)) return fail(-3, "ganq_select_best: null pointer");
    if (m > INT32_MAX / 2) return fail(-1, "ganq_select_best: m too large");
    hipLaunchKernelGGL(select_best_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream_), loss_rows_all, (int)m, K, dists,
                       best_k);
    GANQ_LAUNCH_CHECK();
    return 0;
}

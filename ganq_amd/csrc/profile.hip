// Per-kernel device timing with HIP events on the caller's stream (see include/ganq_hip.h).
#include <mutex>
#include <vector>

#include "common.h"

namespace ganq {

static const char* const kNames[KID_COUNT] = {"solve_s_kernel", "gemm_f32_kernel", "code_masks_kernel",
                                              "onehot_accum_kernel", "t_solve_kernel", "err_kernel",
                                              "dot_reduce_kernels", "dequant_losses_kernel", "hessian_kernel",
                                              "kmeans_kernels", "lut_gemv_kernel", "lut_gemm_kernel", "pack_kernels", "t_prepare_kernels", "t_incremental_kernels", "cholesky_kernels"};

struct Span {
    int kid;
    hipEvent_t a, b;
};
static std::mutex g_mu;
static bool g_on = false;
static int g_sel = -1;  // -1: every kernel, else only this kernel id (events between dependent launches cost idle time)
static std::vector<Span> g_spans;
static std::vector<hipEvent_t> g_pool;

bool profile_enabled(int kid) { return g_on && (g_sel < 0 || g_sel == kid); }

static hipEvent_t get_event() {
    if (!g_pool.empty()) {
        hipEvent_t e = g_pool.back();
        g_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

void profile_mark(int kid, hipStream_t stream, bool begin) {
    std::lock_guard<std::mutex> lock(g_mu);
    if (begin) {
        Span s{kid, get_event(), nullptr};
        (void)hipEventRecord(s.a, stream);
        g_spans.push_back(s);
    } else {
        for (auto it = g_spans.rbegin(); it != g_spans.rend(); ++it)
            if (it->kid == kid && it->b == nullptr) {
                it->b = get_event();
                (void)hipEventRecord(it->b, stream);
                break;
            }
    }
}

}  // namespace ganq

using namespace ganq;

extern "C" int ganq_profile_enable(int on) {
    std::lock_guard<std::mutex> lock(g_mu);
    g_on = on != 0;
    return 0;
}

extern "C" int ganq_profile_select(int kernel_id) {
    std::lock_guard<std::mutex> lock(g_mu);
    g_sel = (kernel_id >= 0 && kernel_id < KID_COUNT) ? kernel_id : -1;
    return 0;
}

extern "C" int ganq_profile_reset(void) {
    std::lock_guard<std::mutex> lock(g_mu);
    for (auto& s : g_spans) {
        if (s.a) g_pool.push_back(s.a);
        if (s.b) g_pool.push_back(s.b);
    }
    g_spans.clear();
    return 0;
}

extern "C" int ganq_profile_num_kernels(void) { return KID_COUNT; }

extern "C" const char* ganq_profile_kernel_name(int kid) { return (kid >= 0 && kid < KID_COUNT) ? kNames[kid] : ""; }

extern "C" int ganq_profile_get(int kid, double* total_ms, int64_t* launches) {
    std::lock_guard<std::mutex> lock(g_mu);
    double tot = 0.0;
    int64_t cnt = 0;
    for (auto& s : g_spans) {
        if (s.kid != kid || !s.a || !s.b) continue;
        float ms = 0.f;
        hipError_t e = hipEventElapsedTime(&ms, s.a, s.b);
        if (e != hipSuccess) return fail(-100, "ganq_profile_get: %s (synchronise the stream first)", hipGetErrorString(e));
        tot += ms;
        ++cnt;
    }
    if (total_ms) *total_ms = tot;
    if (launches) *launches = cnt;
    return 0;
}

// Process-wide runtime state of libganq_hip.so that is NOT on any hot path:
//   * developer / test switches ("options"): read from the environment ONCE, when the library is loaded, and kept in
//     atomics; tests change them through ganq_debug_set_option() instead of the environment.  The compute entry
//     points read an atomic, never getenv();
//   * the per-(device, kernel) record of the dynamic-LDS limit already raised with hipFuncSetAttribute -- the
//     attribute is per device, and the C-ABI promises re-entrancy per (device, stream), so the record is keyed by
//     hipGetDevice() and guarded by a mutex.
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <utility>

#include "common.h"

namespace ganq {

namespace {

struct OptDesc {
    const char* name;  // option name == environment variable read at load
    long long def;
};

// order == enum Opt in common.h
const OptDesc kOpts[OPT_COUNT] = {
    {"GANQ_T_FULL", 0},            // 1: stateless full bucket accumulation in every iteration
    {"GANQ_T_INCR_THR", -1},       // >= 0: changed-index count above which the device falls back to the full accumulation
    {"GANQ_T_JACOBI", 0},          // 1: every row by the Jacobi eigen-solve (no Cholesky fast path)
    {"GANQ_SOLVE_ALL_ROWS", 0},    // 1: the S-solve never skips converged rows
    {"GANQ_MUPDATE_LDS", 0},       // 1: LDS-histogram version of the incremental update
    {"GANQ_WH_F64", 0},            // 1: W @ H_fixed by the fp64 GEMM instead of the split-fp16 product
    {"GANQ_KMEANS_WCAP", 0},       // > 0: force the windowed k-means kernel with this window
    {"GANQ_CHOL_LOOKAHEAD", 1},    // 0: no second stream in the Cholesky trailing update
    {"GANQ_ACCUM_DEBUG", 0},       // 1: print cycle stamps of onehot_accum_kernel (synchronises)
    {"GANQ_LUT_INWG", -1},         // LUT forward: force (1) / forbid (0) the in-workgroup reduction
    {"GANQ_LUT_WGS", 0},           // LUT forward: workgroup target
    {"GANQ_LUT_KS", 0},            // LUT forward: cap of the split-K factor
    {"GANQ_H_EXT", -1},            // fixed-point H: 1 force / 0 forbid the 16-bit extension word (-1: decided on the device)
    {"GANQ_SOLVE_VARIANT", 0},     // S-solve scheduling variant (developer A/B)
    {"GANQ_LUT_NT", 0},            // LUT decode kernel: force 1 / 2 tiles of 16 features per workgroup (0: by shape)
    {"GANQ_LUT_GEMM_RM", 0},       // LUT GEMM (M > 64): force this split factor of in_features (0: by shape)
    {"GANQ_LUT_GEMM_PIPE", -1},    // LUT GEMM: 1 force / 0 forbid the one-workgroup-per-CU pipelined kernel (-1: by tile count)
    {"GANQ_PREP_OVERLAP", 0},      // fused loop: 1 = the T-update's preparation on a helper stream beside the first S-solve (measured:
                                   // 4096^2 14.13 -> 14.28 ms, 1024x4096 9.90 -> 9.87, 768x3072 6.36 -> 6.29: the S-solve holds 130 KB of
                                   // every CU's LDS, what runs beside it runs on its issue slots; off)
    {"GANQ_SOLVE_DUO", 1},         // S-solve: 0 = never launch helper workgroups (solve_s.hip, "duo"); 2 (tests) = helpers that never answer
    {"GANQ_SOLVE_DUO_XA", 44},     // ... source panels the tile keeps of a chain of c: max(XMIN, XA c / 64 - XB) from c >= CMIN on
    {"GANQ_SOLVE_DUO_XB", 12},
    {"GANQ_SOLVE_DUO_XMIN", 2},
    {"GANQ_SOLVE_DUO_CMIN", 8},
    {"GANQ_SOLVE_TRIO", 1},        // S-solve: 0 = never two helpers per tile
    {"GANQ_SOLVE_TRIO_XA", 12},    // ... the tile's share with two helpers: max(XMIN, XA c / 64 - XB)
    {"GANQ_SOLVE_TRIO_XB", 0},
    {"GANQ_HESS_SPLIT", 1},        // Hessian: 0 = whole-tile kernel only; 1 = cut the tokens of the tiles beyond a multiple of the CU count when tiles < workgroup slots; > 1: at most this many parts per tile
    {"GANQ_HESS_WIDE", 1},         // Hessian, staged groups: 256 x 128 tiles from 3072 in_features on (multiples of 256); 0: never, 2: whenever possible
    {"GANQ_HESS_BULK", -1},        // Hessian, developer: at most this many whole tiles (the others are cut); -1: as many as fill the CUs evenly
    {"GANQ_HESS_PARTS", 0},        // Hessian, developer: cut the tiles behind the bulk into exactly this many parts (0: as many as fill the free slots)
    {"GANQ_GEMM_H16_BM", 0},       // dense GEMM of the LUT forward: force 128- / 256-row tiles (0: by tile count)
    {"GANQ_LUT_DENSE_M", -1},      // LUT forward: from this many rows of x on, dequantise once + dense GEMM (-1: by shape; 0: never)
    {"GANQ_HESS_W4", 1},           // Hessian, staged groups: 0 = never the transposed staging + 256 x 256 stream-K kernel (hessian_w4.hip); 2 = from 1024 in_features on (default: 3072)
    {"GANQ_KMEANS_SPAN", -1},      // windowed k-means: node spacing from which the levels are solved span by span (-1: default; 0: level by level as in rounds 1-3)
};

std::atomic<long long> g_opt[OPT_COUNT];

struct OptInit {
    OptInit() {
        for (int i = 0; i < OPT_COUNT; ++i) {
            long long v = kOpts[i].def;
            const char* e = getenv(kOpts[i].name);
            if (e && *e) v = atoll(e);
            g_opt[i].store(v, std::memory_order_relaxed);
        }
    }
} g_opt_init;  // runs when the shared object is loaded

std::mutex g_attr_mu;
std::map<std::pair<int, const void*>, size_t> g_attr;  // (device, kernel) -> dynamic LDS bytes already allowed

}  // namespace

long long opt_get(int id) { return g_opt[id].load(std::memory_order_relaxed); }

int ensure_dynamic_lds(const void* func, size_t bytes) {
    if (bytes <= 64 * 1024) return 0;  // the default limit
    int dev = 0;
    GANQ_HIP_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(g_attr_mu);
    size_t& have = g_attr[std::make_pair(dev, func)];
    if (bytes > have) {
        GANQ_HIP_CHECK(hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        have = bytes;
    }
    return 0;
}

int current_device_cus() {
    static std::atomic<int> cache[64];  // zero-initialised; 0 = not asked yet
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (dev >= 0 && dev < 64 && (v = cache[dev].load(std::memory_order_relaxed)) > 0) return v;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) v = 0;
    if (dev >= 0 && dev < 64 && v > 0) cache[dev].store(v, std::memory_order_relaxed);
    return v;
}

}  // namespace ganq

using namespace ganq;

extern "C" int ganq_debug_set_option(const char* name, long long value) {
    if (!name) return fail(-3, "ganq_debug_set_option: null name");
    for (int i = 0; i < OPT_COUNT; ++i)
        if (std::strcmp(name, kOpts[i].name) == 0) {
            g_opt[i].store(value, std::memory_order_relaxed);
            return 0;
        }
    return fail(-1, "ganq_debug_set_option: unknown option %s", name);
}

extern "C" int ganq_debug_get_option(const char* name, long long* value) {
    if (!name || !value) return fail(-3, "ganq_debug_get_option: null pointer");
    for (int i = 0; i < OPT_COUNT; ++i)
        if (std::strcmp(name, kOpts[i].name) == 0) {
            *value = g_opt[i].load(std::memory_order_relaxed);
            return 0;
        }
    return fail(-1, "ganq_debug_get_option: unknown option %s", name);
}

extern "C" int ganq_debug_reset_option(const char* name) {
    if (!name) return fail(-3, "ganq_debug_reset_option: null name");
    for (int i = 0; i < OPT_COUNT; ++i)
        if (std::strcmp(name, kOpts[i].name) == 0) {
            g_opt[i].store(kOpts[i].def, std::memory_order_relaxed);
            return 0;
        }
    return fail(-1, "ganq_debug_reset_option: unknown option %s", name);
}

// Shared helpers for libganq_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdarg>
#include <cstdio>

#include "../../include/ganq_hip.h"

namespace ganq {

// thread-local error text returned by ganq_hip_last_error()
char* error_buffer();
int fail(int code, const char* fmt, ...);

#define GANQ_HIP_CHECK(expr)                                                                  \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess)                                                                 \
            return ::ganq::fail(-100, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

#define GANQ_LAUNCH_CHECK() GANQ_HIP_CHECK(hipGetLastError())

__host__ __device__ inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// carve aligned sub-buffers out of a caller-provided workspace
struct Carver {
    char* base;
    size_t off = 0;
    explicit Carver(void* p) : base(static_cast<char*>(p)) {}
    template <typename T>
    T* take(size_t count) {
        off = align_up(off, 256);
        T* p = reinterpret_cast<T*>(base ? base + off : nullptr);
        off += count * sizeof(T);
        return p;
    }
    size_t used() const { return align_up(off, 256); }
};

// 1 if v_mfma_f32_16x16x4_f32 accumulates k = 0,1,2,3 in that order (fma(a3,b3,fma(a2,b2,fma(a1,b1,fma(a0,b0,c))))),
// 0 if k = 3,2,1,0; negative until ganq_hip_selftest has run on the current device / on failure.
int mfma_k_ascending();

// developer / test switches (runtime.hip): read from the environment once at library load, changed by tests through
// ganq_debug_set_option(); the compute path reads an atomic
enum Opt : int {
    OPT_T_FULL = 0,
    OPT_T_INCR_THR,
    OPT_T_JACOBI,
    OPT_SOLVE_ALL_ROWS,
    OPT_MUPDATE_LDS,
    OPT_WH_F64,
    OPT_KMEANS_WCAP,
    OPT_CHOL_LOOKAHEAD,
    OPT_ACCUM_DEBUG,
    OPT_LUT_INWG,
    OPT_LUT_WGS,
    OPT_LUT_KS,
    OPT_H_EXT,
    OPT_SOLVE_VARIANT,
    OPT_LUT_NT,
    OPT_LUT_GEMM_RM,
    OPT_LUT_GEMM_PIPE,
    OPT_PREP_OVERLAP,
    OPT_SOLVE_DUO,
    OPT_SOLVE_DUO_XA,
    OPT_SOLVE_DUO_XB,
    OPT_SOLVE_DUO_XMIN,
    OPT_SOLVE_DUO_CMIN,
    OPT_SOLVE_TRIO,
    OPT_SOLVE_TRIO_XA,
    OPT_SOLVE_TRIO_XB,
    OPT_HESS_SPLIT,
    OPT_HESS_WIDE,
    OPT_HESS_BULK,
    OPT_HESS_PARTS,
    OPT_GEMM_H16_BM,
    OPT_LUT_DENSE_M,
    OPT_HESS_W4,
    OPT_KMEANS_SPAN,
    OPT_COUNT
};
long long opt_get(int id);

// raise a kernel's dynamic-LDS limit above the 64 KB default on the CURRENT device if that has not been done yet
// (recorded per (device, kernel) under a mutex)
int ensure_dynamic_lds(const void* func, size_t bytes);

// dense fp16 / bf16 GEMM of the LUT forward's prefill path (gemm_h16.hip): y[M,N] = x[M,K] @ w[N,K]^T (+ bias) (+ addend)
bool gemm_h16_supported(int64_t M, int64_t N, int64_t K);
int gemm_h16(const void* x, const void* w, const void* bias, const float* addend, int dtype, int64_t M, int64_t N, int64_t K, void* y,
             hipStream_t stream);

// Hessian of staged groups from TRANSPOSED activations Xt [in_features][ldt tokens], 256 x 256 tiles cut stream-K (hessian_w4.hip)
const uint32_t* hessian_tile_table(int tiles, int* count);  // device table of the lower-triangular tile pairs (tu << 16 | tv), Z order
bool hessian_w4_supported(int64_t n, int64_t ldt);
size_t hessian_w4_workspace_bytes(int64_t n);
int hessian_w4_stage(void* Xt, int64_t ldt, const void* X, int64_t rows, int64_t n, int64_t tok0, hipStream_t stream);
int hessian_w4(float* H, const void* Xt, int64_t ldt, int dtype, int64_t rows, int64_t n, float decay, float scale, void* workspace,
               size_t workspace_bytes, hipStream_t stream);

// compute units of the CURRENT device (cached per device index; 0 on failure)
int current_device_cus();

// ---- optional per-kernel timing (profile.hip) ----
enum KernelId : int {
    KID_SOLVE_S = 0,
    KID_GEMM_F32,
    KID_SORT_CODES,
    KID_SHT_ACCUM,
    KID_T_SOLVE,
    KID_ERR,
    KID_DOT,
    KID_DEQUANT,
    KID_HESSIAN,
    KID_KMEANS,
    KID_LUT_GEMV,
    KID_LUT_GEMM,
    KID_PACK,
    KID_T_PREP,
    KID_T_INCR,
    KID_CHOLESKY,
    KID_COUNT
};
bool profile_enabled(int kid);
void profile_mark(int kid, hipStream_t stream, bool begin);
struct ProfScope {
    int kid;
    hipStream_t stream;
    bool on;
    ProfScope(int k, hipStream_t s) : kid(k), stream(s), on(profile_enabled(k)) {
        if (on) profile_mark(kid, stream, true);
    }
    ~ProfScope() {
        if (on) profile_mark(kid, stream, false);
    }
};

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;

}  // namespace ganq

// Error plumbing + the MFMA accumulation-order probe.
//
// The S-solve's contract (include/ganq_hip.h, oracle/ganq_oracle.c) is one fp32 fused-multiply-add
// chain per residual, in descending column order.  Part of that chain runs on
// v_mfma_f32_16x16x4_f32, whose internal order over its 4 k-slices is a hardware property: the
// probe measures it once per process and solve_s maps "largest column first" onto it.
#include <cmath>
#include <cstring>
#include <mutex>

#include "common.h"

namespace ganq {

static thread_local char g_err[512] = {0};
char* error_buffer() { return g_err; }

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

// per device (the probe runs once on each device the library is used on; all entries start at "not run")
constexpr int MAX_DEVICES = 64;
static int g_k_ascending[MAX_DEVICES];
static bool g_probe_init = [] {
    for (int& v : g_k_ascending) v = -1;
    return true;
}();
static std::mutex g_probe_mutex;
static int current_device_slot() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) return -1;
    return dev;
}
int mfma_k_ascending() {
    const int dev = current_device_slot();
    if (dev < 0) return -1;
    std::lock_guard<std::mutex> lock(g_probe_mutex);
    return g_k_ascending[dev];
}

__global__ void probe_mfma_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                  const float* __restrict__ C, float* __restrict__ D) {
    // A [16][4], B [4][16], C/D [16][16]
    const int l = threadIdx.x;
    float a = A[(l & 15) * 4 + (l >> 4)];
    float b = B[(l >> 4) * 16 + (l & 15)];
    f32x4 c;
    for (int r = 0; r < 4; ++r) c[r] = C[((l >> 4) * 4 + r) * 16 + (l & 15)];
    f32x4 d = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[((l >> 4) * 4 + r) * 16 + (l & 15)] = d[r];
}

// Brute-force check of the division shortcut used by the S-solve: with rinv = RN(1/b),
//     q0 = RN(a * rinv);  e = fma(-q0, b, a);  q = fma(e, rinv, q0)
// must equal the IEEE quotient RN(a / b) (Markstein).  Counts disagreements over pseudo-random operands drawn from
// the ranges the solve sees (|a| small residuals incl. tiny values, b = Cholesky diagonal > 0).
__global__ __launch_bounds__(256) void div_check_kernel(uint64_t count, uint32_t seed, unsigned long long* mismatches,
                                                        float* first_bad) {
    unsigned long long bad = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (uint64_t)gridDim.x * 256) {
        uint64_t x = (i + 1) * 0x9E3779B97F4A7C15ull + seed;
        x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32; x *= 0x94D049BB133111EBull; x ^= x >> 29;
        const uint32_t ma = (uint32_t)x & 0x007fffffu, mb = (uint32_t)(x >> 23) & 0x007fffffu;
        const uint32_t ea = 127u - 40u + (uint32_t)((x >> 46) % 48u);   // 2^-40 .. 2^7
        const uint32_t eb = 127u - 12u + (uint32_t)((x >> 52) % 24u);   // 2^-12 .. 2^11
        const uint32_t sa = (uint32_t)(x >> 63) << 31;
        const float a = __builtin_bit_cast(float, sa | (ea << 23) | ma);
        const float b = __builtin_bit_cast(float, (eb << 23) | mb);
        const float rinv = 1.0f / b;
        const float q0 = a * rinv;
        const float e = fmaf(-q0, b, a);
        const float q = fmaf(e, rinv, q0);
        const float ref = a / b;
        if (__builtin_bit_cast(uint32_t, q) != __builtin_bit_cast(uint32_t, ref)) {
            if (bad == 0 && atomicAdd(mismatches, 0ull) == 0) {
                first_bad[0] = a;
                first_bad[1] = b;
            }
            ++bad;
        }
    }
    if (bad) atomicAdd(mismatches, bad);
}

}  // namespace ganq

using namespace ganq;

extern "C" int ganq_debug_div_check(uint64_t count, uint32_t seed, unsigned long long* mismatches_dev, float* first_bad_dev,
                                    void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    GANQ_HIP_CHECK(hipMemsetAsync(mismatches_dev, 0, sizeof(unsigned long long), stream));
    hipLaunchKernelGGL(div_check_kernel, dim3(4096), dim3(256), 0, stream, count, seed, mismatches_dev, first_bad_dev);
    GANQ_LAUNCH_CHECK();
    return 0;
}

extern "C" int ganq_hip_version(void) { return GANQ_HIP_ABI_VERSION; }

extern "C" const char* ganq_hip_last_error(void) { return error_buffer(); }

extern "C" int ganq_hip_selftest(void* stream_) {
    const int dev = current_device_slot();
    if (dev < 0) return fail(-100, "ganq_hip_selftest: no current HIP device (or device index >= %d)", MAX_DEVICES);
    std::lock_guard<std::mutex> lock(g_probe_mutex);
    if (g_k_ascending[dev] >= 0) return 0;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    float hA[64], hB[64], hC[256], hD[256];
    uint32_t s = 12345u;
    auto rnd = [&]() {
        s = s * 1664525u + 1013904223u;
        return (float)((s >> 8) & 0xffff) / 65536.0f * 2.0f - 1.0f + (float)(s >> 24) * 1e-7f;
    };
    for (auto& v : hA) v = rnd();
    for (auto& v : hB) v = rnd();
    for (auto& v : hC) v = rnd();
    float* dbuf = nullptr;
    GANQ_HIP_CHECK(hipMalloc(&dbuf, sizeof(float) * (64 + 64 + 256 + 256)));
    float *dA = dbuf, *dB = dbuf + 64, *dC = dbuf + 128, *dD = dbuf + 384;
    hipError_t e = hipMemcpyAsync(dA, hA, sizeof(hA), hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) e = hipMemcpyAsync(dB, hB, sizeof(hB), hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) e = hipMemcpyAsync(dC, hC, sizeof(hC), hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(probe_mfma_kernel, dim3(1), dim3(64), 0, stream, dA, dB, dC, dD);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(hD, dD, sizeof(hD), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    (void)hipFree(dbuf);
    if (e != hipSuccess) return fail(-100, "mfma probe failed: %s", hipGetErrorString(e));
    int asc = 0, desc = 0, differ = 0;
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            float up = hC[i * 16 + j], down = hC[i * 16 + j];
            for (int k = 0; k < 4; ++k) up = fmaf(hA[i * 4 + k], hB[k * 16 + j], up);
            for (int k = 3; k >= 0; --k) down = fmaf(hA[i * 4 + k], hB[k * 16 + j], down);
            float d = hD[i * 16 + j];
            if (up != down) ++differ;
            if (std::memcmp(&d, &up, 4) == 0) ++asc;
            if (std::memcmp(&d, &down, 4) == 0) ++desc;
        }
    if (differ < 16) return fail(-101, "mfma probe data not order-sensitive (%d)", differ);
    if (asc == 256 && desc < 256) {
        g_k_ascending[dev] = 1;
    } else if (desc == 256 && asc < 256) {
        g_k_ascending[dev] = 0;
    } else {
        return fail(-102, "v_mfma_f32_16x16x4_f32 is not an ordered fmaf chain on this device (asc %d desc %d of 256)",
                    asc, desc);
    }
    return 0;
}

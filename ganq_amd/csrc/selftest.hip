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

static int g_k_ascending = -1;
static std::mutex g_probe_mutex;
int mfma_k_ascending() { return g_k_ascending; }

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

}  // namespace ganq

using namespace ganq;

extern "C" int ganq_hip_version(void) { return GANQ_HIP_ABI_VERSION; }

extern "C" const char* ganq_hip_last_error(void) { return error_buffer(); }

extern "C" int ganq_hip_selftest(void* stream_) {
    std::lock_guard<std::mutex> lock(g_probe_mutex);
    if (g_k_ascending >= 0) return 0;
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
        g_k_ascending = 1;
    } else if (desc == 256 && asc < 256) {
        g_k_ascending = 0;
    } else {
        return fail(-102, "v_mfma_f32_16x16x4_f32 is not an ordered fmaf chain on this device (asc %d desc %d of 256)",
                    asc, desc);
    }
    return 0;
}

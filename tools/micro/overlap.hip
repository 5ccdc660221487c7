// microbenchmark: does VALU / DPP work of one wave overlap with a dependent v_mfma_f32_16x16x4_f32 chain of ANOTHER wave
// on the same SIMD?  Workgroup = 8 waves (2 per SIMD): waves 0-3 run a dependent chain of `PKIND` vector instructions,
// waves 4-7 a dependent MFMA chain.  Times: P alone, G alone, both.
//   PKIND 0: plain VALU (v_add_u32), 1: DPP row reductions (v_min_u32_dpp), 2: v_permlane-free LDS-free mix (fma),
//   3: ds_swizzle, 4: DPP mov only
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int CTRL>
__device__ __forceinline__ unsigned dpp_u(unsigned x) {
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, 0xf, 0xf, true);
}

template <int PKIND, bool RUNP, bool RUNG>
__global__ __launch_bounds__(512) void k(float* out, int iters, float a, float b, unsigned seed) {
    const int wv = threadIdx.x >> 6;
    float res = 0.f;
    if (wv < 4) {
        if (RUNP) {
            unsigned x = seed + threadIdx.x;
            float f = (float)x;
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    if (PKIND == 0) { x = x * 3u + 1u; x ^= x >> 3; }
                    else if (PKIND == 1) { x = min(x + 7u, dpp_u<0xB1>(x)); x = min(x ^ 5u, dpp_u<0x4E>(x)); }
                    else if (PKIND == 2) { f = __builtin_fmaf(f, 1.0001f, 0.5f); f = __builtin_fmaf(f, 0.9999f, 0.25f); }
                    else if (PKIND == 3) { x = (unsigned)__builtin_amdgcn_ds_swizzle((int)x + 7, 0x041F) + 1u; x = (unsigned)__builtin_amdgcn_ds_swizzle((int)x ^ 5, 0x081F); }
                    else { x = dpp_u<0xB1>(x) + 7u; x = dpp_u<0x4E>(x) ^ 5u; }
                }
            }
            res = (float)x + f;
        }
    } else if (RUNG) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
        }
        res = acc[0] + acc[1] + acc[2] + acc[3];
    }
    out[blockIdx.x * 512 + threadIdx.x] = res;
}

template <int PKIND, bool RUNP, bool RUNG>
float run(int iters) {
    float* out;
    hipMalloc(&out, 256 * 512 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<PKIND, RUNP, RUNG>), dim3(256), dim3(512), 0, 0, out, iters, 1.0f, 1e-9f, 3u);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<PKIND, RUNP, RUNG>), dim3(256), dim3(512), 0, 0, out, iters, 1.0f, 1e-9f, 3u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipFree(out);
    return ms;
}
template <int PKIND> void report(const char* name) {
    const int iters = 20000;  // P: 32 dependent instr per iter; G: 4 MFMA (128 cycles at 32/MFMA) per iter
    float p = run<PKIND, true, false>(iters), g = run<PKIND, false, true>(iters), both = run<PKIND, true, true>(iters);
    printf("%-22s P alone %.3f ms (%.1f cyc/instr @2.4GHz)  G alone %.3f ms (%.1f cyc/MFMA)  both %.3f ms  -> overlap %.0f%% of min(P,G)\n", name, p,
           p * 2.4e6 / (iters * 32.0), g, g * 2.4e6 / (iters * 4.0), both, 100.0 * (p + g - both) / (p < g ? p : g));
}
int main() {
    report<0>("plain VALU int");
    report<2>("plain VALU fma");
    report<1>("DPP min (fused)");
    report<4>("DPP mov + VALU");
    report<3>("ds_swizzle");
    return 0;
}

// microbenchmark: dependent chain of v_mfma_f32_16x16x4_f32 (1 wave per SIMD, every CU busy)
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;
template <int NACC>
__global__ __launch_bounds__(256) void chain(float* out, int iters, float a, float b, long long* cyc) {
    f32x4 acc[NACC];
    for (int k = 0; k < NACC; ++k) acc[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int k = 0; k < NACC; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[k], 0, 0, 0);
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int k = 0; k < NACC; ++k) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int NACC> void run(int blocks) {
    float* out; long long* cyc;
    hipMalloc(&out, blocks * 256 * 4); hipMalloc(&cyc, blocks * 8);
    int iters = 2048;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(chain<NACC>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 1e-9f, cyc);
    hipEventRecord(e0);
    hipLaunchKernelGGL(chain<NACC>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 1e-9f, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c0; hipMemcpy(&c0, cyc, 8, hipMemcpyDeviceToHost);
    double n = (double)iters * 16 * NACC;
    printf("blocks %d NACC %d: %.3f ms, %.1f ns per MFMA per wave, memtime ticks/MFMA %.1f\n", blocks, NACC, ms, ms * 1e6 / n, c0 / n);
}
int main() { run<1>(256); run<2>(256); run<4>(256); run<1>(1); return 0; }

// developer: what the matrix pipes sustain with nothing else going on -- the ceiling the GEMM kernels are priced against in practice
// (the 2.5 PFLOP/s figure assumes the peak clock; under a chip-wide matrix load the clock is whatever the power budget leaves).
// hipcc --offload-arch=gfx950 -O3 tools/dev/mfma_peak.hip -o build_variants/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void spin(float* out, int iters, unsigned long long* clk) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{(float)i, 0.f, 0.f, (float)threadIdx.x};
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(1.0f / (1 + i)); }
    const unsigned long long t0 = __builtin_readcyclecounter();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b));
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 4000;
    float* out; unsigned long long* clk;
    hipMalloc(&out, 4096 * 512 * 4); hipMalloc(&clk, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wgs_per_cu = 1; wgs_per_cu <= 2; ++wgs_per_cu) {
        for (int ncu : {32, 256}) {
            const int grid = ncu * wgs_per_cu;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(spin<16>, dim3(grid), dim3(256), 0, 0, out, iters, clk);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
                const double flops = (double)grid * 4 * iters * 16 * 16384.0;
                printf("grid %4d (waves/SIMD %d): %.3f ms  %.1f TFLOP/s  | wave 0: %llu cycles (s_memtime), %llu ticks of 100 MHz -> counter runs at %.2f GHz; "
                       "%.2f counter cycles per matrix instruction\n", grid, wgs_per_cu, ms, flops / ms * 1e-9, h[0], h[1],
                       h[0] / (h[1] * 10.0) , (double)h[0] / ((double)iters * 16));
            }
        }
    }
    return 0;
}

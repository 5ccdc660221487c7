// microbenchmark: dependent v_mfma_f32_16x16x4_f32 chain fed from LDS, one wave per SIMD (4 waves per workgroup, 1 per CU)
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, long long* cyc) {
    __shared__ float4 lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = make_float4(1e-9f * i, 1.f, 2.f, 3.f);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const float* f = reinterpret_cast<const float*>(lds);
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {  // 2 ds_read_b32 per MFMA, reads issued one 16-group ahead (software pipelined by hand)
            float a[16], b[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) { a[i] = f[(it & 7) * 1024 + i * 64 + lane]; b[i] = f[8192 + (it & 7) * 1024 + i * 64 + lane]; }
#pragma unroll
            for (int i = 0; i < 16; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[i], acc, 0, 0, 0);
        } else if (MODE == 1) {  // 2 ds_read_b128 per 4 MFMA
            float4 a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { a[i] = lds[(it & 7) * 256 + i * 64 + lane]; b[i] = lds[2048 + (it & 7) * 256 + i * 64 + lane]; }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].x, b[i].x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].y, b[i].y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].z, b[i].z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].w, b[i].w, acc, 0, 0, 0);
            }
        } else if (MODE == 2) {  // as MODE 1 plus a workgroup barrier per 16 MFMA
            float4 a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { a[i] = lds[(it & 7) * 256 + i * 64 + lane]; b[i] = lds[2048 + (it & 7) * 256 + i * 64 + lane]; }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].x, b[i].x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].y, b[i].y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].z, b[i].z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].w, b[i].w, acc, 0, 0, 0);
            }
            __syncthreads();
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE> void run() {
    float* out; long long* cyc;
    (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&cyc, 256 * 8);
    int iters = 4096;
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, out, iters, cyc);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, out, iters, cyc);
    (void)hipDeviceSynchronize();
    long long c0; (void)hipMemcpy(&c0, cyc, 8, hipMemcpyDeviceToHost);
    printf("mode %d: %.1f cycles per MFMA\n", MODE, (double)c0 / (iters * 16.0));
}
int main() { run<0>(); run<1>(); run<2>(); return 0; }

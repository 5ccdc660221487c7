// developer: what the matrix pipes sustain with nothing else going on -- the ceiling the GEMM kernels are priced against in practice
// (the 2.5 PFLOP/s figure assumes the peak clock; under a chip-wide matrix load the clock is whatever the power budget leaves, and
// that depends on the operand bits: constant operands toggle nothing).  Two instruction shapes, constant and random operands.
// hipcc --offload-arch=gfx950 -O3 tools/dev/mfma_peak.hip -o build_variants/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

__device__ inline unsigned int hash32(unsigned int x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
// fp16 bit patterns of modest magnitude: sign + exponent 12..15 + random mantissa
__device__ inline unsigned int rnd_half2(unsigned int h) {
    const unsigned int lo = (h & 0x83ffu) | (((h >> 10) & 3u) + 12u) << 10;
    const unsigned int hi = ((h >> 16) & 0x83ffu) | (((h >> 26) & 3u) + 12u) << 10;
    return lo | hi << 16;
}

template <int SHAPE, bool RANDOM>  // SHAPE 0: 16x16x32 (64 tiles of 4 regs), 1: 32x32x16 (16 tiles of 16 regs)
__global__ __launch_bounds__(256, 1) void spin(float* out, int iters, unsigned long long* clk) {
    u32x4 fa[8], fb[8];
    for (int i = 0; i < 8; ++i)
        for (int k = 0; k < 4; ++k) {
            const unsigned int ha = hash32(threadIdx.x * 64 + i * 4 + k + 1), hb = hash32(0x9e3779b9u + threadIdx.x * 64 + i * 4 + k);
            fa[i][k] = RANDOM ? rnd_half2(ha) : 0x3c003c00u;
            fb[i][k] = RANDOM ? rnd_half2(hb) : 0x3c003c00u;
        }
    const unsigned long long t0 = __builtin_readcyclecounter();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    if constexpr (SHAPE == 0) {
        f32x4 acc[8][8];
        for (int i = 0; i < 8; ++i)
            for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{(float)i, 0.f, (float)j, (float)threadIdx.x};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(fb[j]), "v"(fa[i]));
        }
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
        for (int i = 0; i < 8; ++i)
            for (int j = 0; j < 8; ++j) s += acc[i][j][0] + acc[i][j][3];
    } else if constexpr (SHAPE == 2) {  // fp32 16x16x4 (the S-solve's instruction): 64 tiles, 2048 flop each
        f32x4 acc[8][8];
        for (int i = 0; i < 8; ++i)
            for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{(float)i, 0.f, (float)j, (float)threadIdx.x};
        float ga[8], gb[8];
        for (int i = 0; i < 8; ++i) {
            ga[i] = RANDOM ? __builtin_bit_cast(float, (fa[i][0] & 0x807fffffu) | 0x3f000000u) : 1.0f;
            gb[i] = RANDOM ? __builtin_bit_cast(float, (fb[i][0] & 0x807fffffu) | 0x3f000000u) : 1.0f;
        }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(gb[j]), "v"(ga[i]));
        }
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
        for (int i = 0; i < 8; ++i)
            for (int j = 0; j < 8; ++j) s += acc[i][j][0] + acc[i][j][3];
    } else if constexpr (SHAPE == 3) {  // int8 32x32x32 (the T-update's instruction): 16 tiles, 65536 op each
        i32x16 acc[4][4];
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j)
                for (int k = 0; k < 16; ++k) acc[i][j][k] = i + j + k;
        u32x4 ia[8], ib[8];
        for (int i = 0; i < 8; ++i)
            for (int k = 0; k < 4; ++k) {
                ia[i][k] = RANDOM ? (hash32(threadIdx.x * 64 + i * 4 + k + 7) & 0x01010101u) : 0x01010101u;  // one-hot-like bytes 0 / 1
                ib[i][k] = RANDOM ? hash32(0x51ed270bu + threadIdx.x * 64 + i * 4 + k) : 0x01010101u;        // digit bytes
            }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(ib[j + 4 * kk]), "v"(ia[i + 4 * kk]));
        }
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) s += (float)(acc[i][j][0] + acc[i][j][7]);
    } else {
        f32x16 acc[4][4];
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j)
                for (int k = 0; k < 16; ++k) acc[i][j][k] = (float)(i + j + k) + (float)threadIdx.x;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(fb[j + 4 * kk]), "v"(fa[i + 4 * kk]));
        }
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][7];
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int SHAPE, bool RANDOM>
static void run(const char* tag, float* out, unsigned long long* clk, int iters, double ops_per_iter = 64 * 16384.0) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int grid : {32, 256}) {
        float best = 1e30f; unsigned long long h[2] = {0, 0};
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL((spin<SHAPE, RANDOM>), dim3(grid), dim3(256), 0, 0, out, iters, clk);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep && ms < best) { best = ms; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost); }
        }
        const double flops = (double)grid * 4 * iters * ops_per_iter;  // (fp16 shapes: 1 Mflop per wave per iteration)
        printf("%-40s grid %3d: %.3f ms  %7.1f Top/s  clock %.2f GHz, %.2f cycles per 16384 op\n", tag, grid, best, flops / best * 1e-9,
               h[0] / (h[1] * 10.0), (double)h[0] / ((double)iters * ops_per_iter / 16384.0));
    }
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 4000;
    float* out; unsigned long long* clk;
    if (hipMalloc(&out, 4096 * 512 * 4) != hipSuccess || hipMalloc(&clk, 16) != hipSuccess) return 1;
    run<0, false>("16x16x32 f16, constant operands", out, clk, iters);
    run<0, true>("16x16x32 f16, random operands", out, clk, iters);
    run<1, false>("32x32x16 f16, constant operands", out, clk, iters);
    run<1, true>("32x32x16 f16, random operands", out, clk, iters);
    run<2, false>("16x16x4 f32, constant operands", out, clk, iters, 64 * 2048.0);
    run<2, true>("16x16x4 f32, random operands", out, clk, iters, 64 * 2048.0);
    run<3, false>("32x32x32 i8, constant operands", out, clk, iters / 2, 32 * 65536.0);
    run<3, true>("32x32x32 i8, one-hot x random digits", out, clk, iters / 2, 32 * 65536.0);
    return 0;
}

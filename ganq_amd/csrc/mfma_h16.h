// Pieces shared by the hand-scheduled fp16 / bf16 matrix kernels (gemm_h16.hip, hessian_w4.hip): 16-byte fragment types, the address-space
// pointer types of the LDS-DMA builtin, and v_mfma_f32_16x16x32_{f16,bf16} with the accumulator pinned to the accumulation registers.
#pragma once
#include "common.h"

namespace ganq {

typedef float hg_f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t hg_u32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) void* hg_gptr;
typedef __attribute__((address_space(3))) void* hg_lptr;

// The matrix instruction with its accumulator PINNED to the accumulation registers ("+a"): with 256 of them per wave (one wave per SIMD) hipcc
// otherwise shuttles accumulator tiles between the two register files inside the K loop (1477 v_accvgpr moves and 242 spills in the
// first build of the four-wave GEMM).  An asm statement is not reordered against other volatile asm, so the interleave written is the one issued.
template <bool BF16>
__device__ __forceinline__ void hg_mfma_acc(hg_f32x4& acc, const hg_u32x4& a, const hg_u32x4& b) {
    if constexpr (BF16) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}

// acc = 0, born in an accumulation register: a matrix instruction with the inline constant 0 as C and zero operands (a vector-register
// zero copied in would make the loop's phi a vector register again)
template <bool BF16>
__device__ __forceinline__ void hg_mfma_zero(hg_f32x4& acc, const hg_u32x4& z) {
    // (s_nop: the zero operand was written by a vector move right in front, and hipcc's hazard recogniser does not see a matrix
    // instruction in an asm statement -- without the wait the first tile starts from the register's previous contents)
    if constexpr (BF16) asm volatile("s_nop 4\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %1, 0" : "=a"(acc) : "v"(z));
    else asm volatile("s_nop 4\n\tv_mfma_f32_16x16x32_f16 %0, %1, %1, 0" : "=a"(acc) : "v"(z));
}

}  // namespace ganq

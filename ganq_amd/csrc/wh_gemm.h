// Internal interface of the split-fp16 W @ H_fixed product (wh_gemm.hip), used by t_prepare (update_t.hip).
#pragma once
#include "common.h"

namespace ganq {

struct WhLayout {
    int64_t KT, tiles_m, tiles_n;                      // 32-deep k tiles, 128-row blocks of W and of H
    size_t wp_bytes, hp_bytes, rexp_bytes, wlo_bytes;  // packed fp16 pieces of W and H, row exponents, Wlo flags
};

WhLayout wh_layout(int64_t m, int64_t n);

// WH[m,n] (fp64) = W[m,n] (fp32) @ H_fixed, H_fixed = hscale * (Hint + Jint / 65536) symmetric (Jint used when *ext != 0),
// dexp[n] = per-feature exponents of the symmetric scaling (floor(log2 H_uu / 2)), hdiag64[n] = diag(H_fixed) (added
// exactly, outside the matrix product); all pointers device memory
int wh_gemm(const float* W, const int* Hint, const short* Jint, const int* ext, const int* dexp, const double* hdiag64,
            const double* hscale, int64_t m, int64_t n, const WhLayout& lo, char* wp, char* hp, int* rexp, int* wlo_any, double* WH,
            hipStream_t stream);

}  // namespace ganq

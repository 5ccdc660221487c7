// Internal interface of the T-update (update_t.hip) used by the loop driver (run_layer.hip).
#pragma once
#include "common.h"

namespace ganq {

struct TLayout {
    int64_t nq, ng;  // padded plane row pitch, number of 64-column mask groups
    size_t off_prep, off_planes, off_hdiag, off_bits, off_mpart, off_h64, off_wh64, off_whw, off_lossrows, total;
    // incremental bucket sums (loop driver only): integer H, previous indices, per-row sums, change lists
    size_t off_hint, off_qprev, off_mstate, off_chg, off_chgcnt;
    // split-fp16 W @ H_fixed (wh_gemm.hip): packed pieces of W and H, row exponents, Wlo flags
    size_t off_wp, off_hp, off_rexp, off_wlo;
    size_t off_active;  // rows that changed in the last iteration (S-solve work list)
    // extension word of the fixed-point H (planes 4..5 live behind planes 0..3): its diagonal, its partial / kept bucket
    // sums, its int16 copy; per-feature exponents of the symmetric scaling used by the W @ H product
    size_t off_hdiag_j, off_mpart_lo, off_mstate_lo, off_jint, off_dexp, off_hdiag64;
};

TLayout t_layout(int64_t m, int64_t n, bool with_f64);

// once per layer.  with_f64: also W @ H_fixed (fp64 values; computed on the fp16 matrix cores from split operands,
// wh_gemm.hip, or by the fp64 GEMM with GANQ_WH_F64=1) and w_i^T H w_i (needed for the closed-form loss)
int t_prepare(const float* W, const float* H, int64_t m, int64_t n, const TLayout& lo, char* ws, bool with_f64,
              hipStream_t stream);

// once per iteration.  WH32 != nullptr: b from the caller's fp32 W@H (stage API, no loss);
// WH32 == nullptr: b from the product prepared by t_prepare; loss_mode 0: no loss, 1: per-row losses only
// (t_loss_rows; the caller reduces them), 2: also their sum in loss_out (device double).
// iter < 0: stateless (full accumulation).  iter >= 0 (layout built with with_f64): the bucket sums are kept between
// calls; iter == 0 accumulates them in full, later calls only move the H entries of the indices that changed
// (exact: the sums are integers), falling back to the full accumulation on the device when too many changed.
int t_iterate(const uint8_t* Q, int64_t m, int64_t n, int V, double rcond, const TLayout& lo, char* ws, const float* WH32,
              float* T_out, float* A_out, float* b_out, int loss_mode, double* loss_out, int iter, hipStream_t stream,
              bool changed_rows_only = false);
// changed_rows_only (iter >= 1): T_out holds the PREVIOUS codebooks and only the rows whose indices changed in this iteration are
// solved and overwritten (the others are fixed points: same sums, same codebook, same loss); whether the layout / the test switches
// allow that:
bool t_rows_listable(const TLayout& lo);

// after t_iterate(iter >= 1): rows whose indices changed in that iteration (device list + count; both null when
// there are no change lists).  Unchanged rows are fixed points of the alternation and need no further S-solve.
int t_active_rows(int64_t m, const TLayout& lo, char* ws, const int** list, const int** count, hipStream_t stream,
                  bool already_built = false);  // already_built: t_iterate(changed_rows_only) of this iteration made the list

inline const double* t_loss_rows(const TLayout& lo, const char* ws) { return reinterpret_cast<const double*>(ws + lo.off_lossrows); }

}  // namespace ganq

// Internal interface of the S-solve (solve_s.hip) used by the loop driver (run_layer.hip).
#pragma once
#include "common.h"

namespace ganq {

// packed copy of L into the workspace of ganq_solve_s_workspace_bytes() -- depends on L only: once per layer
int solve_s_pack_l(const float* L, int64_t ldl, int64_t m, int64_t n, void* workspace, hipStream_t stream);

// the solve proper; the workspace already holds the packed L
int solve_s_launch(const float* W, const float* L, int64_t ldl, const float* T, int64_t m, int64_t n, int V, uint8_t* Q_out,
                   float* Err_out, void* workspace, hipStream_t stream, const int* rowlist = nullptr,
                   const int* nactive = nullptr,   // rowlist / nactive (device): solve only these rows
                   bool allow_helpers = true);     // false: never launch helper workgroups (the caller knows the launch is not
                                                   // alone on the device: GANQ_FLAG_NO_HELPERS)

}  // namespace ganq

/*
 * ganq_hip.h -- C-ABI of libganq_hip.so: the MI355X (gfx950) implementation of GANQ's per-layer
 * alternating optimisation and of the LUT-dequant linear forward.
 *
 * This is the drop-in boundary for ONE path of smpanaro/ganq (a GPTQModel fork): what
 * `GANQ._perform_quantization_loop` (gptqmodel/quantization/ganq.py:456-646), the Hessian
 * accumulation it inherits (gptqmodel/quantization/gptq.py:88-131) and
 * `FakeQuantLinear.forward` (gptqmodel/nn_modules/qlinear/fake.py:88-89) compute.  The
 * reference has no FFI for this path (its only native code is Metal source embedded in
 * ganq.py:39-328); each entry point below names the reference lines it replaces, and
 * INTEGRATION.md shows the ctypes binding a maintainer would add to ganq.py.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (hipMalloc / torch tensor.data_ptr()) unless marked host;
 *     buffers are borrowed for the duration of the call only, outputs are caller-allocated;
 *   - matrices are row-major and contiguous unless a leading dimension is given;
 *   - m = out_features (rows of W), n = in_features (columns), V = 2^bits codebook entries;
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream);
 *     calls are asynchronous on that stream and re-entrant per stream;
 *   - return value 0 = ok; <0 = error, text via ganq_hip_last_error() (thread-local).
 *   - Q is one uint8 per index ([m,n]); bits in {2,3,4} (V <= 16) are implemented.
 */
#ifndef GANQ_HIP_H
#define GANQ_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GANQ_HIP_ABI_VERSION 4

/* flags for ganq_run_layer */
#define GANQ_FLAG_ALIAS_Q 1u /* reference torch-branch behaviour: indices of the LAST iteration are returned with \
                                the codebook of the BEST one (ganq.py:487,550,625-626) */
#define GANQ_FLAG_NO_HELPERS 2u /* the S-solve launches no helper workgroups: for callers that run several layers side by \
                                   side on one device (streams), where a helper may not be resident with its tile */

int ganq_hip_version(void);
const char* ganq_hip_last_error(void);

/* Device self-test: determines the accumulation order of v_mfma_f32_16x16x4_f32 on this GPU and
 * checks it against an fmaf chain (the S-solve's bit-exactness contract depends on it).
 * Returns 0 and caches the result; the compute entry points call it lazily. */
int ganq_hip_selftest(void* stream);

/* Developer check (tests): counts operand pairs for which the S-solve's reciprocal-based quotient differs from the IEEE
 * division; mismatches_dev: one uint64 on the device, first_bad_dev: two floats (a, b) of one offending pair. */
int ganq_debug_div_check(uint64_t count, uint32_t seed, unsigned long long* mismatches_dev, float* first_bad_dev,
                         void* stream);

/* Developer check (tests): WH_out [m,n] fp64 = the W @ H_fixed product the fused driver feeds to the T-update (fp16 matrix
 * cores, operands split into two fp16 pieces; option GANQ_WH_F64=1: the fp64 GEMM), Hfixed_out [n,n] fp64 (may be NULL) = the
 * 31-bit fixed-point H it is formed with.  Allocates its own scratch and synchronises the stream. */
int ganq_debug_wh_product(const float* W, const float* H, int64_t m, int64_t n, double* WH_out, double* Hfixed_out, void* stream);

/* Developer / tests: the dense fp16 / bf16 GEMM the LUT forward uses from ~1024 rows of x on (csrc/gemm_h16.hip), on its own:
 * y [M,N] = x [M,K] @ w [N,K]^T (+ bias [N]) (+ addend [M,N] fp32), dtype 0 = fp16, 1 = bf16; K a multiple of 64, N of 4. */
int ganq_debug_gemm_h16(const void* x, const void* w, const void* bias, const float* addend, int dtype, int64_t M, int64_t N,
                        int64_t K, void* y, void* stream);

/* ---- a1: Hessian accumulation (gptq.py:96-131 process_batch) --------------------------------
 * One calibration batch: X [rows, n] fp16 or bf16 (dtype: 0 = fp16, 1 = bf16), `batch` = number
 * of sequences in it (gptq.py:104), nsamples_before = sequences accumulated so far.
 *     H <- H * N/(N+batch) + (2/(N+batch)) * X^T X          H [n,n] fp32, updated in place
 * workspace: ganq_hessian_workspace_bytes() of caller-owned device scratch (partial tiles of the token-split launches
 * of staged groups; borrowed for the call, nothing is retained).  NULL / too small is allowed: the call then uses the
 * whole-tile kernel only (same H up to fp32 summation order).                                                   */
size_t ganq_hessian_workspace_bytes(int64_t rows, int64_t n);
int ganq_hessian_accum(float* H, const void* X, int dtype, int64_t rows, int64_t n, int64_t nsamples_before,
                       int64_t batch, void* workspace, size_t workspace_bytes, void* stream);

/* ---- a1, staged groups from TRANSPOSED activations (round 4).  The reference multiplies every batch as it arrives
 * (gptq.py:122-131: inp.t() in [in_features, tokens] layout, `self.H += inp.matmul(inp.t())`).  H telescopes over a
 * group of batches, so the host stages them and hands the group over once -- and because the matrix cores want the
 * TOKENS of a feature contiguous (the layout of the reference's `inp` after its `.t()`), the staging copy transposes:
 *     ganq_hessian_stage_t:  Xt[:, tok0 : tok0 + rows] = X^T          X [rows, n] one batch, Xt [n, ldt] fp16 / bf16
 *     ganq_hessian_accum_t:  H <- H * N/(N+batch) + (2/(N+batch)) * Xt[:, :rows] Xt[:, :rows]^T
 * rows of accum_t: a multiple of 32 (the host zero-fills the columns of a ragged last slice); tok0, rows of stage_t and
 * n: multiples of 8.  ganq_hessian_t_supported(n, ldt) says whether a layer is served (in_features >= 3072, a multiple
 * of 8, n * ldt * 2 B < 4 GiB); otherwise the host stages row-major and calls ganq_hessian_accum.  workspace:
 * ganq_hessian_t_workspace_bytes(n) of caller-owned scratch (ticket counters + two partial tiles per compute unit).
 * Same products as ganq_hessian_accum, another grouping of the fp32 sums; exactly symmetric; deterministic.       */
int ganq_hessian_t_supported(int64_t n, int64_t ldt);
size_t ganq_hessian_t_workspace_bytes(int64_t n);
int ganq_hessian_stage_t(void* Xt, int64_t ldt, const void* X, int64_t rows, int64_t n, int64_t tok0, void* stream);
int ganq_hessian_accum_t(float* H, const void* Xt, int64_t ldt, int dtype, int64_t rows, int64_t n, int64_t nsamples_before,
                         int64_t batch, void* workspace, size_t workspace_bytes, void* stream);
/* tests (host only, no GPU): how ganq_hessian_accum_t cuts its (tile, token slice) pairs over `ncu` workgroups, computed by
 * the code the kernels run.  hdr[7] = {Ks, G, W, R, nprim, P, Sh}; segs [ncu][W + 3][5] = {tile, s0, s1, parts, slot}
 * (tile -1: unused); parts_loc [R][max_parts][2] = {workgroup, slot} of a left-over tile's parts in summation order. */
int ganq_debug_hessian_t_cut(int64_t n, int64_t rows, int ncu, int* hdr, int* segs, int segs_cap, int* parts_loc, int max_parts);

/* ---- a2: prologue (gptq.py:280-309) -- lower Cholesky factor A = L L^T in fp32, in place (row-major, leading
 * dimension lda; the strictly upper triangle is zeroed like torch.linalg.cholesky does).  *info (device int32) is 0
 * on success or the 1-based column of the first non-positive pivot (the factor then holds NaN); nothing is
 * synchronised.  workspace: ganq_cholesky_workspace_bytes().                                                   */
size_t ganq_cholesky_workspace_bytes(int64_t n);
int ganq_cholesky(float* A, int64_t n, int64_t lda, int32_t* info, void* workspace, size_t workspace_bytes, void* stream);

/* ---- a3: codebook initialisation (ganq.py:423-438, kmeans_fit :27-30) -----------------------
 * Optimal weighted 1-D k-means per row; col_weight [n] fp64 (caller passes diag(Hinv)^-4).
 * T0 [m,V] fp32 ascending per row.  workspace: ganq_kmeans_workspace_bytes().               */
size_t ganq_kmeans_workspace_bytes(int64_t m, int64_t n, int V);
int ganq_kmeans_init(const float* W, const double* col_weight, int64_t m, int64_t n, int V, float* T0,
                     void* workspace, size_t workspace_bytes, void* stream);

/* ---- a4/a5: S-solve (ganq.py:533-565 torch branch == Metal kernel compute_s :39-270) --------
 * W [m,n], L [n,n] lower-triangular with leading dimension ldl, T [m,V].
 * Q_out [m,n] uint8.  Err_out [m,n] fp32 (W - T[Q]; the Metal kernel's `Werr`) or NULL.
 * workspace: ganq_solve_s_workspace_bytes() -- the per-tile Err scratch and the packed copy of L, plus the
 * hand-over buffers of the helper workgroups that launches with at most half as many 16-row tiles as
 * the chip has CUs use (csrc/solve_s.hip, "duo"); contents need no initialisation.              */
size_t ganq_solve_s_workspace_bytes(int64_t m, int64_t n, int V);
int ganq_solve_s(const float* W, const float* L, int64_t ldl, const float* T, int64_t m, int64_t n, int V,
                 uint8_t* Q_out, float* Err_out, void* workspace, size_t workspace_bytes, void* stream);

/* ---- dense fp32 product C[m,n] = A[m,k] @ B[k,n] on the fp32 matrix cores (W@H, ganq.py:590) */
int ganq_matmul_f32(const float* A, const float* B, int64_t m, int64_t k, int64_t n, float* C, void* stream);

/* ---- a6: T-update (ganq.py:570-591, CPU/gelsd branch) ---------------------------------------
 * WH = W @ H [m,n], H = Xxt_damped [n,n], Q [m,n].  T_out [m,V] = min-norm lstsq(S H S^T, S (WH)^T)
 * with cut-off rcond (<0: eps_fp32 * V, the torch default).  A_out [m,V,V] / b_out [m,V]: the
 * normal-equation matrices (optional, may be NULL).                                            */
size_t ganq_update_t_workspace_bytes(int64_t m, int64_t n, int V);
int ganq_update_t(const float* WH, const float* H, const uint8_t* Q, int64_t m, int64_t n, int V, double rcond,
                  float* T_out, float* A_out, float* b_out, void* workspace, size_t workspace_bytes, void* stream);

/* ---- a7: loss (ganq.py:392-395 quad_loss_2 on Wq = T.gather(1,Q), :621-622) -----------------
 * loss_out: one fp64 on the device.  workspace: ganq_quad_loss_workspace_bytes().             */
size_t ganq_quad_loss_workspace_bytes(int64_t m, int64_t n, int V);
int ganq_quad_loss(const float* W, const float* H, const float* T, const uint8_t* Q, int64_t m, int64_t n, int V,
                   double* loss_out, void* workspace, size_t workspace_bytes, void* stream);

/* ---- a7: outputs (ganq.py:633-638): Wq = T.gather(1,Q); Losses = (W-Wq)^2 / diag(Hinv)^2 / 2 -
 * Wq_out / Losses_out [m,n] fp32, either may be NULL.                                          */
int ganq_dequant_losses(const float* W, const float* T, const uint8_t* Q, const float* hinv_diag, int64_t m,
                        int64_t n, int V, float* Wq_out, float* Losses_out, void* stream);

/* ---- the whole loop (ganq.py:516-634): K x (S-solve, T-update, loss) + best-of-K ------------
 * Inputs as above plus T0 [m,V].  Outputs: T_best [m,V], Q_out [m,n], dists [K] fp64 (device),
 * best_k (device int32).  No host synchronisation inside.                                      */
size_t ganq_run_layer_workspace_bytes(int64_t m, int64_t n, int V);
int ganq_run_layer(const float* W, const float* H, const float* L, int64_t ldl, const float* T0, int64_t m,
                   int64_t n, int V, int K, uint32_t flags, double rcond, float* T_best, uint8_t* Q_out,
                   double* dists, int32_t* best_k, void* workspace, size_t workspace_bytes, void* stream);

/* ---- e: the same fused loop on a SLICE of a layer's rows, for a layer whose rows are sharded over several GPUs
 * (rows are independent in every stage of ganq.py:525-634; only the best-of-K decision :621-626 sums over all rows).
 * Same arguments as ganq_run_layer for the slice's m rows, plus optional per-iteration records (device, may be NULL):
 *   T_all [K,m,V] fp32 codebook after each iteration, loss_rows_all [K,m] fp64 per-row loss of each iteration,
 *   Q_all [K,m,n] uint8 indices of each iteration (only needed without GANQ_FLAG_ALIAS_Q).
 * T_best / Q_out / dists / best_k describe the slice alone.  The owner of the layer gathers loss_rows_all of every
 * slice in row order and calls ganq_select_best, which forms the distances in the single-call loop's own summation
 * order: a sharded layer takes bit-for-bit the decision of the unsharded one.  No collective is needed inside the loop. */
int ganq_run_layer_rows(const float* W, const float* H, const float* L, int64_t ldl, const float* T0, int64_t m,
                        int64_t n, int V, int K, uint32_t flags, double rcond, float* T_best, uint8_t* Q_out,
                        double* dists, int32_t* best_k, float* T_all, double* loss_rows_all, uint8_t* Q_all,
                        void* workspace, size_t workspace_bytes, void* stream);
/* loss_rows_all [K,m] (all rows of the layer, row order) -> dists [K] fp64, best_k int32 (device): first k with the
 * smallest distance, strict <, NaN never wins (ganq.py:625); -1 if no iteration wins. */
int ganq_select_best(const double* loss_rows_all, int64_t m, int K, double* dists, int32_t* best_k, void* stream);

/* ---- a2: prologue of GPTQ.quantize() (gptq.py:259-319) in three passes -- dead columns, act_sort permutation, ganq-style
 * offset, damping and the inputs of the two factorisations; the n-vector steps between them (dead = diag == 0, argsort, offset,
 * damp) stay with the host.
 *   ganq_prologue_rowstats: diag[i] = H[i][i], rowabs[i] = sum_j |H[i][j]|                         (gptq.py:267-269, :289-291)
 *   ganq_prologue_gather:   out_k[i][j] = H'[perm i][perm j] + (i == j ? add_k[i] : 0), index-reversed when flip_k; H' = H with
 *                           diag_fixed on its diagonal (dead columns read 1, gptq.py:268); perm NULL = identity; any out_k / add_k
 *                           may be NULL; outputs must not alias H                                   (gptq.py:281-308)
 *   ganq_prologue_weights:  W_out[r][c] = dead[perm c] ? fill_r : W[r][perm c]; fill_r = 0 (mean_fill 0, dead="zero") or the
 *                           row's mean over the live columns (dead="mean")                         (gptq.py:270-276, :283)      */
int ganq_prologue_rowstats(const float* H, int64_t n, float* diag, float* rowabs, void* stream);
int ganq_prologue_gather(const float* H, const int64_t* perm, const float* diag_fixed, int64_t n, float* out0, const float* add0,
                         int flip0, float* out1, const float* add1, int flip1, float* out2, const float* add2, int flip2,
                         void* stream);
int ganq_prologue_weights(const float* W, const int64_t* perm, const uint8_t* dead, int64_t m, int64_t n, int mean_fill,
                          float* W_out, void* stream);

/* ---- a9: LUT-dequant linear forward (replaces FakeQuantLinear.forward, fake.py:88-89) -------
 * y[M,m] = x[M,n] @ dequant(qweight, lut)^T + bias.   dtype: 0 = fp16, 1 = bf16 (x, lut, bias, y).
 * qweight: indices packed `bits` per index along the in_features dimension in the GPTQ int32
 * layout qweight[n*bits/32, m] (qlinear/__init__.py:508-517); lut [m,V]; bias [m] or NULL.     */
/* Any M, x 16-byte aligned.  M <= 64 (decode / small batches): one kernel, split-K partial tiles plus one ticket
 * counter per block of 128 features live in the workspace; the last workgroup of a block reduces and writes y.
 * M > 64 (prefill, perplexity evaluation): the fused LUT-dequant GEMM (csrc/lut_gemm.hip) -- weights decoded into the
 * matrix-core operand tile, never materialised; for few row blocks in_features is split over workgroups, the fp32
 * partial tiles live in the workspace and a second launch sums them in split order (deterministic).
 * The workspace must be zero-filled ONCE after allocation (ganq_lut_linear_workspace_init); every call leaves the
 * counters zero again, so it can be reused by later calls of any shape that fits.  One workspace serves one stream
 * at a time.                                                                                                    */
size_t ganq_lut_linear_workspace_bytes(int64_t M, int64_t m, int64_t n, int bits);
int ganq_lut_linear_workspace_init(void* workspace, size_t workspace_bytes, void* stream);
int ganq_lut_linear_fwd(const void* x, const int32_t* qweight, const void* lut, const void* bias, int dtype,
                        int64_t M, int64_t m, int64_t n, int bits, void* y, void* workspace, size_t workspace_bytes,
                        void* stream);
/* same with an fp32 addend [M,m] (or NULL) added before the single rounding to the activation dtype: the sparse-outlier
 * product of ganq_outlier_matmul */
int ganq_lut_linear_fwd_add(const void* x, const int32_t* qweight, const void* lut, const void* bias, const float* addend,
                            int dtype, int64_t M, int64_t m, int64_t n, int bits, void* y, void* workspace,
                            size_t workspace_bytes, void* stream);
/* both in one call for a layer with outliers (ganq_outlier_matmul into the tail of the workspace, then the LUT kernel);
 * the workspace is initialised like the plain one (ganq_lut_linear_workspace_init over all of it) */
size_t ganq_lut_linear_outliers_workspace_bytes(int64_t M, int64_t m, int64_t n, int bits);
int ganq_lut_linear_fwd_outliers(const void* x, const int32_t* qweight, const void* lut, const void* bias,
                                 const int32_t* rowptr, const int32_t* cols, const void* vals, int dtype, int64_t M, int64_t m,
                                 int64_t n, int bits, void* y, void* workspace, size_t workspace_bytes, void* stream);
/* prefill path: materialise Wq [m,n] = lut[o][index] in the activation dtype for a library GEMM */
int ganq_lut_dequant(const int32_t* qweight, const void* lut, int dtype, int64_t m, int64_t n, int bits, void* Wq_out,
                     void* stream);

/* pack Q [m,n] uint8 (original column order) into qweight [n*bits/32, m] int32, and back */
int ganq_pack_indices(const uint8_t* Q, int64_t m, int64_t n, int bits, int32_t* qweight, void* stream);
int ganq_unpack_indices(const int32_t* qweight, int64_t m, int64_t n, int bits, uint8_t* Q, void* stream);

/* ---- outlier split in front of GANQ (paper section 3.3 + Appendix A Algorithm 2, paper.md:195-197,882-899; SURVEY
 * 8(f) row 4 -- the reference repository does not implement it).  Row by row, with p = 1 - ratio/2, the entries
 * w >= sorted[floor(n p)] or w <= sorted[ceil(n (1-p))] are outliers; they are kept exactly in a CSR matrix and zeroed
 * in W, which GANQ then quantizes; the layer computes LUT(x) + x @ W_sparse^T.
 *   ganq_outlier_cutoffs: cut [m,2] = (c_lower, c_upper), counts [m], rowptr [m+1] (exclusive scan; rowptr[m] = nnz).
 *                         n <= 16384.  The caller reads rowptr[m] to size cols / vals.
 *   ganq_outlier_extract: cols / vals [nnz] in ascending column order per row; W is overwritten by W_dense.
 *   ganq_outlier_matmul:  out [M,m] fp32 = x [M,n] @ W_sparse^T, x and vals in `dtype` (0 = fp16, 1 = bf16).       */
int ganq_outlier_cutoffs(const float* W, int64_t m, int64_t n, double ratio, float* cut, int32_t* counts, int32_t* rowptr,
                         void* stream);
int ganq_outlier_extract(float* W, int64_t m, int64_t n, const float* cut, const int32_t* rowptr, int32_t* cols, float* vals,
                         void* stream);
int ganq_outlier_matmul(const void* x, int dtype, int64_t M, int64_t m, int64_t n, const int32_t* rowptr, const int32_t* cols,
                        const void* vals, float* out, void* stream);

/* ---- developer / test switches.  They are read from the environment (variable == option name) once, when the library
 * is loaded; afterwards only these calls change them -- the compute entry points never call getenv().  Options:
 * GANQ_T_FULL, GANQ_T_INCR_THR, GANQ_T_JACOBI, GANQ_SOLVE_ALL_ROWS, GANQ_MUPDATE_LDS, GANQ_WH_F64, GANQ_KMEANS_WCAP,
 * GANQ_CHOL_LOOKAHEAD, GANQ_ACCUM_DEBUG, GANQ_LUT_INWG, GANQ_LUT_WGS, GANQ_LUT_KS, GANQ_LUT_NT, GANQ_H_EXT, GANQ_SOLVE_VARIANT
 * (meanings: ganq_amd/csrc/runtime.hip).  None of them changes a result except where a test says so. */
int ganq_debug_set_option(const char* name, long long value);
int ganq_debug_get_option(const char* name, long long* value);
int ganq_debug_reset_option(const char* name);

/* ---- per-kernel device timing (HIP events recorded on the caller's stream around every kernel launch) ----
 * ganq_profile_enable(1) starts collecting, ganq_profile_get() sums what has completed: the caller must have
 * synchronised the stream.  kernel ids are 0 .. ganq_profile_num_kernels()-1.  Every instrumented launch puts two
 * event packets between dependent kernels (about 1 ms of idle time per 4096x4096 layer with all kernels instrumented):
 * ganq_profile_select(id) restricts the instrumentation to one kernel (-1: all, the default).                    */
int ganq_profile_enable(int on);
int ganq_profile_select(int kernel_id);
int ganq_profile_reset(void);
int ganq_profile_num_kernels(void);
const char* ganq_profile_kernel_name(int kernel_id);
int ganq_profile_get(int kernel_id, double* total_ms, int64_t* launches);

#ifdef __cplusplus
}
#endif
#endif /* GANQ_HIP_H */

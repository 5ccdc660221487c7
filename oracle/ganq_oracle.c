/*
 * ganq_oracle.c -- CPU restatement of the reference's GANQ hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the parity checker for the HIP kernels in ganq_amd/csrc.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, load or call it.  The
 * product path (ganq_amd/) never imports it and has no CPU fallback.
 *
 * Every function cites the reference lines it restates (paths relative to the upstream
 * smpanaro/ganq tree).  Pinning: tests/test_oracle_golden.py checks each function against
 * the .npz files under tests/golden, which were produced by running the reference's own ganq.py / gptq.py
 * (tests/golden/make_golden.py).  The one stage that is NOT pinned is the initial codebook:
 * the reference takes it from the third-party `kmeans1d` package
 * (git+https://github.com/smpanaro/kmeans1d@831c169c3729aba18ca9ff4e57c4a7d26bcc8271,
 * requirements.txt:16), which is not vendored in the reference tree and not installed here.
 * ganq_oracle_kmeans_init restates the published algorithm (optimal weighted 1-D k-means by
 * dynamic programming over the sorted values) -- "parity unpinned" for T0; it is verified
 * against brute-force enumeration instead (tests/test_oracle_kmeans.py).
 *
 * Arithmetic contract ("canonical order") for the S-solve, shared bit-for-bit with the HIP
 * kernel: see ganq_oracle_solve_s.  Build with -ffp-contract=off (oracle/Makefile).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define GANQ_MAX_V 256

int ganq_oracle_version(void) { return 1; }

int ganq_oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void ganq_oracle_set_num_threads(int t) {
#ifdef _OPENMP
    if (t > 0) omp_set_num_threads(t);
#else
    (void)t;
#endif
}

/* ------------------------------------------------------------------------------------------
 * S-solve  (ganq.py:533-565 torch branch; ganq.py:94-247 Metal kernel `compute_s`)
 *
 * For every row i independently, for j = n-1 .. 0:
 *     eff   = W[i,j] + r_j / L[j,j]                         ganq.py:540-542
 *     idx   = first argmin_s |eff - T[i,s]|  (strict <; a NaN distance -- NaN codebook entry or residual -- counts as
 *             smaller than everything, the first one wins: torch.argmin)          ganq.py:546-547, :107-123
 *     Q[i,j]= idx                                           ganq.py:550
 *     err_j = W[i,j] - T[i,idx]                             ganq.py:125-126 / :564-565
 *     r_c   = sum_{u>c} err_u * L[u,c]                      ganq.py:565 (column j-1 of L)
 *
 * The reference evaluates r_{j-1} as a fresh gemv over u = j..n-1 whose fp32 summation order
 * is whatever the BLAS picks.  The canonical order fixed here (and implemented identically on
 * the GPU) is one fused-multiply-add chain per column c, running over u in DESCENDING order:
 *     r_c = fmaf(err_{c+1}, L[c+1,c], ... fmaf(err_{n-2}, L[n-2,c], fmaf(err_{n-1}, L[n-1,c], 0)))
 * The division is IEEE round-to-nearest fp32, the add W + q is a separate rounding.
 *
 * W [m,n], L [n,n] row-major lower-triangular (L[u*ldl + c]), T [m,V].
 * Q out [m,n] uint8; Err out [m,n] fp32 or NULL.
 * ------------------------------------------------------------------------------------------ */
/* Speed (round 4; the bits are those of the plain loop it replaces): SS_R rows share every load of a row of L, and the
 * update of the residual accumulators -- independent fmaf chains, one per (row, column) -- is a SIMD loop.  fmaf() is
 * exactly rounded whether libm computes it or a vfmadd instruction does, so the two builds of the body (AVX2 + FMA when
 * the CPU has them, baseline otherwise; chosen at run time) give the same indices and errors. */
#define SS_R 8

static inline __attribute__((always_inline)) void solve_rows_body(const float* W, const float* L, int64_t ldl, const float* T,
                                                                  int64_t n, int V, int64_t i0, int R, uint8_t* Q, float* Err,
                                                                  float* racc /* [SS_R][n] */) {
    float err[SS_R];
    for (int r = 0; r < R; ++r)
        for (int64_t c = 0; c < n; ++c) racc[(size_t)r * n + c] = 0.0f;
    for (int64_t j = n - 1; j >= 0; --j) {
        const float* lrow = L + j * ldl;
        for (int r = 0; r < R; ++r) {
            const int64_t i = i0 + r;
            const float* w = W + i * n;
            const float* t = T + i * V;
            float q = racc[(size_t)r * n + j] / lrow[j];
            float eff = w[j] + q;
            float best = INFINITY;
            int idx = 0, have_nan = 0;
            for (int s = 0; s < V; ++s) {  /* torch.argmin (ganq.py:547): the first NaN wins, else the first minimum */
                float d = fabsf(eff - t[s]);
                if (d != d) {
                    if (!have_nan) {
                        have_nan = 1;
                        idx = s;
                    }
                } else if (!have_nan && d < best) {
                    best = d;
                    idx = s;
                }
            }
            err[r] = w[j] - t[idx];
            Q[i * n + j] = (uint8_t)idx;
            if (Err) Err[i * n + j] = err[r];
        }
        for (int r = 0; r < R; ++r) {
            float* ra = racc + (size_t)r * n;
            const float e = err[r];
#pragma omp simd
            for (int64_t c = 0; c < j; ++c) ra[c] = fmaf(e, lrow[c], ra[c]);
        }
    }
}

__attribute__((target("avx2,fma"))) static void solve_rows_fma(const float* W, const float* L, int64_t ldl, const float* T, int64_t n,
                                                               int V, int64_t i0, int R, uint8_t* Q, float* Err, float* racc) {
    solve_rows_body(W, L, ldl, T, n, V, i0, R, Q, Err, racc);
}

static void solve_rows_base(const float* W, const float* L, int64_t ldl, const float* T, int64_t n, int V, int64_t i0, int R,
                            uint8_t* Q, float* Err, float* racc) {
    solve_rows_body(W, L, ldl, T, n, V, i0, R, Q, Err, racc);
}

static int cpu_has_fma(void) {
    static int known = -1;
    if (known < 0) {
        __builtin_cpu_init();
        known = (__builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma")) ? 1 : 0;
    }
    return known;
}

/* fp64 accumulation of fp32 vectors, element by element (no reduction: the order of additions INTO an element is the
 * caller's loop order); an AVX2 build of the same statements where the CPU has it -- no FMA flag, so still one rounded
 * product and one rounded sum per element (-ffp-contract=off besides) */
static inline __attribute__((always_inline)) void axpy_body(double* acc, const float* x, double a, int64_t n) {
#pragma omp simd
    for (int64_t v = 0; v < n; ++v) acc[v] += a * (double)x[v];
}
static inline __attribute__((always_inline)) void addw_body(double* acc, const float* x, int64_t n) {
#pragma omp simd
    for (int64_t v = 0; v < n; ++v) acc[v] += (double)x[v];
}
__attribute__((target("avx2"))) static void axpy_avx2(double* acc, const float* x, double a, int64_t n) { axpy_body(acc, x, a, n); }
__attribute__((target("avx2"))) static void addw_avx2(double* acc, const float* x, int64_t n) { addw_body(acc, x, n); }
static void axpy_base(double* acc, const float* x, double a, int64_t n) { axpy_body(acc, x, a, n); }
static void addw_base(double* acc, const float* x, int64_t n) { addw_body(acc, x, n); }
typedef void (*axpy_fn)(double*, const float*, double, int64_t);
typedef void (*addw_fn)(double*, const float*, int64_t);
static axpy_fn pick_axpy(void) { return cpu_has_fma() ? axpy_avx2 : axpy_base; }
static addw_fn pick_addw(void) { return cpu_has_fma() ? addw_avx2 : addw_base; }

int ganq_oracle_solve_s(const float* W, const float* L, int64_t ldl, const float* T, int64_t m, int64_t n,
                        int V, uint8_t* Q, float* Err) {
    if (V < 1 || V > GANQ_MAX_V || m < 0 || n < 0) return -1;
    int fail = 0;
    const int use_fma = cpu_has_fma();
    const int64_t blocks = (m + SS_R - 1) / SS_R;
#pragma omp parallel
    {
        float* racc = (float*)malloc(sizeof(float) * SS_R * (size_t)(n > 0 ? n : 1));
        if (!racc) {
#pragma omp atomic write
            fail = 1;
        } else {
#pragma omp for schedule(dynamic, 1)
            for (int64_t blk = 0; blk < blocks; ++blk) {
                const int64_t i0 = blk * SS_R;
                const int R = (int)((m - i0) < SS_R ? (m - i0) : SS_R);
                if (use_fma) solve_rows_fma(W, L, ldl, T, n, V, i0, R, Q, Err, racc);
                else solve_rows_base(W, L, ldl, T, n, V, i0, R, Q, Err, racc);
            }
            free(racc);
        }
    }
    return fail ? -2 : 0;
}

/* ------------------------------------------------------------------------------------------
 * Symmetric eigen-decomposition by cyclic Jacobi, fp64, in place.
 * A [V,V] symmetric (destroyed; diagonal holds eigenvalues on return), E [V,V] eigenvectors
 * in columns.  Used for the minimum-norm solve below.
 * ------------------------------------------------------------------------------------------ */
static void jacobi_eig(double* A, double* E, int V) {
    for (int a = 0; a < V; ++a)
        for (int b = 0; b < V; ++b) E[a * V + b] = (a == b) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 64; ++sweep) {
        double off = 0.0, diag = 0.0;
        for (int a = 0; a < V; ++a) {
            diag += A[a * V + a] * A[a * V + a];
            for (int b = a + 1; b < V; ++b) off += A[a * V + b] * A[a * V + b];
        }
        if (off <= 1e-30 * diag || off == 0.0) break;
        for (int p = 0; p < V - 1; ++p) {
            for (int q = p + 1; q < V; ++q) {
                double apq = A[p * V + q];
                if (apq == 0.0) continue;
                double app = A[p * V + p], aqq = A[q * V + q];
                double theta = (aqq - app) / (2.0 * apq);
                double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < V; ++k) { /* columns p,q */
                    double akp = A[k * V + p], akq = A[k * V + q];
                    A[k * V + p] = c * akp - s * akq;
                    A[k * V + q] = s * akp + c * akq;
                }
                for (int k = 0; k < V; ++k) { /* rows p,q */
                    double apk = A[p * V + k], aqk = A[q * V + k];
                    A[p * V + k] = c * apk - s * aqk;
                    A[q * V + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < V; ++k) {
                    double ekp = E[k * V + p], ekq = E[k * V + q];
                    E[k * V + p] = c * ekp - s * ekq;
                    E[k * V + q] = s * ekp + c * ekq;
                }
            }
        }
    }
}

/* Minimum-norm least-squares solution of the symmetric system A x = b with the gelsd cut-off:
 * eigenvalues with |lambda| <= rcond * max|lambda| are dropped (torch.linalg.lstsq(...,
 * driver="gelsd"), default rcond = eps_fp32 * max(rows, cols); ganq.py:589-591). */
static void minnorm_solve(double* A, const double* b, int V, double rcond, double* x, double* E) {
    jacobi_eig(A, E, V);
    double lmax = 0.0;
    for (int k = 0; k < V; ++k) lmax = fmax(lmax, fabs(A[k * V + k]));
    for (int a = 0; a < V; ++a) x[a] = 0.0;
    for (int k = 0; k < V; ++k) {
        double lam = A[k * V + k];
        if (!(fabs(lam) > rcond * lmax)) continue;
        double proj = 0.0;
        for (int a = 0; a < V; ++a) proj += E[a * V + k] * b[a];
        proj /= lam;
        for (int a = 0; a < V; ++a) x[a] += proj * E[a * V + k];
    }
}

/* ------------------------------------------------------------------------------------------
 * T-update  (ganq.py:570-591, CPU "least_squares" branch)
 *     A_i = S_i H S_i^T   [V,V],   b_i = S_i (W H)_i^T   [V],   T_i = lstsq(A_i, b_i) (min-norm)
 * S_i is the one-hot expansion of Q[i,:]; it is never materialised:
 *     A_i[a][b] = sum_{u,v} [Q_iu == a][Q_iv == b] H[u,v],   b_i[a] = sum_u [Q_iu == a] WH[i,u]
 * Sums are taken in fp64 and rounded to fp32 (the reference holds A and b in fp32), the
 * solve runs in fp64 on those fp32 values, T is rounded to fp32.
 * rcond < 0 selects the reference default eps_fp32 * V.
 * Optional outputs A_out [m,V,V], b_out [m,V] (fp32) for stage-wise checks.
 * ------------------------------------------------------------------------------------------ */
/* Speed (round 4): UT_R rows share every load of H, column strip by column strip of UT_W columns, so that the rows' partial
 * sums G stay in cache.  Every element keeps its own order of additions -- G[a][v] over ascending u, then A[a][b] over
 * ascending v -- so the sums are the bits of the plain row-at-a-time loops. */
#define UT_R 8
#define UT_W 512
int ganq_oracle_update_t(const float* WH, const float* H, const uint8_t* Q, int64_t m, int64_t n, int V,
                         double rcond, float* T_out, float* A_out, float* b_out) {
    if (V < 1 || V > GANQ_MAX_V) return -1;
    if (rcond < 0) rcond = 1.1920928955078125e-07 * (double)V;
    int fail = 0;
    const addw_fn addw = pick_addw();
    const int64_t blocks = (m + UT_R - 1) / UT_R;
#pragma omp parallel
    {
        double* G = (double*)malloc(sizeof(double) * UT_R * (size_t)V * UT_W); /* strips of S_i H: [UT_R][V][UT_W] */
        double* Aall = (double*)malloc(sizeof(double) * UT_R * V * V);
        double* E = (double*)malloc(sizeof(double) * V * V);
        double* b = (double*)malloc(sizeof(double) * V);
        double* x = (double*)malloc(sizeof(double) * V);
        if (!G || !Aall || !E || !b || !x) {
#pragma omp atomic write
            fail = 1;
        } else {
#pragma omp for schedule(dynamic, 1)
            for (int64_t blk = 0; blk < blocks; ++blk) {
                const int64_t i0 = blk * UT_R;
                const int R = (int)((m - i0) < UT_R ? (m - i0) : UT_R);
                for (int a = 0; a < R * V * V; ++a) Aall[a] = 0.0;
                for (int64_t v0 = 0; v0 < n; v0 += UT_W) {
                    const int64_t w = (n - v0) < UT_W ? (n - v0) : UT_W;
                    memset(G, 0, sizeof(double) * (size_t)R * (size_t)V * UT_W);
                    for (int64_t u = 0; u < n; ++u) {
                        const float* h = H + u * n + v0;
                        for (int r = 0; r < R; ++r) {
                            addw(G + ((size_t)r * V + Q[(i0 + r) * n + u]) * UT_W, h, w);
                        }
                    }
                    for (int r = 0; r < R; ++r) {
                        const uint8_t* q = Q + (i0 + r) * n + v0;
                        for (int a = 0; a < V; ++a) {
                            const double* g = G + ((size_t)r * V + a) * UT_W;
                            double* arow = Aall + ((size_t)r * V + a) * V;
                            for (int64_t v = 0; v < w; ++v) arow[q[v]] += g[v];
                        }
                    }
                }
                for (int r = 0; r < R; ++r) {
                    const int64_t i = i0 + r;
                    const uint8_t* q = Q + i * n;
                    double* A = Aall + (size_t)r * V * V;
                    for (int a = 0; a < V; ++a) b[a] = 0.0;
                    for (int64_t u = 0; u < n; ++u) b[q[u]] += (double)WH[i * n + u];
                    for (int a = 0; a < V * V; ++a) A[a] = (double)(float)A[a];
                    for (int a = 0; a < V; ++a) b[a] = (double)(float)b[a];
                    /* symmetrise (H is symmetric up to fp32 noise; the eigen-solve wants exact symmetry) */
                    for (int a = 0; a < V; ++a)
                        for (int c = a + 1; c < V; ++c) {
                            double sy = 0.5 * (A[a * V + c] + A[c * V + a]);
                            A[a * V + c] = sy;
                            A[c * V + a] = sy;
                        }
                    if (A_out)
                        for (int a = 0; a < V * V; ++a) A_out[i * V * V + a] = (float)A[a];
                    if (b_out)
                        for (int a = 0; a < V; ++a) b_out[i * V + a] = (float)b[a];
                    minnorm_solve(A, b, V, rcond, x, E);
                    for (int a = 0; a < V; ++a) T_out[i * V + a] = (float)x[a];
                }
            }
        }
        free(G);
        free(Aall);
        free(E);
        free(b);
        free(x);
    }
    return fail ? -2 : 0;
}

/* Solve only: A [m,V,V] fp32, b [m,V] fp32 -> T [m,V]  (same solver as above). */
int ganq_oracle_minnorm_solve(const float* A_in, const float* b_in, int64_t m, int V, double rcond, float* T_out) {
    if (V < 1 || V > GANQ_MAX_V) return -1;
    if (rcond < 0) rcond = 1.1920928955078125e-07 * (double)V;
#pragma omp parallel
    {
        double* A = (double*)malloc(sizeof(double) * V * V);
        double* E = (double*)malloc(sizeof(double) * V * V);
        double* b = (double*)malloc(sizeof(double) * V);
        double* x = (double*)malloc(sizeof(double) * V);
#pragma omp for schedule(dynamic, 8)
        for (int64_t i = 0; i < m; ++i) {
            for (int a = 0; a < V; ++a)
                for (int c = 0; c < V; ++c)
                    A[a * V + c] = 0.5 * ((double)A_in[i * V * V + a * V + c] + (double)A_in[i * V * V + c * V + a]);
            for (int a = 0; a < V; ++a) b[a] = (double)b_in[i * V + a];
            minnorm_solve(A, b, V, rcond, x, E);
            for (int a = 0; a < V; ++a) T_out[i * V + a] = (float)x[a];
        }
        free(A);
        free(E);
        free(b);
        free(x);
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Dense fp32 product C[m,n] = A[m,k] @ B[k,n] with fp64 accumulation (W@H, ganq.py:590).
 * ------------------------------------------------------------------------------------------ */
#define MM_R 8
#define MM_W 448
int ganq_oracle_matmul(const float* A, const float* B, int64_t m, int64_t k, int64_t n, float* C) {
    const axpy_fn axpy = pick_axpy();
    const int64_t blocks = (m + MM_R - 1) / MM_R;
#pragma omp parallel
    {
        double* acc = (double*)malloc(sizeof(double) * MM_R * (size_t)(n > 0 ? n : 1));
#pragma omp for schedule(dynamic, 1)
        for (int64_t blk = 0; blk < blocks; ++blk) {  /* MM_R rows share every row of B; each element: one sum over ascending u */
            const int64_t i0 = blk * MM_R;
            const int R = (int)((m - i0) < MM_R ? (m - i0) : MM_R);
            for (int64_t v = 0; v < (int64_t)R * n; ++v) acc[v] = 0.0;
            for (int64_t v0 = 0; v0 < n; v0 += MM_W) {  /* column strips: the R partial rows of a strip stay in L1 */
                const int64_t w = (n - v0) < MM_W ? (n - v0) : MM_W;
                for (int64_t u = 0; u < k; ++u) {
                    const float* brow = B + u * n + v0;
                    for (int r = 0; r < R; ++r) axpy(acc + (size_t)r * n + v0, brow, (double)A[(i0 + r) * k + u], w);
                }
            }
            for (int r = 0; r < R; ++r)
                for (int64_t v = 0; v < n; ++v) C[(i0 + r) * n + v] = (float)acc[(size_t)r * n + v];
        }
        free(acc);
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * quad_loss_2  (ganq.py:392-395, called at :621-622):  Wq = T.gather(1,Q);
 *     dist = sum( ((W - Wq) @ H) * (W - Wq) )
 * fp64 accumulation of fp32 inputs; per-row terms optionally returned.
 * ------------------------------------------------------------------------------------------ */
int ganq_oracle_quad_loss(const float* W, const float* H, const float* T, const uint8_t* Q, int64_t m, int64_t n,
                          int V, double* loss_out, double* row_loss_out) {
    /* per-row terms first, then one sum in row order: the total does not depend on the OpenMP schedule (two
     * iterations with the same indices must give the same distance bit for bit, best-of-K compares them) */
    double* rows = (double*)malloc(sizeof(double) * (size_t)(m > 0 ? m : 1));
    const axpy_fn axpy = pick_axpy();
    const int64_t blocks = (m + MM_R - 1) / MM_R;
#pragma omp parallel
    {
        float* e = (float*)malloc(sizeof(float) * MM_R * (size_t)(n > 0 ? n : 1));
        double* acc = (double*)malloc(sizeof(double) * MM_R * (size_t)(n > 0 ? n : 1));
#pragma omp for schedule(dynamic, 1)
        for (int64_t blk = 0; blk < blocks; ++blk) {  /* MM_R rows share every row of H; each element: one sum over ascending u */
            const int64_t i0 = blk * MM_R;
            const int R = (int)((m - i0) < MM_R ? (m - i0) : MM_R);
            for (int r = 0; r < R; ++r) {
                const int64_t i = i0 + r;
                for (int64_t u = 0; u < n; ++u) e[(size_t)r * n + u] = W[i * n + u] - T[i * V + Q[i * n + u]];
            }
            for (int64_t v = 0; v < (int64_t)R * n; ++v) acc[v] = 0.0;
            for (int64_t v0 = 0; v0 < n; v0 += MM_W) {
                const int64_t w = (n - v0) < MM_W ? (n - v0) : MM_W;
                for (int64_t u = 0; u < n; ++u) {
                    const float* h = H + u * n + v0;
                    for (int r = 0; r < R; ++r) axpy(acc + (size_t)r * n + v0, h, (double)e[(size_t)r * n + u], w);
                }
            }
            for (int r = 0; r < R; ++r) {
                double rs = 0.0;
                for (int64_t v = 0; v < n; ++v) rs += acc[(size_t)r * n + v] * (double)e[(size_t)r * n + v];
                rows[i0 + r] = rs;
            }
        }
        free(e);
        free(acc);
    }
    double total = 0.0;
    for (int64_t i = 0; i < m; ++i) {
        if (row_loss_out) row_loss_out[i] = rows[i];
        total += rows[i];
    }
    free(rows);
    *loss_out = total;
    return 0;
}

/* Wq = T.gather(1,Q) (ganq.py:633-634) and Losses = (W-Wq)^2 / diag(Hinv)^2 / 2 (ganq.py:637-638). */
int ganq_oracle_dequant_losses(const float* W, const float* T, const uint8_t* Q, const float* hinv_diag, int64_t m,
                               int64_t n, int V, float* Wq, float* Losses) {
    for (int64_t i = 0; i < m; ++i)
        for (int64_t u = 0; u < n; ++u) {
            float wq = T[i * V + Q[i * n + u]];
            if (Wq) Wq[i * n + u] = wq;
            if (Losses) {
                float d = hinv_diag[u];
                float e = W[i * n + u] - wq;
                Losses[i * n + u] = ((e * e) / (d * d)) / 2.0f;
            }
        }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Codebook initialisation  (ganq.py:423-438 + kmeans_fit :27-30)
 * Per row: optimal weighted 1-D k-means of the n weights into V clusters, point weight for
 * column u = weights[u] (the caller passes diag(Hinv)^-4, ganq.py:427-429), centroids =
 * weighted means, ascending.  Exact dynamic programme over the sorted values
 *     D[k][i] = min_j D[k-1][j-1] + cost(j..i),   cost = sum w x^2 - (sum w x)^2 / sum w
 * solved with divide-and-conquer over the monotone argmin (the cost is concave-Monge on
 * sorted points), fp64 throughout, leftmost argmin on ties.  Restates the algorithm of the
 * un-vendored `kmeans1d` dependency; parity with it is unpinned (see file header).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    const double *cw, *cwx, *cwxx; /* prefix sums, length n+1 */
} km_prefix;

static inline double km_cost(const km_prefix* p, int64_t j, int64_t i) { /* points j..i inclusive */
    if (i < j) return 0.0;
    double w = p->cw[i + 1] - p->cw[j];
    double wx = p->cwx[i + 1] - p->cwx[j];
    double wxx = p->cwxx[i + 1] - p->cwxx[j];
    if (!(w > 0.0)) return 0.0;
    double c = wxx - (wx * wx) / w;
    return c > 0.0 ? c : 0.0;
}

static void km_dc(const km_prefix* p, const double* prev, double* cur, int32_t* arg, int64_t lo, int64_t hi,
                  int64_t optlo, int64_t opthi) {
    /* recursion depth is log2(n) */
    if (lo > hi) return;
    int64_t mid = (lo + hi) / 2;
    int64_t jhi = opthi < mid ? opthi : mid;
    double best = INFINITY;
    int64_t bj = optlo;
    for (int64_t j = optlo; j <= jhi; ++j) {
        double c = (j == 0 ? 0.0 : prev[j - 1]) + km_cost(p, j, mid);
        if (c < best) {
            best = c;
            bj = j;
        }
    }
    cur[mid] = best;
    arg[mid] = (int32_t)bj;
    km_dc(p, prev, cur, arg, lo, mid - 1, optlo, bj);
    km_dc(p, prev, cur, arg, mid + 1, hi, bj, opthi);
}

static int cmp_pair(const void* a, const void* b) {
    const double* x = (const double*)a;
    const double* y = (const double*)b;
    if (x[0] < y[0]) return -1;
    if (x[0] > y[0]) return 1;
    if (x[2] < y[2]) return -1; /* stable on original index */
    if (x[2] > y[2]) return 1;
    return 0;
}

int ganq_oracle_kmeans_row(const float* w, const double* weights, int64_t n, int V, float* centroids) {
    if (n < 1 || V < 1) return -1;
    double* pairs = (double*)malloc(sizeof(double) * 3 * (size_t)n);
    double* pre = (double*)malloc(sizeof(double) * 3 * (size_t)(n + 1));
    double* D0 = (double*)malloc(sizeof(double) * (size_t)n);
    double* D1 = (double*)malloc(sizeof(double) * (size_t)n);
    int32_t* arg = (int32_t*)malloc(sizeof(int32_t) * (size_t)V * (size_t)n);
    if (!pairs || !pre || !D0 || !D1 || !arg) return -2;
    for (int64_t u = 0; u < n; ++u) {
        pairs[3 * u] = (double)w[u];
        pairs[3 * u + 1] = weights ? weights[u] : 1.0;
        pairs[3 * u + 2] = (double)u;
    }
    qsort(pairs, (size_t)n, sizeof(double) * 3, cmp_pair);
    double* cw = pre;
    double* cwx = pre + (n + 1);
    double* cwxx = pre + 2 * (n + 1);
    /* two-level summation (chunks of 16 summed left to right, chunk totals accumulated left to right): the
     * association order the GPU kernel uses, so both produce the same fp64 prefix sums */
    {
        double oa = 0.0, ob = 0.0, od = 0.0;
        for (int64_t c0 = 0; c0 < n; c0 += 16) {
            double a = 0.0, b = 0.0, d = 0.0;
            int64_t hi = c0 + 16 < n ? c0 + 16 : n;
            for (int64_t u = c0; u < hi; ++u) {
                cw[u] = oa + a;
                cwx[u] = ob + b;
                cwxx[u] = od + d;
                double x = pairs[3 * u], ww = pairs[3 * u + 1];
                a += ww;
                b += ww * x;
                d += ww * x * x;
            }
            oa += a;
            ob += b;
            od += d;
        }
        cw[n] = oa;
        cwx[n] = ob;
        cwxx[n] = od;
    }
    km_prefix p = {cw, cwx, cwxx};
    for (int64_t i = 0; i < n; ++i) {
        D0[i] = km_cost(&p, 0, i);
        arg[i] = 0;
    }
    double *prev = D0, *cur = D1;
    for (int k = 1; k < V; ++k) {
        /* cluster k starts at j >= 1 when possible (every earlier cluster non-empty) -- with
         * fewer than k+1 points the layer degenerates to empty clusters of cost 0 */
        int32_t* a = arg + (size_t)k * n;
        for (int64_t i = 0; i < n; ++i) {
            cur[i] = INFINITY;
            a[i] = 0;
        }
        km_dc(&p, prev, cur, a, 0, n - 1, 0, n - 1);
        double* t = prev;
        prev = cur;
        cur = t;
    }
    /* backtrack */
    int64_t end = n - 1;
    for (int k = V - 1; k >= 0; --k) {
        int64_t start = (end >= 0) ? arg[(size_t)k * n + end] : 0;
        if (k == 0) start = 0;
        if (end >= start && end >= 0) {
            double sw = cw[end + 1] - cw[start], swx = cwx[end + 1] - cwx[start];
            centroids[k] = (float)(sw > 0.0 ? swx / sw : pairs[3 * start]);
        } else {
            /* empty cluster: repeat the neighbouring value (cannot happen with >= V distinct points) */
            centroids[k] = (k + 1 < V) ? centroids[k + 1] : (float)pairs[3 * (n - 1)];
        }
        end = start - 1;
    }
    free(pairs);
    free(pre);
    free(D0);
    free(D1);
    free(arg);
    return 0;
}

int ganq_oracle_kmeans_init(const float* W, const double* weights, int64_t m, int64_t n, int V, float* T0) {
    int fail = 0;
#pragma omp parallel for schedule(dynamic, 4)
    for (int64_t i = 0; i < m; ++i) {
        int rc = ganq_oracle_kmeans_row(W + i * n, weights, n, V, T0 + i * V);
        if (rc) {
#pragma omp atomic write
            fail = rc;
        }
    }
    return fail;
}

/* Total weighted within-cluster cost of assigning sorted points to the nearest centroid set --
 * helper for the brute-force k-means test. */
double ganq_oracle_kmeans_cost(const float* w, const double* weights, int64_t n, const int32_t* labels, int V) {
    double sw[GANQ_MAX_V] = {0}, swx[GANQ_MAX_V] = {0}, swxx[GANQ_MAX_V] = {0};
    for (int64_t u = 0; u < n; ++u) {
        double ww = weights ? weights[u] : 1.0, x = (double)w[u];
        sw[labels[u]] += ww;
        swx[labels[u]] += ww * x;
        swxx[labels[u]] += ww * x * x;
    }
    double c = 0.0;
    for (int k = 0; k < V; ++k)
        if (sw[k] > 0.0) c += swxx[k] - swx[k] * swx[k] / sw[k];
    return c;
}

/* ------------------------------------------------------------------------------------------
 * Full alternating optimisation  (ganq.py:516-634): K x (S-solve, T-update, loss), best-of-K.
 *
 * alias_q != 0 reproduces the reference's torch branch exactly: there `Q` is ONE tensor that
 * every iteration overwrites in place (ganq.py:487,550) while `best` stores a reference to it
 * (ganq.py:625-626), so the returned indices are always those of the LAST iteration, paired
 * with the codebook of the BEST iteration.  alias_q == 0 pairs the best codebook with its own
 * indices (what the reference's MLX branch does, where Q is rebound each iteration, :529).
 *
 * Outputs: T_best [m,V], Q_out [m,n], dists [K] (fp64), best_k.
 * ------------------------------------------------------------------------------------------ */
/* Optional per-iteration records for sampled-row parity at full size (rows are independent in every stage, the
 * best-of-K decision is the only global step): T_all [K,m,V], Q_all [K,m,n], loss_rows_all [K,m] -- each may be NULL. */
int ganq_oracle_run_layer_trace(const float* W, const float* H, const float* L, const float* T0, int64_t m, int64_t n,
                                int V, int K, int alias_q, double rcond, float* T_best, uint8_t* Q_out, double* dists,
                                int* best_k, float* T_all, uint8_t* Q_all, double* loss_rows_all) {
    float* WH = (float*)malloc(sizeof(float) * (size_t)m * (size_t)n);
    float* T = (float*)malloc(sizeof(float) * (size_t)m * V);
    float* Tn = (float*)malloc(sizeof(float) * (size_t)m * V);
    uint8_t* Q = (uint8_t*)malloc((size_t)m * (size_t)n);
    double* rows = (double*)malloc(sizeof(double) * (size_t)(m > 0 ? m : 1));
    if (!WH || !T || !Tn || !Q || !rows) return -2;
    ganq_oracle_matmul(W, H, m, n, n, WH);
    memcpy(T, T0, sizeof(float) * (size_t)m * V);
    double best = INFINITY;
    *best_k = -1;
    for (int k = 0; k < K; ++k) {
        int rc = ganq_oracle_solve_s(W, L, n, T, m, n, V, Q, NULL);
        if (rc) return rc;
        rc = ganq_oracle_update_t(WH, H, Q, m, n, V, rcond, Tn, NULL, NULL);
        if (rc) return rc;
        memcpy(T, Tn, sizeof(float) * (size_t)m * V);
        double d;
        ganq_oracle_quad_loss(W, H, T, Q, m, n, V, &d, rows);
        dists[k] = d;
        if (T_all) memcpy(T_all + (size_t)k * m * V, T, sizeof(float) * (size_t)m * V);
        if (Q_all) memcpy(Q_all + (size_t)k * m * n, Q, (size_t)m * (size_t)n);
        if (loss_rows_all) memcpy(loss_rows_all + (size_t)k * m, rows, sizeof(double) * (size_t)m);
        if (d < best) {
            best = d;
            *best_k = k;
            memcpy(T_best, T, sizeof(float) * (size_t)m * V);
            if (!alias_q) memcpy(Q_out, Q, (size_t)m * (size_t)n);
        }
    }
    if (alias_q || *best_k < 0) memcpy(Q_out, Q, (size_t)m * (size_t)n);
    if (*best_k < 0) memcpy(T_best, T, sizeof(float) * (size_t)m * V);
    free(WH);
    free(T);
    free(Tn);
    free(Q);
    free(rows);
    return 0;
}

int ganq_oracle_run_layer(const float* W, const float* H, const float* L, const float* T0, int64_t m, int64_t n,
                          int V, int K, int alias_q, double rcond, float* T_best, uint8_t* Q_out, double* dists,
                          int* best_k) {
    return ganq_oracle_run_layer_trace(W, H, L, T0, m, n, V, K, alias_q, rcond, T_best, Q_out, dists, best_k, NULL, NULL,
                                       NULL);
}

/* ------------------------------------------------------------------------------------------
 * Hessian accumulation  (gptq.py:96-131).  X [rows = b*seq, n] as raw IEEE fp16 bits,
 * nsamples_before = sequences seen so far, b = sequences in this batch (gptq.py:104).
 *     H *= N/(N+b);  N += b;  Xs = sqrt(2/N) * float(X);  H += Xs^T Xs
 * The scaled activations are rounded to fp32 as in the reference (gptq.py:129); the product
 * sum is taken in fp64 and rounded once per batch.
 * ------------------------------------------------------------------------------------------ */
static float half_to_float(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1f;
    uint32_t man = h & 0x3ffu;
    uint32_t f;
    if (exp == 0) {
        if (man == 0) {
            f = sign;
        } else {
            int e = -1;
            do {
                man <<= 1;
                ++e;
            } while (!(man & 0x400u));
            f = sign | ((uint32_t)(127 - 15 - e) << 23) | ((man & 0x3ffu) << 13);
        }
    } else if (exp == 31) {
        f = sign | 0x7f800000u | (man << 13);
    } else {
        f = sign | ((exp + 112) << 23) | (man << 13);
    }
    float out;
    memcpy(&out, &f, 4);
    return out;
}

int ganq_oracle_hessian_accum(float* H, const uint16_t* X, int64_t rows, int64_t n, int64_t nsamples_before,
                              int64_t b) {
    double nn = (double)(nsamples_before + b);
    float decay = (float)((double)nsamples_before / nn);
    float scale = (float)sqrt(2.0 / nn);
    float* Xs = (float*)malloc(sizeof(float) * (size_t)rows * (size_t)n);
    if (!Xs) return -2;
    for (int64_t t = 0; t < rows * n; ++t) Xs[t] = scale * half_to_float(X[t]);
#pragma omp parallel for schedule(dynamic, 4)
    for (int64_t u = 0; u < n; ++u) {
        for (int64_t v = 0; v < n; ++v) {
            double acc = 0.0;
            for (int64_t t = 0; t < rows; ++t) acc += (double)Xs[t * n + u] * (double)Xs[t * n + v];
            float h = (nsamples_before > 0) ? H[u * n + v] * decay : 0.0f;
            H[u * n + v] = h + (float)acc;
        }
    }
    free(Xs);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * LUT-dequant linear forward.  Oracle for GanqHipQuantLinear == FakeQuantLinear.forward
 * (fake.py:88-89) on Wq = T.gather(1,Q).half():   y = x @ Wq^T + bias.
 * x [M,n] fp16 bits, lut [m,V] fp16 bits, Q [m,n] uint8, bias [m] fp16 bits or NULL.
 * y_out [M,m] as fp32 (exact fp64-accumulated value rounded to fp32; the test rounds to fp16).
 * ------------------------------------------------------------------------------------------ */
int ganq_oracle_lut_linear(const uint16_t* x, const uint8_t* Q, const uint16_t* lut, const uint16_t* bias,
                           int64_t M, int64_t m, int64_t n, int V, float* y_out) {
#pragma omp parallel for schedule(dynamic, 8)
    for (int64_t o = 0; o < m; ++o) {
        for (int64_t r = 0; r < M; ++r) {
            double acc = 0.0;
            for (int64_t u = 0; u < n; ++u)
                acc += (double)half_to_float(x[r * n + u]) * (double)half_to_float(lut[o * V + Q[o * n + u]]);
            if (bias) acc += (double)half_to_float(bias[o]);
            y_out[r * m + o] = (float)acc;
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Reproducible Cholesky factor -- NOT a restatement of anything in the reference: an input generator for the
 * large golden cases (tests/golden/exact_inputs.py).  Those cases hand the reference's own loop
 * (ganq.py:456-646) a lower-triangular L that every box must be able to rebuild bit for bit from a seed; a
 * LAPACK factorisation does not qualify (blocking, threads and instruction set change its rounding).  Here every
 * entry is ONE ascending fp64 sum with separately rounded products (Cholesky-Crout),
 *     L[i][j] = (A[i][j] - sum_{k<j} L[i][k] L[j][k]) / L[j][j],   L[j][j] = sqrt(A[j][j] - sum_{k<j} L[j][k]^2),
 * so the result depends on nothing but A (threads split the rows of a column).  A [n,n] fp32 symmetric (lower
 * triangle read), out [n,n] fp32 lower-triangular (upper part zero).  Returns j+1 when pivot j is not positive.
 * ------------------------------------------------------------------------------------------ */
/* Blocked form (round 4; n = 14336 inputs): the columns are taken DC_B at a time so that a row's DC_B sums share every load
 * of the row -- the chains of different entries are independent, the chain of ONE entry is still the single ascending
 * sequence of separately rounded products and subtractions above, so the bits are those of the plain Crout loop (the
 * fixtures' sha256 of L, made with the plain loop in round 3, are the check).  No FMA: -ffp-contract=off, and the AVX2
 * clone carries no FMA flag. */
#define DC_B 16

/* the sums over k < J0 of one row: t[c] = a[c] - sum_k Li[k] * P[k][c], k ascending (P = the block's rows, transposed) */
__attribute__((target_clones("avx2", "default")))
static void dc_row_partial(const double* Li, const double* P, int64_t J0, double* t) {
    double acc[DC_B];
    for (int c = 0; c < DC_B; ++c) acc[c] = t[c];
    for (int64_t k = 0; k < J0; ++k) {
        const double lik = Li[k];
        const double* Pk = P + k * DC_B;
#pragma omp simd
        for (int c = 0; c < DC_B; ++c) acc[c] -= lik * Pk[c];
    }
    for (int c = 0; c < DC_B; ++c) t[c] = acc[c];
}

int ganq_oracle_det_cholesky(const float* A, int64_t n, float* out) {
    double* Ld = (double*)calloc((size_t)n * (size_t)n, sizeof(double));
    double* P = (double*)malloc((size_t)n * DC_B * sizeof(double));
    if (!Ld || !P) { free(Ld); free(P); return -1; }
    int bad = 0;
    for (int64_t J0 = 0; J0 < n && !bad; J0 += DC_B) {
        const int bw = (int)((n - J0) < DC_B ? (n - J0) : DC_B);
        /* the block's own rows, transposed: P[k][c] = L[J0+c][k], k < J0 (columns beyond bw: zeros) */
#pragma omp parallel for schedule(static)
        for (int64_t k = 0; k < J0; ++k)
            for (int c = 0; c < DC_B; ++c) P[k * DC_B + c] = c < bw ? Ld[(J0 + c) * n + k] : 0.0;
        /* diagonal block: rows J0 .. J0+bw-1, one after the other */
        double d[DC_B];
        for (int r = 0; r < bw && !bad; ++r) {
            const int64_t j = J0 + r;
            double t[DC_B];
            for (int c = 0; c < DC_B; ++c) t[c] = (c <= r) ? (double)A[j * n + J0 + c] : 0.0;
            dc_row_partial(Ld + j * n, P, J0, t);
            double* Lj = Ld + j * n;
            for (int c = 0; c < r; ++c) {          /* L[j][J0+c], c < r: continue the sum over k = J0 .. J0+c-1 */
                double v = t[c];
                const double* Lc = Ld + (J0 + c) * n;
                for (int64_t k = J0; k < J0 + c; ++k) v -= Lj[k] * Lc[k];
                Lj[J0 + c] = v / d[c];
            }
            double s = t[r];
            for (int64_t k = J0; k < j; ++k) s -= Lj[k] * Lj[k];
            if (!(s > 0.0)) { bad = (int)(j + 1); break; }
            d[r] = sqrt(s);
            Lj[j] = d[r];
        }
        if (bad) break;
        /* the rows below the block */
#pragma omp parallel for schedule(dynamic, 16)
        for (int64_t i = J0 + bw; i < n; ++i) {
            double t[DC_B];
            for (int c = 0; c < DC_B; ++c) t[c] = (c < bw) ? (double)A[i * n + J0 + c] : 0.0;
            double* Li = Ld + i * n;
            dc_row_partial(Li, P, J0, t);
            for (int c = 0; c < bw; ++c) {
                double v = t[c];
                const double* Lc = Ld + (J0 + c) * n;
                for (int64_t k = J0; k < J0 + c; ++k) v -= Li[k] * Lc[k];
                Li[J0 + c] = v / d[c];
            }
        }
    }
    if (!bad) {
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < n * n; ++i) out[i] = (float)Ld[i];
    }
    free(Ld);
    free(P);
    return bad;
}

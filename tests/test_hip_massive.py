"""Hessians with the dynamic range of real LLM activations (massive-activation features).

The T-update forms A = S H S^T exactly on the integer matrix cores from a FIXED-POINT copy of H.  With one global scale
(max|H| / 2^30) an entry 10^6 below the maximum keeps ~10 significant bits, while the reference sums fp32 entries
(24 bits each, ganq.py:589-591).  The library therefore extends the fixed point by a 16-bit word when
max|H| > 16 mean(diag H) (47 bits in all; csrc/update_t.hip) -- these tests pin the result against the oracle (fp64 sums of
the fp32 entries) on Hessians whose feature scales span 10^3 (H spans 10^6) and with a handful of 100x outlier features,
for the stage API (ganq_update_t) and for the fused loop (ganq_run_layer), to the same bars as everywhere else.
"""
import numpy as np
import pytest
import torch

from conftest import rel_fro

pytestmark = pytest.mark.gpu

TOL_T = 1e-5
# The loss is a closed form, dist = w^T H w - 2 t^T b + t^T A t.  With a massive feature f the best codebook puts an entry
# ON w_f, so the true loss no longer contains f while each of the three terms does (10^4 x the result): the 3e-7
# relative rounding of the off-diagonal part of (W H)[:, f] (split-fp16 product, wh_gemm.hip) shows as ~1e-5 of a row's
# loss (measured: 1.3e-6 / 6.5e-6 / 2.8e-5 on the three cases).  It averages out over rows in the only place the loss is
# used -- the sum over all rows that picks the best iteration -- and touches neither indices nor codebooks.
TOL_LOSS_ROW = 1e-4
TOL_DIST = 1e-5


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def hessian_with_scales(n, scales, seed, corr=0.0, p=None):
    rng = np.random.default_rng(seed)
    p = p or max(2 * n, 512)
    X = rng.standard_normal((p, n)).astype(np.float32) * scales.astype(np.float32)
    if corr:
        X = X + corr * (X @ (rng.standard_normal((n, n)) / np.sqrt(n)).astype(np.float32))
    H = (2.0 / p) * (X.T.astype(np.float64) @ X.astype(np.float64))
    H += 0.01 * np.mean(np.diag(H)) * np.eye(n)  # gptq.py:296-298
    H = H.astype(np.float32)
    Hd = H.astype(np.float64)
    off = np.clip(np.abs(Hd).sum(1) - 2 * np.diag(Hd), 1e-8, None)  # gptq.py:289-291
    L = np.linalg.cholesky(Hd + np.diag(off)).astype(np.float32)
    return H, L


def scale_cases(n, seed):
    rng = np.random.default_rng(seed)
    log_uniform = 10.0 ** rng.uniform(-1.5, 1.5, n)              # feature scales spanning 10^3, H spanning 10^6
    outliers = 0.1 + rng.random(n)
    idx = rng.choice(n, size=5, replace=False)
    outliers[idx] *= 100.0                                        # a handful of 100x features (H_ii 10^4 x)
    massive = 0.1 + rng.random(n)
    massive[rng.choice(n, size=2, replace=False)] *= 1000.0       # two massive features (H_ii 10^6 x)
    return {"log_uniform_1e3": log_uniform, "five_100x_outliers": outliers, "two_1000x_massive": massive}


CASES = ["log_uniform_1e3", "five_100x_outliers", "two_1000x_massive"]


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("m,n,V", [(48, 1024, 16), (32, 768, 8)])
def test_update_t_stage_wide_range_hessian(case, m, n, V):
    from ganq_amd import _lib
    from oracle import c_oracle

    _lib.selftest()
    H, L = hessian_with_scales(n, scale_cases(n, 11)[case], seed=n + V)
    rng = np.random.default_rng(3)
    W = (0.02 * rng.standard_normal((m, n))).astype(np.float16).astype(np.float32)
    T0 = np.quantile(W, (np.arange(V) + 0.5) / V, axis=1).T.astype(np.float32).copy()
    Q = c_oracle.solve_s(W, L, T0)
    WH = c_oracle.matmul(W, H)
    To, Ao, bo = c_oracle.update_t(WH, H, Q, V, want_ab=True)
    T, A, b = _lib.update_t(dev(WH), dev(H), dev(Q), V, want_ab=True)
    # A against the oracle's fp64 bucket sums of the fp32 entries, rounded to fp32 like the reference holds it
    assert rel_fro(A.cpu().numpy(), Ao) < 1e-6, case
    assert rel_fro(b.cpu().numpy(), bo) < 1e-6, case
    assert rel_fro(T.cpu().numpy(), To) < TOL_T, case


def exact_codebooks(W, H, Q, V):
    """the T-update without any fp32 rounding: A = S H S^T and b = S (W H)^T in fp64, minimum-norm solve in fp64"""
    m, n = W.shape
    Hd, Wd = H.astype(np.float64), W.astype(np.float64)
    WH = Wd @ Hd
    T = np.zeros((m, V))
    for i in range(m):
        S = np.zeros((V, n))
        S[Q[i], np.arange(n)] = 1.0
        T[i] = np.linalg.lstsq(S @ Hd @ S.T, S @ WH[i], rcond=1.1920929e-07 * V)[0]
    return T


@pytest.mark.parametrize("case", CASES)
def test_run_layer_wide_range_hessian(case):
    """The fused loop, every iteration checked STAGE BY STAGE against the oracle fed with the GPU's own inputs (indices
    bit-exact given the GPU's previous codebook; codebook and per-row loss given the GPU's indices), plus the
    free-running comparison for the record.

    Tolerance of the codebook.  The reference holds A = S H S^T and b in fp32 (ganq.py:589-591); so do the oracle and the
    GPU path, which round fp64-exact sums.  With a massive feature f, A[a][a] ~ H_ff for its bucket, and one fp32 step
    of it (6e-8 relative) is 1e-3 of what the bucket's other ~64 members contribute: two correct evaluations that
    differ by 1e-9 before the rounding land one step apart now and then, and the solutions of the two rounded systems
    can differ by more than either differs from the exact one -- the same effect SURVEY.md section 7 measured on the
    reference itself (fp32 vs fp64: 4e-4..3e-3 on real activations).  So the bar is 1e-5 against the oracle where the
    rounding is harmless (the first two cases) and, for the massive case, 1e-4 against the oracle AND 1e-5 against the
    fp64-exact codebooks (measured: GPU 2.9e-6, oracle 7.8e-7; the GPU's share comes from the 3e-7 rounding of W H,
    which the coupling to the massive feature magnifies ten times)."""
    from ganq_amd import _lib
    from oracle import c_oracle

    _lib.selftest()
    m, n, V, K = 64, 1024, 16, 6
    H, L = hessian_with_scales(n, scale_cases(n, 5)[case], seed=77, corr=0.05)
    rng = np.random.default_rng(9)
    W = (0.02 * rng.standard_normal((m, n))).astype(np.float16).astype(np.float32)
    T0 = c_oracle.kmeans_init(W, None, V)
    tr = c_oracle.run_layer_trace(W, H, L, T0, K)
    rec = _lib.run_layer_rows(dev(W), dev(H), dev(L), dev(T0), K, alias_q=True, want_q_all=True)
    torch.cuda.synchronize()
    Qg, Tg, lg = rec["Q_all"].cpu().numpy(), rec["T_all"].cpu().numpy(), rec["loss_rows_all"].cpu().numpy()
    massive = case == "two_1000x_massive"
    tol_t = 1e-4 if massive else TOL_T
    WH = c_oracle.matmul(W, H)
    free_running_flips = []
    for k in range(K):
        Tprev = T0 if k == 0 else Tg[k - 1]
        assert np.array_equal(c_oracle.solve_s(W, L, Tprev), Qg[k]), f"{case}: iteration {k}: indices given the GPU's codebook"
        Ts = c_oracle.update_t(WH, H, Qg[k], V)
        e = rel_fro(Tg[k], Ts)
        assert e < tol_t, f"{case}: codebook of iteration {k} rel. Frobenius {e:.3e}"
        if massive and k in (0, K - 1):
            Tx = exact_codebooks(W, H, Qg[k], V)
            e_gpu, e_orc = rel_fro(Tg[k], Tx), rel_fro(Ts, Tx)
            print(f"[{case}] iteration {k}: vs fp64-exact codebooks: GPU {e_gpu:.3e}, oracle (fp32-held A, b) {e_orc:.3e}; GPU vs oracle {e:.3e}")
            assert e_gpu < TOL_T  # within the north-star tolerance of the exact solution, like the oracle
        _, rows_o = c_oracle.quad_loss(W, H, Tg[k], Qg[k], want_rows=True)
        el = np.abs(lg[k] - rows_o).max() / np.abs(rows_o).max()
        assert el < TOL_LOSS_ROW, f"{case}: per-row loss of iteration {k} differs by {el:.3e}"
        assert abs(lg[k].sum() - rows_o.sum()) < TOL_DIST * rows_o.sum()
        if not np.array_equal(Qg[k], tr["Q_all"][k]):
            free_running_flips.append((k, int((Qg[k] != tr["Q_all"][k]).sum())))
    print(f"[{case}] free-running oracle: indices differing per iteration {free_running_flips or 'none'}; "
          f"best_k GPU {int(rec['best_k'])} oracle {tr['best_k']}")
    if not massive:
        assert not free_running_flips or sum(c for _, c in free_running_flips) <= 4 * K
        assert np.allclose(rec["dists"].cpu().numpy(), tr["dists"], rtol=1e-4)


def test_wh_product_wide_range_hessian():
    """W @ H_fixed of the fused driver against fp64 on a two-massive-feature Hessian (the fp16 split works on the
    symmetrically scaled matrix, so small rows of H keep their precision)"""
    from ganq_amd import _lib

    m, n = 128, 1024
    H, _ = hessian_with_scales(n, scale_cases(n, 5)["two_1000x_massive"], seed=5)
    rng = np.random.default_rng(1)
    W = (0.02 * rng.standard_normal((m, n))).astype(np.float32)
    WH, Hf = _lib.debug_wh_product(dev(W), dev(H))
    ref = W.astype(np.float64) @ H.astype(np.float64)
    # column-wise: every column of W @ H (one per input feature, whatever that feature's scale) is accurate
    err = np.linalg.norm(WH.cpu().numpy() - ref, axis=0) / np.linalg.norm(ref, axis=0)
    assert err.max() < 2e-6, err.max()
    # the fixed-point H itself: entries relative to sqrt(H_uu H_vv), i.e. as correlations
    dsq = np.sqrt(np.diag(H).astype(np.float64))
    rel = np.abs(Hf.cpu().numpy() - H.astype(np.float64)) / np.outer(dsq, dsq)
    assert rel.max() < 1e-6, rel.max()

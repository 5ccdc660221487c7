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


@pytest.mark.parametrize("case", CASES)
def test_run_layer_wide_range_hessian(case):
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
    # the very first S-solve sees the same T0: bit-exact whatever the T-update does
    assert np.array_equal(Qg[0], tr["Q_all"][0])
    for k in range(K):
        e = rel_fro(Tg[k], tr["T_all"][k])
        assert e < TOL_T, f"{case}: codebook of iteration {k} rel. Frobenius {e:.3e}"
        assert np.array_equal(Qg[k], tr["Q_all"][k]), f"{case}: indices of iteration {k} differ ({int((Qg[k] != tr['Q_all'][k]).sum())})"
        el = np.abs(lg[k] - tr["loss_rows_all"][k]).max() / np.abs(tr["loss_rows_all"][k]).max()
        assert el < TOL_LOSS_ROW, f"{case}: per-row loss of iteration {k} differs by {el:.3e}"
    d = rec["dists"].cpu().numpy()
    assert np.allclose(d, tr["dists"], rtol=TOL_DIST)
    assert int(rec["best_k"]) == tr["best_k"]


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

"""Parity at the BASELINE size (4096 x 4096, V = 16) through properties that do not need the oracle to run the full
problem: rows of W are independent in every stage, so a sample of rows is checked bit-for-bit / to tolerance against
the CPU oracle run on just those rows; the loss is checked against an fp64 evaluation of its definition on the GPU."""
import numpy as np
import pytest
import torch

from conftest import rel_fro

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def layer():
    from ganq_amd import _lib

    m = n = 4096
    V = 16
    g = torch.Generator(device="cuda").manual_seed(0)
    W = (0.02 * torch.randn(m, n, device="cuda", generator=g)).half().float()
    X = torch.randn(4 * n, n, device="cuda", generator=g) * (0.1 + torch.rand(n, device="cuda", generator=g))
    H = (2.0 / X.shape[0]) * (X.T @ X)
    H += 0.01 * H.diag().mean() * torch.eye(n, device="cuda")
    H = 0.5 * (H + H.T)
    off = (H.abs().sum(1) - 2 * H.diag()).clamp(min=1e-8)
    L = torch.linalg.cholesky(H + torch.diag(off))
    T0 = torch.quantile(W, (torch.arange(V, device="cuda") + 0.5) / V, dim=1).T.contiguous()
    _lib.selftest()
    return dict(W=W, H=H, L=L, T0=T0, V=V)


ROWS = [0, 1, 15, 16, 17, 1000, 2047, 2048, 3333, 4080, 4094, 4095]


def test_solve_s_full_size_sampled_rows_bit_exact(layer):
    from ganq_amd import _lib
    from oracle import c_oracle

    Q = _lib.solve_s(layer["W"], layer["L"], layer["T0"])
    layer["Q"] = Q
    rows = torch.tensor(ROWS, device="cuda")
    Qo = c_oracle.solve_s(layer["W"][rows].cpu().numpy(), layer["L"].cpu().numpy(), layer["T0"][rows].cpu().numpy())
    assert np.array_equal(Q[rows].cpu().numpy(), Qo)
    # every index in range, every row uses most of its codebook
    assert int(Q.max()) < layer["V"]
    assert int(torch.stack([(Q == v).any(1) for v in range(layer["V"])]).sum(0).min()) >= 12


def test_update_t_and_loss_full_size(layer):
    from ganq_amd import _lib
    from oracle import c_oracle

    W, H, V = layer["W"], layer["H"], layer["V"]
    Q = layer.get("Q")
    if Q is None:
        Q = _lib.solve_s(W, layer["L"], layer["T0"])
    WH = _lib.matmul_f32(W, H)
    assert float((WH - W @ H).norm() / (W @ H).norm()) < 1e-6
    T = _lib.update_t(WH, H, Q, V)
    rows = torch.tensor(ROWS, device="cuda")
    To = c_oracle.update_t(WH[rows].cpu().numpy(), H.cpu().numpy(), Q[rows].cpu().numpy(), V)
    assert rel_fro(T[rows].cpu().numpy(), To) < 1e-5
    # loss: GEMM-based stage kernel and the closed form inside run_layer both equal the fp64 definition
    E = (W - T.gather(1, Q.long())).double()
    ref = float(((E @ H.double()) * E).sum())
    d_stage = float(_lib.quad_loss(W, H, T, Q))
    assert abs(d_stage - ref) < 1e-6 * ref
    T1, Q1, dists, best_k = _lib.run_layer(W, H, layer["L"], layer["T0"], 1)
    assert torch.equal(Q1, Q)
    assert abs(float(dists[0]) - ref) < 1e-6 * ref
    assert float((T1 - T).norm() / T.norm()) < 1e-6


def test_run_layer_full_size_is_deterministic_and_improves(layer):
    from ganq_amd import _lib

    a = _lib.run_layer(layer["W"], layer["H"], layer["L"], layer["T0"], 3)
    b = _lib.run_layer(layer["W"], layer["H"], layer["L"], layer["T0"], 3)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])  # bit-reproducible
    d = a[2].cpu().numpy()
    assert np.all(np.isfinite(d)) and d[-1] < d[0]
    assert int(a[3]) == int(np.argmin(d))


# ------------------------------------------------------------------------------------------------------------------
# Llama-class in_features (down_proj of a 7B model: n = 11008 -- not a power of two, 172 panels): every stage against
# the oracle on a handful of rows
@pytest.fixture(scope="module")
def wide_layer():
    from ganq_amd import _lib

    m, n, V = 24, 11008, 16
    g = torch.Generator(device="cuda").manual_seed(1)
    W = (0.02 * torch.randn(m, n, device="cuda", generator=g)).half().float()
    X = torch.randn(2 * n, n, device="cuda", generator=g) * (0.1 + torch.rand(n, device="cuda", generator=g))
    H = (2.0 / X.shape[0]) * (X.T @ X)
    del X
    H += 0.01 * H.diag().mean() * torch.eye(n, device="cuda")
    H = 0.5 * (H + H.T)
    off = (H.abs().sum(1) - 2 * H.diag()).clamp(min=1e-8)
    L = _lib.cholesky(H + torch.diag(off))
    T0 = torch.quantile(W, (torch.arange(V, device="cuda") + 0.5) / V, dim=1).T.contiguous()
    return dict(W=W, H=H, L=L, T0=T0, V=V, m=m, n=n)


def test_wide_layer_solve_and_update_vs_oracle(wide_layer):
    from ganq_amd import _lib
    from oracle import c_oracle

    d = wide_layer
    Wn, Hn, Ln, Tn = (d[k].cpu().numpy() for k in ("W", "H", "L", "T0"))
    Q = _lib.solve_s(d["W"], d["L"], d["T0"])
    Qo = c_oracle.solve_s(Wn, Ln, Tn)
    assert np.array_equal(Q.cpu().numpy(), Qo)
    WH = _lib.matmul_f32(d["W"], d["H"])
    T1 = _lib.update_t(WH, d["H"], Q, d["V"])
    T1o = c_oracle.update_t(WH.cpu().numpy(), Hn, Qo, d["V"])
    assert rel_fro(T1.cpu().numpy(), T1o) < 1e-5


def test_wide_layer_kmeans_and_loop(wide_layer):
    from ganq_amd import _lib
    from oracle import c_oracle

    d = wide_layer
    g = torch.Generator(device="cuda").manual_seed(2)
    cw = (torch.rand(d["n"], device="cuda", generator=g, dtype=torch.float64) + 0.5) ** 4
    rows = slice(0, 3)
    T0 = _lib.kmeans_init(d["W"][rows], cw, d["V"])  # windowed kernel (n > 4.7 k)
    T0o = c_oracle.kmeans_init(d["W"][rows].cpu().numpy(), cw.cpu().numpy(), d["V"])
    assert rel_fro(T0.cpu().numpy(), T0o) < 1e-6
    # the whole loop, incremental bucket sums included, is deterministic and improves the loss
    a = _lib.run_layer(d["W"], d["H"], d["L"], d["T0"], 3)
    b = _lib.run_layer(d["W"], d["H"], d["L"], d["T0"], 3)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    dist = a[2].cpu().numpy()
    assert dist[-1] < dist[0]

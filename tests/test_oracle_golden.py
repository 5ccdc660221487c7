"""Pins the CPU oracle (oracle/) against golden vectors captured from the reference itself
(tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from conftest import rel_fro
from oracle import c_oracle, ganq_ref

# north_star tolerance: indices bit-exact, codebooks / reconstructed weights 1e-5 relative Frobenius
TOL_T = 1e-5
TOL_LOSS = 1e-6


def test_solve_s_bit_exact(golden):
    g = golden
    for k in range(int(g["K"])):
        Q = c_oracle.solve_s(g["W_perm"], g["L"], g["T"][k])
        assert np.array_equal(Q, g["Q"][k]), f"iteration {k}: {(Q != g['Q'][k]).sum()} index mismatches"


def test_solve_s_err_output(golden):
    g = golden
    Q, Err = c_oracle.solve_s(g["W_perm"], g["L"], g["T"][0], want_err=True)
    ref = g["W_perm"] - np.take_along_axis(g["T"][0], Q.astype(np.int64), axis=1)
    assert np.array_equal(Err, ref)


def test_update_t(golden):
    g = golden
    V = 2 ** int(g["bits"])
    WH = c_oracle.matmul(g["W_perm"], g["Xxt_damped"])
    for k in range(int(g["K"])):
        T, A, b = c_oracle.update_t(WH, g["Xxt_damped"], g["Q"][k], V, want_ab=True)
        assert rel_fro(A, g["A"][k]) < 1e-6
        assert rel_fro(b, g["B"][k]) < 1e-6
        assert rel_fro(T, g["T"][k + 1]) < TOL_T


def test_minnorm_unused_entry_is_zero():
    # SURVEY 7 hard part 1: an unused codebook entry (zero row/col) must come out as 0 (min-norm)
    rng = np.random.default_rng(0)
    V = 8
    M = rng.standard_normal((V, V))
    A = (M @ M.T).astype(np.float32)
    A[3, :] = 0
    A[:, 3] = 0
    b = rng.standard_normal(V).astype(np.float32)
    b[3] = 0
    T = c_oracle.minnorm_solve(A[None], b[None])
    ref = torch.linalg.lstsq(torch.from_numpy(A)[None], torch.from_numpy(b)[None, :, None], driver="gelsd").solution
    assert abs(T[0, 3]) < 1e-7
    assert rel_fro(T[0], ref[0, :, 0].numpy()) < 1e-5


def test_quad_loss(golden):
    g = golden
    for k in range(int(g["K"])):
        d = c_oracle.quad_loss(g["W_perm"], g["Xxt_damped"], g["T"][k + 1], g["Q"][k])
        assert abs(d - g["dists"][k]) <= TOL_LOSS * abs(g["dists"][k])


def test_run_layer_end_to_end(golden):
    g = golden
    K = int(g["K"])
    T, Q, dists, best_k = c_oracle.run_layer(g["W_perm"], g["Xxt_damped"], g["L"], g["T"][0], K, alias_q=True)
    assert best_k == int(np.argmin(g["dists"]))
    assert np.allclose(dists, g["dists"], rtol=1e-5)
    # reference quirk (ganq.py:487,550,625-626): indices of the LAST iteration, codebook of the BEST
    assert np.array_equal(Q, g["Q"][K - 1])
    Wq, Losses = c_oracle.dequant_losses(g["W_perm"], T, Q, g["Hinv_diag"])
    assert rel_fro(Wq, g["Wq_loop"]) < TOL_T
    assert rel_fro(Losses, g["Losses"]) < 1e-4


def test_torch_restatement_matches_reference(golden):
    g = golden
    if int(g["n"]) > 256:
        pytest.skip("op-sequence restatement checked on the small cases")
    W, H, L = (torch.from_numpy(g[k]) for k in ("W_perm", "Xxt_damped", "L"))
    K = int(g["K"])
    T, Q, dists, best_k = ganq_ref.run_layer(W, H, L, torch.from_numpy(g["T"][0]), K, alias_q=True)
    assert np.array_equal(Q.numpy().astype(np.uint8), g["Q"][K - 1])
    assert best_k == int(np.argmin(g["dists"]))
    assert rel_fro(T.numpy(), g["T"][best_k + 1]) < TOL_T


def test_lut_linear_oracle(golden):
    g = golden
    K = int(g["K"])
    best_k = int(np.argmin(g["dists"]))
    perm = g["perm"]
    invperm = np.argsort(perm)
    Q = g["Q"][K - 1]
    if bool(g["desc_act"]):
        Q = Q[:, invperm]
    lut = g["T"][best_k + 1].astype(np.float16)
    y = c_oracle.lut_linear(g["x_fwd"], Q, lut, g["bias"])
    if bool(g["desc_act"]) or str(g["act_sort"]) == "none":
        # fp16 F.linear on CPU accumulates in fp32 and rounds once
        assert np.allclose(y.astype(np.float16).astype(np.float32), g["y_fwd"].astype(np.float32), rtol=2e-3, atol=2e-3)


def test_oracle_quantizer_object_reproduces_reference_seven_tuple(golden):
    """tests/oracle_quantizer.py (the CPU quantizer object the model-boundary GPU tests compare the HIP path with) against
    the reference's own quantize() output: indices exact, weight / avg_loss to rounding"""
    import torch
    import torch.nn as nn

    from ganq_amd.looper.named_module import NamedModule
    from ganq_amd.quantization import QuantizeConfig
    from oracle_quantizer import OracleGANQ

    g = golden
    m, n, K = int(g["m"]), int(g["n"]), int(g["K"])
    lin = nn.Linear(n, m, bias=True).half()
    with torch.no_grad():
        lin.weight.copy_(torch.from_numpy(g["W"]))
    qcfg = QuantizeConfig(bits=int(g["bits"]), quant_method="ganq", format="fake", act_sort=str(g["act_sort"]),
                          l_damp_style=str(g["l_damp_style"]), dead=str(g["dead"]), desc_act=bool(g["desc_act"]),
                          ganq_iterations=K, group_size=int(g["group_size"]), damp_percent=0.01)
    q = OracleGANQ(NamedModule(lin, "fc1", "model.layers.0.fc1", 0), qcfg)
    q.quantizer.configure(perchannel=True)
    for xb in g["X"]:
        q.add_batch(torch.from_numpy(xb), None)
    assert rel_fro(q.H.numpy(), g["H_raw"]) < 1e-6
    wq, scale, zero, g_idx, _, avg_loss, damp = q.quantize()
    Qref = g["Q"][K - 1]
    if bool(g["desc_act"]) and str(g["act_sort"]) != "none":
        Qref = Qref[:, np.argsort(g["perm"])]
    assert np.array_equal(q.ganq_indices.numpy(), Qref)
    diff = wq.float().numpy() - g["Wq"].astype(np.float32)
    assert np.all(np.abs(diff) <= np.spacing(np.abs(g["Wq"]).astype(np.float16)).astype(np.float32))
    assert abs(avg_loss - float(g["avg_loss"])) < 1e-5 * float(g["avg_loss"])
    assert np.array_equal(g_idx.numpy(), g["g_idx"].reshape(-1))

"""The reference's own loop at n = 2048 and on a 4096 x 4096 layer (tests/golden/large/*.npz, make_golden_large.py):
CPU oracle here (`-m "not gpu"`), the HIP path below (`-m gpu`).  Inputs are rebuilt bit for bit from the seed
(tests/golden/exact_inputs.py; the fixture holds their sha256).  Bars: indices bit-exact against the REFERENCE's captured
`torch.argmin` results, stage-wise on every iteration and free-running over all K; codebooks 1e-5, distances 1e-6."""
import glob
import os
import sys

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR, rel_fro

sys.path.insert(0, GOLDEN_DIR)
import exact_inputs  # noqa: E402

LARGE = sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "large", "*.npz")))
TOL_T, TOL_LOSS = 1e-5, 1e-6
_cache = {}

# Free-running budgets (round 4: per case, about twice what was measured, instead of one formula 3-20 x looser).
# Stage-wise -- the reference's codebook in, its indices out -- every case is bit-exact, oracle and HIP alike, with no budget.
# Free-running, iteration k solves against the path's OWN codebook, which agrees with the reference's to ~1e-7 (the reference
# holds A, b in fp32 and solves by SVD; the oracle rounds A, b to fp32 and solves in fp64; the HIP path keeps A exact and solves
# in fp64): a near-tie of the next S-solve -- about one index in 10^6 .. 10^7 -- may then fall the other way, and the row is on
# a trajectory of its own from there.  What is bounded is the number of rows LEAVING the reference's trajectory: (most in any
# one iteration, total over the K iterations).  Measured (MI355X, round 4; DESIGN.md section 2 lists the vectors):
#   h4096x4096_b4_k2 [0, 19]; h3072x768_b4_k10 [0,4,1,0,3,1,0,0,1,1]; h768x3072_b4_k10 [0,3,8,1,1,2,0,2,2,1];
#   h768x768_b4_k10 [0,1,2,0,1,1,0,0,0,0]; h1024x4096_b3_k2, l64x8192_b4_k3, l32x14336_b4_k2, l32x14336_b3_k2: see HIP_LEAVING
HIP_LEAVING = {
    "h4096x4096_b4_k2": (40, 40), "h3072x768_b4_k10": (8, 22), "h768x3072_b4_k10": (16, 40), "h768x768_b4_k10": (4, 10),
    "h1024x4096_b3_k2": (12, 12), "l64x8192_b4_k3": (2, 2), "l32x14336_b4_k2": (2, 2), "l32x14336_b3_k2": (2, 2),
}
# the CPU oracle, free-running: rows leaving (the small traced cases: none); measured l32x14336_b4_k2 [0, 1]
ORACLE_LEAVING = {"l64x8192_b4_k3": (1, 1), "l32x14336_b4_k2": (1, 2), "l32x14336_b3_k2": (1, 1)}


def case(name):
    if name not in _cache:
        _cache.clear()  # one case in memory at a time (the 4096 x 4096 inputs are 200 MB)
        fx = np.load(os.path.join(GOLDEN_DIR, "large", name + ".npz"))
        inp = exact_inputs.make(int(fx["m"]), int(fx["n"]), int(fx["bits"]), int(fx["seed"]), int(fx["tokens"]))
        if "nan_entries" in fx:  # the NaN-semantics case: the same codebook entries are NaN as in the reference's run
            for r, e in np.asarray(fx["nan_entries"]):
                inp["T0"][int(r), int(e)] = np.nan
        exact_inputs.check(inp, fx)
        assert np.array_equal(inp["T0"], fx["T0"], equal_nan=True)
        _cache[name] = (fx, inp)
    return _cache[name]


def test_large_cases_present():
    assert {"l128x2048_b4_k3", "l128x2048_b3_k3", "l128x2048_b4_k10", "l256x512_b2_k3", "h4096x4096_b4_k2",
            "h3072x768_b4_k10", "h768x3072_b4_k10", "h768x768_b4_k10", "nan48x256_b4_k1",
            "l64x8192_b4_k3", "l32x14336_b4_k2", "l32x14336_b3_k2", "h1024x4096_b3_k2"} <= set(LARGE)


# ------------------------------------------------------------------------------------------ CPU oracle
@pytest.mark.parametrize("name", [n for n in LARGE if n.startswith("l")])
def test_oracle_vs_reference_large(name):
    from oracle import c_oracle

    fx, inp = case(name)
    K, V = int(fx["K"]), 2 ** int(fx["bits"])
    Qs = exact_inputs.unpack_q_trace(fx)
    WH = c_oracle.matmul(inp["W"], inp["H"])
    for k in range(K):  # stage-wise: the reference's codebook in, the reference's indices / next codebook out
        Q = c_oracle.solve_s(inp["W"], inp["L"], fx["T"][k])
        assert np.array_equal(Q, Qs[k]), f"{name} iteration {k}: {(Q != Qs[k]).sum()} index mismatches vs the reference"
        assert exact_inputs.sha(Q) == str(fx["sha_Q"][k])
        assert rel_fro(c_oracle.update_t(WH, inp["H"], Qs[k], V), fx["T"][k + 1]) < TOL_T
        d = c_oracle.quad_loss(inp["W"], inp["H"], fx["T"][k + 1], Qs[k])
        assert abs(d - fx["dists"][k]) <= TOL_LOSS * abs(fx["dists"][k])
    tr = c_oracle.run_layer_trace(inp["W"], inp["H"], inp["L"], inp["T0"], K)  # free-running
    flips = [int((tr["Q_all"][k] != Qs[k]).sum()) for k in range(K)]
    differ = np.stack([(tr["Q_all"][k] != Qs[k]).any(axis=1) for k in range(K)])
    left = np.zeros(differ.shape[1], dtype=bool)
    leaving = []
    for k in range(K):
        leaving.append(int((differ[k] & ~left).sum()))
        left |= differ[k]
    print(f"{name}: oracle free-running vs the reference: index flips per iteration {flips}, rows leaving its trajectory {leaving}")
    per_it, total = ORACLE_LEAVING.get(name, (0, 0))
    assert max(leaving) <= per_it and sum(leaving) <= total, f"{name}: rows leaving the reference's trajectory {leaving}"
    assert not differ[0].any()  # the first solve starts from the reference's own T0
    clean = ~left
    assert all(rel_fro(tr["T_all"][k][clean], fx["T"][k + 1][clean]) < TOL_T for k in range(K))
    assert np.allclose(tr["dists"], fx["dists"], rtol=TOL_LOSS if not left.any() else 1e-4)
    best = int(np.argmin(fx["dists"]))
    Wq, Lo = c_oracle.dequant_losses(inp["W"], fx["T"][best + 1], Qs[K - 1], inp["hinv_diag"])
    assert exact_inputs.sha(Wq) == str(fx["sha_Wq"])
    assert abs(float(Lo.astype(np.float64).sum()) - float(fx["losses_sum"])) <= 1e-5 * float(fx["losses_sum"])


@pytest.mark.parametrize("name", [n for n in LARGE if n.startswith("h")])
def test_oracle_vs_reference_hash_cases_sampled_rows(name):
    """CPU: up to 192 rows of the hash-only cases -- the 4096 x 4096 layer and the three module shapes of opt-125m with all
    K = 10 iterations (rows are independent in the S-solve and the T-update); the whole layers are the GPU test's"""
    from oracle import c_oracle

    fx, inp = case(name)
    m, V = int(fx["m"]), 2 ** int(fx["bits"])
    rows = np.unique(np.r_[0:64, m // 2:m // 2 + 64, m - 64:m])
    WH = c_oracle.matmul(inp["W"][rows], inp["H"])
    for k in range(int(fx["K"])):
        Q = c_oracle.solve_s(inp["W"][rows], inp["L"], fx["T"][k][rows])
        assert np.array_equal(exact_inputs.row_digest(Q), fx["Q_row_digest"][k][rows]), f"{name} iteration {k}"
        assert rel_fro(c_oracle.update_t(WH, inp["H"], Q, V), fx["T"][k + 1][rows]) < TOL_T


# ------------------------------------------------------------------------------------------ HIP path
def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.fixture(scope="module")
def hip():
    from ganq_amd import _lib

    assert torch.cuda.is_available(), "these tests need the MI355X"
    _lib.selftest()
    return _lib


def test_oracle_nan_semantics_vs_reference():
    """torch.argmin in the reference's S-solve (ganq.py:547) returns the FIRST NaN when a distance is NaN -- a NaN codebook
    entry wins its column, and the NaN residual it leaves makes every later column of that row pick index 0.  Pinned by the
    reference's own indices on a codebook with three NaN entries."""
    from oracle import c_oracle

    fx, inp = case("nan48x256_b4_k1")
    Q = c_oracle.solve_s(inp["W"], inp["L"], inp["T0"])
    assert np.array_equal(Q, fx["Q_first"]), f"{(Q != fx['Q_first']).sum()} index mismatches vs the reference"
    n = int(fx["n"])
    assert Q[3, n - 1] == 6 and not Q[3, : n - 1].any() and not Q[10].any()  # what the semantics amounts to


@pytest.mark.gpu
def test_hip_nan_semantics_vs_reference(hip):
    fx, inp = case("nan48x256_b4_k1")
    W, L, T0 = dev(inp["W"]), dev(inp["L"]), dev(inp["T0"])
    for variant in (0, 1):  # threshold path with its fallback / reductions only
        hip.debug_option("GANQ_SOLVE_VARIANT", variant)
        try:
            Q = hip.solve_s(W, L, T0).cpu().numpy()
        finally:
            hip.debug_option("GANQ_SOLVE_VARIANT", None)
        assert np.array_equal(Q, fx["Q_first"]), f"variant {variant}: {(Q != fx['Q_first']).sum()} index mismatches vs the reference"


@pytest.mark.gpu
@pytest.mark.parametrize("name", [n for n in LARGE if n[0] in "lh"])
def test_hip_vs_reference_large(hip, name):
    fx, inp = case(name)
    K, V, m = int(fx["K"]), 2 ** int(fx["bits"]), int(fx["m"])
    hash_only = bool(fx["hash_only"])
    Qs = None if hash_only else exact_inputs.unpack_q_trace(fx)
    W, H, L = dev(inp["W"]), dev(inp["H"]), dev(inp["L"])
    WH = hip.matmul_f32(W, H)
    Q_ref_dev = []
    for k in range(K):  # stage-wise against the reference
        Qd = hip.solve_s(W, L, dev(fx["T"][k]))
        Q = Qd.cpu().numpy()
        if hash_only:
            bad = int((exact_inputs.row_digest(Q) != fx["Q_row_digest"][k]).sum())
            assert bad == 0, f"{name} iteration {k}: {bad} of {m} rows differ from the reference's indices"
            assert exact_inputs.sha(Q) == str(fx["sha_Q"][k])
        else:
            assert np.array_equal(Q, Qs[k]), f"{name} iteration {k}: {(Q != Qs[k]).sum()} index mismatches vs the reference"
        Q_ref_dev.append(Qd)  # == the reference's Q_k, just verified
        T = hip.update_t(WH, H, Qd, V)
        assert rel_fro(T.cpu().numpy(), fx["T"][k + 1]) < TOL_T
        d = float(hip.quad_loss(W, H, dev(fx["T"][k + 1]), Qd).cpu())
        assert abs(d - fx["dists"][k]) <= TOL_LOSS * abs(fx["dists"][k])
    # free-running: the fused driver, K iterations on its own codebooks
    rec = hip.run_layer_rows(W, H, L, dev(inp["T0"]), K, alias_q=True, want_q_all=True)
    torch.cuda.synchronize()
    differ = np.stack([(rec["Q_all"][k] != Q_ref_dev[k]).any(dim=1).cpu().numpy() for k in range(K)])  # [K, m] rows off the reference
    flips = [int((rec["Q_all"][k] != Q_ref_dev[k]).sum()) for k in range(K)]
    left = np.zeros(m, dtype=bool)  # rows that have left the reference's trajectory so far
    leaving = []
    for k in range(K):
        leaving.append(int((differ[k] & ~left).sum()))
        left |= differ[k]
    print(f"{name}: HIP free-running vs the reference: rows leaving its trajectory per iteration {leaving} of {m}; "
          f"index flips per iteration {flips} of {m * int(fx['n'])}")
    # Free-running, iteration k solves against the path's OWN codebook T_k, which agrees with the reference's to ~1e-7
    # (the reference holds A, b in fp32 and solves by SVD; here A is exact and the solve fp64): a near-tie -- about one index
    # in 10^6..10^7 -- may then fall the other way.  From there the row is on a trajectory of its own (the flip moves the
    # residual of every later column, the next codebook, ...), so what is bounded is the number of rows LEAVING the
    # reference's trajectory per iteration: none on the small traced cases; elsewhere the case's budget (HIP_LEAVING: twice
    # what was measured).  Every row off the reference, in every
    # iteration, must be exactly what the CPU oracle computes from the path's own previous codebook -- so a difference is the
    # codebook's rounding, never the solve -- and the rows still on the trajectory keep the reference's codebooks.
    per_it, total = HIP_LEAVING.get(name, (0, 0))
    assert max(leaving) <= per_it and sum(leaving) <= total, \
        f"{name}: rows leaving the reference's trajectory {leaving} exceed the case's budget ({per_it} per iteration, {total} in all)"
    if left.any():
        from oracle import c_oracle

        assert not differ[0].any()  # the first solve starts from the reference's own T0
        for k in range(1, K):
            rows = np.nonzero(differ[k])[0]
            if rows.size:
                Qo = c_oracle.solve_s(inp["W"][rows], inp["L"], rec["T_all"][k - 1].cpu().numpy()[rows])
                assert np.array_equal(Qo, rec["Q_all"][k].cpu().numpy()[rows]), f"{name}: iteration {k}, rows off the reference are not the oracle's"
    clean = np.ones(m, dtype=bool)  # rows still on the trajectory: their codebooks must track the reference's
    for k in range(K):
        clean &= ~differ[k]
        assert rel_fro(rec["T_all"][k].cpu().numpy()[clean], fx["T"][k + 1][clean]) < TOL_T
    dev_d = float(np.max(np.abs(rec["dists"].cpu().numpy() - fx["dists"]) / np.abs(fx["dists"])))
    print(f"{name}: rows off the reference's trajectory at the end: {int(left.sum())} of {m}; codebooks of the other rows within "
          f"{TOL_T}; layer loss per iteration within {dev_d:.2e}")
    assert dev_d <= (1e-4 if left.any() else TOL_LOSS)
    T, Q, dists, best_k = hip.run_layer(W, H, L, dev(inp["T0"]), K, alias_q=True)
    assert int(best_k) == int(np.argmin(fx["dists"]))
    assert torch.equal(Q, rec["Q_all"][K - 1])
    if not left.any():
        assert exact_inputs.sha(Q.cpu().numpy()) == str(fx["sha_Q"][K - 1])
    Wq, Lo = hip.dequant_losses(W, T, Q, dev(inp["hinv_diag"]))
    Wq_ref = np.take_along_axis(fx["T"][int(best_k) + 1], Q.cpu().numpy().astype(np.int64), axis=1)
    assert rel_fro(Wq.cpu().numpy()[clean], Wq_ref[clean]) < TOL_T  # reconstructed weights (rows without a flip)
    assert torch.equal(Wq, torch.gather(T, 1, Q.long()))  # every row, also those off the trajectory: the path's own T[Q]
    assert abs(float(Lo.double().sum()) - float(fx["losses_sum"])) <= 1e-4 * float(fx["losses_sum"])

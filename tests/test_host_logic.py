"""CPU tests of the host-side mirror of the reference interface: config validation, QuantLinear capability
checks, codebook recovery, work assignment, and the N>1 collective logic under gloo (world_size 2) with the
compute calls served by the CPU oracle (tests may use the oracle; the product path never does)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from ganq_amd import distributed as gdist  # noqa: E402
from ganq_amd.nn_modules.qlinear.ganq_hip import GanqHipQuantLinear  # noqa: E402
from ganq_amd.quantization.config import FORMAT, QuantizeConfig  # noqa: E402


def test_config_defaults_and_validation():
    q = QuantizeConfig(bits=4, quant_method="ganq")
    assert q.ganq_iterations == 5 and q.act_sort == "desc" and q.l_damp_style == "gptq" and q.dead == "zero"
    assert QuantizeConfig(desc_act=False).act_sort == "none"  # config.py:275-276
    for bad in (dict(bits=5), dict(damp_percent=1.5), dict(format="gptq"), dict(group_size=0), dict(pack_dtype="fp8"),
                dict(ganq_outlier_ratio=1.0), dict(ganq_outlier_ratio=-0.1)):
        with pytest.raises(ValueError):
            QuantizeConfig(**bad)
    q = QuantizeConfig(dynamic={r"-:.*\.k_proj": {}, r".*\.fc1": {"bits": 3}})
    assert q.dynamic_get("model.layers.0.k_proj") is False
    assert q.dynamic_get("model.layers.0.fc1", "bits", 4) == 3
    assert q.dynamic_get("model.layers.0.fc2", "bits", 4) == 4
    assert QuantizeConfig.from_dict(q.to_dict()).dynamic == q.dynamic
    assert q.ganq_outlier_ratio == 0.0  # the outlier split of the paper is opt-in
    assert QuantizeConfig.from_dict(QuantizeConfig(ganq_outlier_ratio=0.005).to_dict()).ganq_outlier_ratio == 0.005


def test_outlier_split_restatement():
    """oracle restatement of the paper's Algorithm 2 (Appendix A): cut-off positions, symmetric tails, W = sparse + dense"""
    from oracle import ganq_ref

    g = torch.Generator().manual_seed(0)
    W = torch.randn(40, 2000, generator=g)
    Ws, Wd, mask, c_lo, c_hi = ganq_ref.outlier_split(W, 0.01)
    assert torch.equal(Ws + Wd, W) and torch.equal(Wd[mask], torch.zeros(int(mask.sum())))
    srt = torch.sort(W, dim=1).values
    # floor(2000 * 0.995) = 1990; ceil(2000 * (1 - 0.995)) = 11 in double arithmetic (1 - 0.995 = 0.005000000000000004)
    assert torch.equal(c_hi, srt[:, 1990]) and torch.equal(c_lo, srt[:, 11])
    assert torch.equal(mask.sum(1), torch.full((40,), 10 + 12))  # >= sorted[1990]: 10 entries, <= sorted[11]: 12
    W[3, :1500] = 0.25  # ties with a cut-off all count
    _, _, mask, c_lo, _ = ganq_ref.outlier_split(W, 0.01)
    assert int(mask[3].sum()) >= 22


def test_quantlinear_validate_falls_through():
    ok, err = GanqHipQuantLinear.validate(bits=4, group_size=128, desc_act=True, sym=True, in_features=4096,
                                          out_features=4096, pack_dtype=torch.int32)
    assert ok and err is None
    for kw in (dict(bits=8), dict(in_features=100), dict(pack_dtype=torch.int16)):
        args = dict(bits=4, group_size=128, desc_act=True, sym=True, in_features=4096, out_features=4096,
                    pack_dtype=torch.int32)
        args.update(kw)
        ok, err = GanqHipQuantLinear.validate(**args)
        assert not ok and isinstance(err, NotImplementedError)  # caller tries the next backend (utils/model.py:234-239)


def test_codebook_from_weight_roundtrip():
    g = torch.Generator().manual_seed(0)
    T = torch.randn(12, 16, generator=g)
    T[3, 5] = T[3, 4]  # duplicate entry
    Q = torch.randint(0, 16, (12, 96), generator=g)
    W = T.gather(1, Q).half()
    Q2, T2 = GanqHipQuantLinear.codebook_from_weight(W, 4)
    assert torch.equal(T2.gather(1, Q2.long()).half(), W)
    with pytest.raises(ValueError):
        GanqHipQuantLinear.codebook_from_weight(torch.randn(2, 64), 4)


def test_assign_and_row_slices():
    shapes = {"q_proj": (4096, 4096), "k_proj": (1024, 4096), "v_proj": (1024, 4096), "o_proj": (4096, 4096)}
    for world in (1, 2, 3, 8):
        owners = gdist.assign(shapes, world)
        assert set(owners) == set(shapes) and all(0 <= r < world for r in owners.values())
        if world >= 2:
            assert owners["q_proj"] != owners["o_proj"]  # the two big ones never share a rank
    for m, world in ((4096, 8), (100, 3), (16, 4), (5, 2)):
        sl = gdist.row_slices(m, world)
        assert sl[0][0] == 0 and sl[-1][1] == m and all(a[1] == b[0] for a, b in zip(sl, sl[1:]))
        assert all((a % 16 == 0) for a, _ in sl if a < m)


sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle_quantizer import OracleSolver  # noqa: E402


def _golden(name):
    return np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))


def _worker(rank, world, port, name, alias, out_q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    import torch.distributed as td

    td.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import c_oracle

        c_oracle.set_num_threads(1)
        g = _golden(name)
        W, H, L = (torch.from_numpy(g[k]) for k in ("W_perm", "Xxt_damped", "L"))
        T0 = torch.from_numpy(g["T"][0])
        T, Q, dists, best_k = gdist.run_layer_row_sharded(W, H, L, T0, int(g["K"]), alias_q=alias,
                                                          dist=gdist.Dist(rank, world, torch.device("cpu")),
                                                          solver=OracleSolver())
        # activation broadcast + Hessian all-reduce helpers
        x = torch.arange(12.0).reshape(3, 4) if rank == 0 else None
        xb = gdist.broadcast_activations(x, (3, 4), torch.float32, 0, gdist.Dist(rank, world, torch.device("cpu")))
        Hp = torch.full((2, 2), float(rank + 1))
        gdist.allreduce_hessian(Hp, gdist.Dist(rank, world, torch.device("cpu")))
        out_q.put((rank, T.numpy(), Q.numpy(), dists.numpy(), best_k, xb.numpy(), Hp.numpy()))
    finally:
        td.destroy_process_group()


@pytest.mark.parametrize("name,alias", [("b32x64_b3", True), ("b32x64_b3", False), ("t24x48_b2", True)])
def test_row_sharded_two_ranks_gloo(name, alias):
    ctx = mp.get_context("spawn")
    out_q = ctx.Queue()
    port = 29600 + (os.getpid() % 200)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, name, alias, out_q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((out_q.get(timeout=120) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g = _golden(name)
    K = int(g["K"])
    best = int(np.argmin(g["dists"]))
    for rank, T, Q, dists, best_k, xb, Hp in res:
        assert best_k == best
        assert np.allclose(dists, g["dists"], rtol=1e-5)
        assert np.array_equal(Q, g["Q"][K - 1] if alias else g["Q"][best])
        assert np.linalg.norm(T - g["T"][best + 1]) / np.linalg.norm(g["T"][best + 1]) < 1e-5
        assert np.array_equal(xb, np.arange(12.0).reshape(3, 4)) and np.all(Hp == 3.0)
    assert np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][2], res[1][2])


def _worker4(rank, world, port, mode, out_q):
    """world-4 cases: ragged row slices with a rank that owns NO rows, sharded k-means (t0_fn), failure propagation"""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    import datetime

    import torch.distributed as td

    td.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    try:
        from oracle import c_oracle

        c_oracle.set_num_threads(1)
        dist = gdist.Dist(rank, world, torch.device("cpu"))
        g = _golden("c64x256_b4")   # 64 rows: 4 tiles of 16 -> with 40 rows kept, slices 16 + 16 + 8 + 0
        W, H, L = (torch.from_numpy(g[k]) for k in ("W_perm", "Xxt_damped", "L"))
        rows = 40
        W = W[:rows].contiguous()
        V, K = int(g["T"].shape[2]), int(g["K"])

        def t0_fn(W_rows):
            if mode == "fail" and rank == 1:
                raise RuntimeError("injected failure in the local part of rank 1")
            return torch.from_numpy(c_oracle.kmeans_init(W_rows.numpy(), None, V))

        if mode == "badbatch":
            x = torch.zeros(2, 3, dtype=torch.float64) if rank == 0 else None  # a dtype the exchange does not carry
            try:
                gdist.broadcast_calibration_batch(x, 0, dist)
                out_q.put((rank, "no error"))
            except ValueError as e:
                out_q.put((rank, "ValueError: " + str(e)))
            return
        try:
            T, Q, dists, best_k = gdist.run_layer_row_sharded(W, H, L, None, K, alias_q=True, dist=dist, solver=OracleSolver(),
                                                              t0_fn=t0_fn, V=V)
            out_q.put((rank, T.numpy(), Q.numpy(), dists.numpy(), best_k, gdist.row_slices(rows, world, align=16)))
        except RuntimeError as e:
            out_q.put((rank, "RuntimeError: " + str(e)))
    finally:
        td.destroy_process_group()


def _run4(mode):
    ctx = mp.get_context("spawn")
    out_q = ctx.Queue()
    port = 29850 + (os.getpid() % 100)
    procs = [ctx.Process(target=_worker4, args=(r, 4, port, mode, out_q)) for r in range(4)]
    for p in procs:
        p.start()
    res = sorted((out_q.get(timeout=180) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def test_row_sharded_four_ranks_zero_row_rank_gloo():
    """40 rows over 4 ranks in 16-row tiles: 16 + 16 + 8 + 0 -- the fourth rank owns nothing, clusters nothing (V is passed in)
    and still takes part in every collective; the initial codebook is clustered per slice and never exchanged; every rank ends
    with the result of the unsharded oracle run"""
    from oracle import c_oracle

    res = _run4("ok")
    g = _golden("c64x256_b4")
    W, H, L = g["W_perm"][:40].copy(), g["Xxt_damped"], g["L"]
    V, K = int(g["T"].shape[2]), int(g["K"])
    T0 = c_oracle.kmeans_init(W, None, V)
    To, Qo, do, bko = c_oracle.run_layer(W, H, L, T0, K)
    for rank, T, Q, dists, best_k, slices in res:
        assert slices == [(0, 16), (16, 32), (32, 40), (40, 40)]
        assert best_k == bko and np.array_equal(Q, Qo) and np.array_equal(T, To) and np.allclose(dists, do, rtol=1e-12)


def test_row_sharded_failure_on_one_rank_raises_on_all_gloo():
    """a rank whose local part fails (here: its k-means) must not leave the others waiting in the all-gather: the status
    exchange in front of the first collective makes EVERY rank raise"""
    res = _run4("fail")
    assert all(isinstance(r[1], str) and r[1].startswith("RuntimeError") for r in res), res
    assert "injected failure" in res[1][1] and all("rank 1 failed" in r[1] for i, r in enumerate(res) if i != 1)


def test_broadcast_calibration_batch_bad_dtype_raises_on_all_gloo():
    res = _run4("badbatch")
    assert all(r[1].startswith("ValueError") and "cannot send" in r[1] for r in res), res


def test_backend_registration_selects_the_lut_layer():
    """BACKEND / FORMAT_DICT / select_quant_linear as code (utils/importer.py:45-68,157-262 of the reference)"""
    from ganq_amd.nn_modules.backend import AUTO_SELECT_BACKEND_ORDER, BACKEND, FORMAT_DICT, select_quant_linear
    from ganq_amd.nn_modules.qlinear.ganq_hip import GanqHipQuantLinear
    from ganq_amd.quantization.config import FORMAT

    assert FORMAT_DICT[FORMAT.GANQ_LUT] == [BACKEND.GANQ_HIP] and FORMAT_DICT[FORMAT.FAKE] == []
    assert AUTO_SELECT_BACKEND_ORDER[BACKEND.GANQ_HIP] is GanqHipQuantLinear
    for bits in (2, 3, 4):
        assert select_quant_linear(bits, 128, True, True, format=FORMAT.GANQ_LUT) is GanqHipQuantLinear
        assert select_quant_linear(bits, -1, False, True, backend=BACKEND.GANQ_HIP, format=FORMAT.GANQ_LUT,
                                   multi_select=True) == [GanqHipQuantLinear]
    with pytest.raises(NotImplementedError):  # validate() refuses, so an AUTO caller would fall through (utils/model.py:234-239)
        select_quant_linear(8, 128, True, True, format=FORMAT.GANQ_LUT)
    with pytest.raises(ValueError):
        select_quant_linear(4, 128, True, True, backend=BACKEND.GANQ_HIP, format=FORMAT.FAKE)


# ------------------------------------------------------------------------------------------ the looper over two ranks
def _tiny_model(seed=0):
    import torch.nn as nn

    class Blk(nn.Module):
        def __init__(self):
            super().__init__()
            self.q_proj, self.k_proj, self.v_proj = nn.Linear(64, 64, bias=False), nn.Linear(64, 32, bias=False), nn.Linear(64, 32, bias=False)
            self.o_proj = nn.Linear(64, 64, bias=False)

        def forward(self, x):
            h = self.q_proj(x) + torch.cat([self.k_proj(x), self.v_proj(x)], -1)
            return x + self.o_proj(torch.tanh(h))

    torch.manual_seed(seed)
    return [Blk(), Blk()]


def _tiny_inputs():
    g = torch.Generator().manual_seed(5)
    scale = 0.2 + torch.rand(64, generator=g)
    return [torch.randn(2, 24, 64, generator=g) * scale for _ in range(5)]  # 5 batches: uneven shares over two ranks


def _run_looper(calibration, dist_mode):
    from ganq_amd.looper.module_looper import ModuleLooper
    from oracle_quantizer import OracleProcessor

    layers = _tiny_model()
    proc = OracleProcessor(QuantizeConfig(bits=3, act_sort="asc", l_damp_style="ganq", dead="mean", ganq_iterations=2))
    looper = ModuleLooper(proc, layers, [["q_proj", "k_proj", "v_proj"], ["o_proj"]], layers_prefix="layers",
                          dist_mode=dist_mode, calibration=calibration)
    outs = looper.loop(_tiny_inputs())
    res = {k: (v["ganq_q"].numpy().copy(), v["ganq_lut"].numpy().copy()) for k, v in proc.results().items()}
    return res, [o.numpy().copy() for o in outs], [dict(r) for r in proc.log]


def _looper_worker(rank, world, port, calibration, out_q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    import torch.distributed as td

    td.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import c_oracle

        c_oracle.set_num_threads(1)
        out_q.put((rank,) + _run_looper(calibration, "rows"))
    finally:
        td.destroy_process_group()


@pytest.mark.parametrize("calibration", ["broadcast", "allreduce"])
def test_looper_rows_mode_two_ranks_gloo(calibration):
    """ModuleLooper(dist_mode="rows") over two gloo ranks with the CPU oracle in the quantizer slot: data-parallel calibration
    (ranks forward batches 0,2,4 / 1,3), Hessian exchange, row-sharded k-means + loop, un-hit bookkeeping.  With
    calibration="broadcast" every bit equals the single-process run; with "allreduce" the ranks agree with each other and
    with the single-process run up to the rounding of the summed Hessian."""
    from oracle import c_oracle

    torch.set_num_threads(1)
    c_oracle.set_num_threads(1)
    base, base_outs, base_log = _run_looper("allreduce", "none")
    ctx = mp.get_context("spawn")
    out_q = ctx.Queue()
    port = 29300 + (os.getpid() % 200) + (7 if calibration == "broadcast" else 0)
    procs = [ctx.Process(target=_looper_worker, args=(r, 2, port, calibration, out_q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((out_q.get(timeout=180) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, r0, o0, log0), (_, r1, o1, log1) = res
    assert sorted(r0) == sorted(r1) == sorted(base) and len(base) == 8
    assert [(r["layer"], r["module"]) for r in log0] == [(r["layer"], r["module"]) for r in base_log]
    for name in base:
        assert np.array_equal(r0[name][0], r1[name][0]) and np.array_equal(r0[name][1], r1[name][1]), name  # ranks agree
        if calibration == "broadcast":
            assert np.array_equal(r0[name][0], base[name][0]) and np.array_equal(r0[name][1], base[name][1]), name
        else:
            assert (r0[name][0] != base[name][0]).mean() < 0.02, name
            assert np.linalg.norm(r0[name][1] - base[name][1]) / np.linalg.norm(base[name][1]) < 1e-3, name
    assert len(o0) == 3 and len(o1) == 2  # every rank keeps the outputs of its own batches
    if calibration == "broadcast":
        for got, want in zip(o0 + o1, [base_outs[i] for i in (0, 2, 4, 1, 3)]):
            assert np.array_equal(got, want)

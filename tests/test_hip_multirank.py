"""GPU rehearsal of the N>1 paths with two ranks sharing the one GPU of the test box (gloo, GANQ_DIST_SHARE_DEVICE=1;
RCCL refuses two ranks on one device): row-sharded loop and module dispatch in the looper, both against the
single-rank HIP result."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rows_mode_through_looper(rank):
    """two decoder-like layers, groups [q,k,v] / [o]; returns per calibration mode the comparison with the SAME process
    quantizing alone (dist_mode="none") and the packed tensors for the cross-rank comparison"""
    import copy

    import torch.nn as nn

    from ganq_amd.looper.gptq_processor import GPTQProcessor
    from ganq_amd.looper.module_looper import ModuleLooper
    from ganq_amd.quantization import QuantizeConfig

    class Blk(nn.Module):
        def __init__(self):
            super().__init__()
            self.q_proj, self.k_proj, self.v_proj = nn.Linear(256, 256, bias=False), nn.Linear(256, 128, bias=False), nn.Linear(256, 128, bias=False)
            self.o_proj = nn.Linear(256, 256, bias=False)

        def forward(self, x):
            h = self.q_proj(x) + torch.cat([self.k_proj(x), self.v_proj(x)], -1)
            return x + self.o_proj(torch.tanh(h))

    torch.manual_seed(3)
    layers0 = nn.ModuleList([Blk(), Blk()]).half().cuda()
    g = torch.Generator(device="cuda").manual_seed(9)
    scale = 0.2 + torch.rand(256, device="cuda", generator=g)
    xs = [(torch.randn(2, 96, 256, device="cuda", generator=g) * scale).half() for _ in range(5)]
    groups = [["q_proj", "k_proj", "v_proj"], ["o_proj"]]

    def run(dist_mode, calibration, share):
        layers = copy.deepcopy(layers0)
        model = nn.Module()
        model.layers = layers
        proc = GPTQProcessor(QuantizeConfig(bits=4, act_sort="asc", l_damp_style="ganq", dead="mean", ganq_iterations=3,
                                            ganq_outlier_ratio=0.02))
        with torch.no_grad():
            outs = ModuleLooper(proc, list(layers), groups, layers_prefix="layers", dist_mode=dist_mode, calibration=calibration,
                                share_group_hessian=share).loop(xs)
        proc.finalize(model)
        st = {k: v.detach().cpu().numpy() for k, v in layers.state_dict().items()}
        return st, [o.float().cpu().numpy() for o in outs]

    report = {}
    alone, alone_outs = run("none", "allreduce", True)
    for calibration, share in (("broadcast", True), ("broadcast", False), ("allreduce", True)):
        st, outs = run("rows", calibration, share)
        same = {k: bool(np.array_equal(st[k], alone[k])) for k in alone}
        mine = list(range(rank, 5, 2))
        outs_same = all(np.array_equal(o, alone_outs[b]) for o, b in zip(outs, mine))
        qdiff = max(float((st[k] != alone[k]).mean()) for k in alone if k.endswith("qweight"))
        report[f"{calibration}/{share}"] = dict(identical_to_alone=all(same.values()), outs_identical=outs_same,
                                                 n_tensors=len(same), packed_word_mismatch=qdiff, state=st)
    return report


def _worker(rank, world, port, out_q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(world), GANQ_DIST_SHARE_DEVICE="1")
    import torch.distributed as td
    import torch.nn as nn

    from ganq_amd import _lib
    from ganq_amd import distributed as gdist
    from ganq_amd.looper.gptq_processor import GPTQProcessor
    from ganq_amd.looper.module_looper import ModuleLooper
    from ganq_amd.quantization import QuantizeConfig
    from test_hip_stages import synth

    dist = gdist.init_from_env()
    try:
        # row-sharded loop = the fused driver on each rank's slice + one exchange of per-row losses: every returned bit
        # (codebooks, indices, the K distances, best_k) equals the single-rank ganq_run_layer result
        ok_rows = True
        for (m, n, V, K, alias) in [(80, 256, 16, 3, True), (300, 512, 16, 4, True), (300, 512, 8, 4, False)]:
            W, H, L, T0 = (torch.from_numpy(a).cuda() for a in synth(m, n, V, 77 + m, corr=0.2))
            T, Q, dists, best_k = gdist.run_layer_row_sharded(W, H, L, T0, K, alias_q=alias, dist=dist)
            T1, Q1, d1, b1 = _lib.run_layer(W, H, L, T0, K, alias_q=alias)
            ok_rows = ok_rows and bool(torch.equal(Q, Q1)) and best_k == int(b1) and bool(torch.equal(dists, d1))
            ok_rows = ok_rows and bool(torch.equal(T, T1))

        torch.manual_seed(0)  # same model on both ranks
        layer = nn.ModuleDict({"q_proj": nn.Linear(64, 64, bias=False), "k_proj": nn.Linear(64, 32, bias=False),
                               "v_proj": nn.Linear(64, 32, bias=False)}).half().cuda()

        class Blk(nn.Module):
            def __init__(self, d):
                super().__init__()
                self.q_proj, self.k_proj, self.v_proj = d["q_proj"], d["k_proj"], d["v_proj"]

            def forward(self, x):
                return x + self.q_proj(x) + torch.cat([self.k_proj(x), self.v_proj(x)], -1)

        blk = Blk(layer)
        model = nn.Module()
        model.layers = nn.ModuleList([blk])
        g = torch.Generator(device="cuda").manual_seed(5)
        xs = [torch.randn(2, 40, 64, device="cuda", generator=g).half() for _ in range(3)]
        # with the outlier split on: the non-owner ranks must receive the owner's exact outliers as well
        proc = GPTQProcessor(QuantizeConfig(bits=4, act_sort="asc", l_damp_style="ganq", dead="mean", ganq_iterations=2,
                                            ganq_outlier_ratio=0.05))
        with torch.no_grad():
            ModuleLooper(proc, [blk], [["q_proj", "k_proj", "v_proj"]], layers_prefix="layers", dist_mode="modules").loop(xs)
        owners = gdist.assign({"q_proj": (64, 64), "k_proj": (32, 64), "v_proj": (32, 64)}, world)
        mine = sorted(n for n, r in owners.items() if r == rank)
        proc.finalize(model)  # nn.Linear -> GanqHipQuantLinear built from the (received) results
        state = {k: v.float().cpu().numpy() for k, v in blk.state_dict().items()}
        for name in ("q_proj", "k_proj", "v_proj"):
            state["dequant." + name] = getattr(blk, name).dequantize_weight().float().cpu().numpy()
            state["nnz." + name] = np.array([int(proc.results()["layers.0." + name]["ganq_outliers"][1].numel())])
        # ---- dist_mode="rows" through the looper: data-parallel calibration + row-sharded k-means / loop for every module
        rows_report = _rows_mode_through_looper(rank)
        out_q.put((rank, ok_rows, mine, sorted(proc.results()), state, rows_report))
    except Exception as e:  # report instead of letting the parent wait for its queue timeout
        out_q.put((rank, False, [f"ERROR {type(e).__name__}: {e}"], [], {}, {}))
        raise
    finally:
        td.destroy_process_group()


def test_two_ranks_share_one_gpu():
    ctx = mp.get_context("spawn")
    out_q = ctx.Queue()
    port = 29800 + (os.getpid() % 100)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out_q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((out_q.get(timeout=400) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, ok0, mine0, names0, st0, rows0), (r1, ok1, mine1, names1, st1, rows1) = res
    assert ok0 and ok1, f"row-sharded loop differs from the single-rank result / worker error: {mine0} {mine1}"
    assert mine0 and mine1 and not set(mine0) & set(mine1)          # the group's modules were split over the ranks
    assert names0 == names1 and len(names0) == 3                      # both ranks hold every result afterwards
    for k in st0:
        assert np.array_equal(st0[k], st1[k]), k                      # and identical quantized weights / packed layers
    assert all(int(st0["nnz." + n][0]) > 0 for n in ("q_proj", "k_proj", "v_proj"))  # the outliers travelled
    # dist_mode="rows" through the looper (data-parallel calibration, row-sharded k-means + loop for every module)
    assert set(rows0) == {"broadcast/True", "broadcast/False", "allreduce/True"}
    for key in rows0:
        a, b = rows0[key], rows1[key]
        assert a["n_tensors"] >= 2 * 4 * 2  # 2 layers x 4 modules x (qweight, lut, ...)
        for k in a["state"]:
            assert np.array_equal(a["state"][k], b["state"][k]), f"{key}: ranks disagree on {k}"
        if key.startswith("broadcast"):
            # activations broadcast, every rank accumulates every batch in single-GPU order: every packed bit equals the
            # run of one process on its own
            assert a["identical_to_alone"] and b["identical_to_alone"], key
            assert a["outs_identical"] and b["outs_identical"], key
        else:
            # partial Hessians summed by all-reduce: same statistics up to fp32 rounding of the sum
            assert a["packed_word_mismatch"] < 0.05, (key, a["packed_word_mismatch"])

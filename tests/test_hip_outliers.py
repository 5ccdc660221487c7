"""GPU tests of the outlier split in front of GANQ (paper section 3.3 / Appendix A): the HIP kernels
behind the C-ABI against the oracle's restatement of Algorithm 2, the sparse product and the fused LUT forward against
torch, and the quantizer / QuantLinear plugin path end to end."""
import numpy as np
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from ganq_amd import _lib

    assert torch.cuda.is_available(), "these tests need the MI355X"
    _lib.lib()
    return _lib


def heavy_tailed(m, n, seed, half=True):
    g = torch.Generator().manual_seed(seed)
    W = 0.02 * torch.randn(m, n, generator=g)
    W += 0.3 * torch.randn(m, n, generator=g) * (torch.rand(m, n, generator=g) < 0.004)  # a few large entries
    return W.half().float() if half else W


@pytest.mark.parametrize("m,n,ratio", [(8, 64, 0.05), (33, 1000, 0.005), (128, 4096, 0.005), (16, 11008, 0.0045),
                                       (5, 16384, 0.01), (64, 768, 0.5)])
def test_outlier_split_matches_algorithm_2(hip, m, n, ratio):
    from oracle import ganq_ref

    W = heavy_tailed(m, n, seed=m + n)
    W[0, : n // 2] = W[0, 0]  # a row with massive ties (the ties with a cut-off are all outliers)
    Ws, Wd, mask, c_lo, c_hi = ganq_ref.outlier_split(W, ratio)
    Wg = W.cuda().clone()
    rowptr, cols, vals, cut = hip.outlier_split(Wg, ratio)
    assert torch.equal(cut[:, 0].cpu(), c_lo) and torch.equal(cut[:, 1].cpu(), c_hi)
    assert torch.equal(Wg.cpu(), Wd)  # W_dense, bit for bit
    counts = mask.sum(1)
    assert torch.equal((rowptr[1:] - rowptr[:-1]).cpu().long(), counts)
    rows_ref, cols_ref = torch.nonzero(mask, as_tuple=True)  # row-major: ascending columns inside a row
    assert torch.equal(cols.cpu().long(), cols_ref)
    assert torch.equal(vals.cpu(), W[rows_ref, cols_ref])


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M,m,n,ratio", [(1, 256, 1024, 0.01), (7, 100, 512, 0.1), (64, 4096, 4096, 0.005), (3, 48, 2048, 0.0),
                                         # decode kernel (M <= 32, out_features >= 128): the sparse entries are added inside the
                                         # LUT kernel's launch; 16 / 32 features per workgroup, one / two row tiles
                                         (1, 4096, 1024, 0.01), (16, 2048, 512, 0.05), (20, 1024, 768, 0.02), (3, 8192, 256, 0.03),
                                         (1, 1500, 512, 0.0),
                                         # ... and with in_features split across workgroups (the sparse part goes with split 0)
                                         (1, 2048, 8192, 0.01), (16, 1024, 4096, 0.02)])
def test_outlier_matmul_and_fused_forward(hip, dtype, M, m, n, ratio):
    g = torch.Generator().manual_seed(M * 13 + m)
    W = heavy_tailed(m, n, seed=n + M)
    if dtype == torch.bfloat16:
        W = W.bfloat16().float()
    if ratio > 0:
        rowptr, cols, vals, _ = hip.outlier_split(W.cuda().clone(), ratio)
    else:  # a layer whose rows hold no outliers at all
        rowptr = torch.zeros(m + 1, dtype=torch.int32, device="cuda")
        cols = torch.zeros(0, dtype=torch.int32, device="cuda")
        vals = torch.zeros(0, dtype=torch.float32, device="cuda")
    x = torch.randn(M, n, generator=g).to(dtype).cuda()
    out = hip.outlier_matmul(x, rowptr, cols, vals.to(dtype), m)
    Ws = torch.zeros(m, n, dtype=torch.float64, device="cuda")
    rows = torch.repeat_interleave(torch.arange(m, device="cuda"), (rowptr[1:] - rowptr[:-1]).long())
    Ws[rows, cols.long()] = vals.double()
    ref = x.double() @ Ws.T
    assert (out.double() - ref).abs().max() <= 1e-5 * max(1.0, float(ref.abs().max()))

    # fused: LUT forward with the sparse product as fp32 addend == F.linear on (dequantised + sparse) weights
    bits, V = 4, 16
    Q = torch.randint(0, V, (m, n), generator=g, dtype=torch.uint8).cuda()
    lut = (0.02 * torch.randn(m, V, generator=g)).to(dtype).cuda()
    bias = (0.1 * torch.randn(m, generator=g)).to(dtype).cuda()
    qweight = hip.pack_indices(Q, bits)
    y = hip.lut_linear(x, qweight, lut, bias, bits, addend=out)
    Wq = lut.double().gather(1, Q.long()) + Ws
    y_ref = x.double() @ Wq.T + bias.double()
    tol = (2.0 ** -10 if dtype == torch.float16 else 2.0 ** -7) * float(y_ref.abs().max()) + 1e-3
    assert (y.double() - y_ref).abs().max() <= tol
    # the one-call entry (decode sizes: one launch; otherwise sparse product + LUT kernel) against the same reference
    y2 = hip.lut_linear_outliers(x, qweight, lut, bias, bits, rowptr, cols, vals.to(dtype))
    assert (y2.double() - y_ref).abs().max() <= tol


@pytest.mark.parametrize("act_sort,desc_act", [("asc", True), ("none", False)])
@torch.no_grad()
def test_ganq_with_outlier_split_end_to_end(hip, act_sort, desc_act, tmp_path):
    """quantizer plugin with ganq_outlier_ratio: effective weight = T[Q] + W_sparse, lower error than plain GANQ on a
    heavy-tailed weight, packed layer == fake-quant layer, save / load round trip"""
    from ganq_amd.looper.gptq_processor import GPTQProcessor
    from ganq_amd.looper.module_looper import ModuleLooper
    from ganq_amd.models.quantize import load_quantized, save_quantized
    from ganq_amd.nn_modules.qlinear.ganq_hip import GanqHipQuantLinear
    from ganq_amd.quantization import QuantizeConfig

    class Layer(nn.Module):
        def __init__(self, W):
            super().__init__()
            self.proj = nn.Linear(W.shape[1], W.shape[0], bias=True)
            self.proj.weight.data = W.clone()

        def forward(self, x):
            return self.proj(x)

    def build():
        torch.manual_seed(5)
        model = nn.Module()
        model.layers = nn.ModuleList([Layer(heavy_tailed(96, 256, seed=9))])
        return model.half().cuda()

    xs = [(torch.randn(2, 64, 256, generator=torch.Generator().manual_seed(50 + i)) *
           (0.2 + torch.rand(256, generator=torch.Generator().manual_seed(3)))).half().cuda() for i in range(4)]
    W0 = build().layers[0].proj.weight.data.float()
    H = sum((x.reshape(-1, 256).float().T @ x.reshape(-1, 256).float()) for x in xs)
    errs = {}
    for ratio in (0.0, 0.02):
        model = build()
        qcfg = QuantizeConfig(bits=3, act_sort=act_sort, desc_act=desc_act, l_damp_style="ganq", dead="mean",
                              ganq_iterations=3, ganq_outlier_ratio=ratio)
        proc = GPTQProcessor(qcfg)
        ModuleLooper(proc, model.layers, [["proj"]], layers_prefix="layers").loop(xs)
        res = proc.results()["layers.0.proj"]
        wq = model.layers[0].proj.weight.data.float()
        E = wq - W0
        errs[ratio] = float(((E @ H) * E).sum())
        if ratio == 0.0:
            assert res["ganq_outliers"] is None
            continue
        rowptr, cols, vals = res["ganq_outliers"]
        assert int(rowptr[-1]) == cols.numel() > 0
        # the effective weight is codebook[indices] + outliers, in the column order of the returned weight
        eff = res["ganq_lut"].gather(1, res["ganq_q"].long())
        rows = torch.repeat_interleave(torch.arange(96, device="cuda"), (rowptr[1:] - rowptr[:-1]).long())
        eff[rows, cols.long()] += vals
        assert torch.equal(eff.half(), model.layers[0].proj.weight.data)
        x = xs[0][:1, :5]
        y_fake = model.layers[0](x)
        proc.finalize(model)
        q = model.layers[0].proj
        assert isinstance(q, GanqHipQuantLinear) and q.outliers == cols.numel()
        y_lut = model.layers[0](x)
        assert (y_lut.float() - y_fake.float()).abs().max() <= 2e-2 * float(y_fake.float().abs().max())
        assert (q.dequantize_weight().float() - eff).abs().max() <= 2e-3 * float(eff.abs().max())
        y_big = model.layers[0](torch.cat([xs[0], xs[1]], 0))  # 256 rows: the prefill path adds the dense W_sparse
        assert y_big.shape == (4, 64, 96)

        class Wrap(nn.Module):  # save / load needs a model object with a state dict
            def __init__(self, layers):
                super().__init__()
                self.layers = layers

        saved = Wrap(model.layers)
        save_quantized(saved, str(tmp_path), qcfg)
        fresh = Wrap(build().layers)
        load_quantized(fresh, str(tmp_path))
        q2 = fresh.layers[0].proj
        assert isinstance(q2, GanqHipQuantLinear) and q2.outliers == q.outliers
        assert torch.equal(q2.outlier_cols, q.outlier_cols) and torch.equal(q2.outlier_vals, q.outlier_vals)
        assert torch.equal(fresh.layers[0](x), y_lut)
    assert errs[0.02] < errs[0.0]

"""The fused LUT-dequant GEMM (csrc/lut_gemm.hip; M > 64: prefill, perplexity evaluation) through the C-ABI, against the
oracle of the QuantLinear forward: F.linear(x, T.gather(1, Q).to(dtype), bias) == FakeQuantLinear.forward (fake.py:88-89),
evaluated in fp64 on the dequantised weight.  Tolerance: fp32 accumulation + ONE rounding to the activation dtype."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from ganq_amd import _lib

    assert torch.cuda.is_available(), "these tests need the MI355X"
    _lib.selftest()
    return _lib


def make(bits, M, m, n, dtype, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    V = 2 ** bits
    Q = torch.randint(0, V, (m, n), device="cuda", generator=g, dtype=torch.int32).to(torch.uint8)
    lut = (0.02 * torch.randn(m, V, device="cuda", generator=g)).to(dtype)
    x = torch.randn(M, n, device="cuda", generator=g).to(dtype)
    bias = (0.1 * torch.randn(m, device="cuda", generator=g)).to(dtype)
    return Q, lut, x, bias


def check(y, x, Q, lut, bias, dtype, addend=None):
    Wq = torch.gather(lut.double(), 1, Q.long())
    ref = x.double() @ Wq.T
    if addend is not None:
        ref = ref + addend.double()
    if bias is not None:
        ref = ref + bias.double()
    eps = 2 ** -10 if dtype == torch.float16 else 2 ** -7
    assert torch.isfinite(y).all()
    assert torch.allclose(y.double(), ref, rtol=eps, atol=eps * float(ref.abs().max()) * 0.05 + 1e-6), float((y.double() - ref).abs().max())


@pytest.mark.parametrize("bits,M,dtype,m,n", [
    (4, 65, torch.float16, 256, 512), (4, 128, torch.float16, 128, 256), (4, 200, torch.bfloat16, 300, 1024),
    (3, 129, torch.float16, 1000, 2048), (2, 512, torch.float16, 512, 768), (3, 333, torch.bfloat16, 1025, 320),
    (4, 512, torch.float16, 4096, 4096), (4, 2048, torch.bfloat16, 2048, 2048), (4, 1000, torch.float16, 129, 64),
    (4, 257, torch.float16, 8200, 512), (2, 96, torch.bfloat16, 64, 4096)])
def test_lut_gemm_vs_dense(hip, bits, M, dtype, m, n):
    Q, lut, x, bias = make(bits, M, m, n, dtype, bits * 1000 + M)
    qw = hip.pack_indices(Q, bits)
    check(hip.lut_linear(x, qw, lut, bias, bits), x, Q, lut, bias, dtype)
    check(hip.lut_linear(x, qw, lut, None, bits), x, Q, lut, None, dtype)


@pytest.mark.parametrize("M", [65, 512, 2048, 4096])
@pytest.mark.parametrize("m,n", [(4096, 4096), (14336, 4096)])
def test_lut_gemm_baseline_shapes_vs_f_linear(hip, M, m, n):
    """the shapes VERDICT r02 item 6 names; besides the fp64 bound, the distance to the path it replaces
    (dequantised weight + library fp16 GEMM) is of the order of one rounding"""
    Q, lut, x, bias = make(4, M, m, n, torch.float16, M + m)
    qw = hip.pack_indices(Q, 4)
    y = hip.lut_linear(x, qw, lut, bias, 4)
    check(y, x, Q, lut, bias, torch.float16)
    y_lib = torch.nn.functional.linear(x, hip.lut_dequant(qw, lut, n, 4), bias)
    assert float((y.float() - y_lib.float()).abs().max()) <= 2 ** -8 * float(y_lib.float().abs().max())
    assert torch.equal(y, hip.lut_linear(x, qw, lut, bias, 4))  # no split-K, no atomics: bitwise repeatable


def test_lut_gemm_with_addend_and_outliers(hip):
    Q, lut, x, bias = make(4, 300, 512, 1024, torch.float16, 5)
    qw = hip.pack_indices(Q, 4)
    add = torch.randn(300, 512, device="cuda")
    check(hip.lut_linear(x, qw, lut, bias, 4, addend=add), x, Q, lut, bias, torch.float16, addend=add)
    # CSR outliers: one entry in every 5th output feature
    rows = torch.arange(0, 512, 5, device="cuda")
    cnt = torch.zeros(512, dtype=torch.int32, device="cuda")
    cnt[rows] = 1
    rowptr = torch.zeros(513, dtype=torch.int32, device="cuda")
    rowptr[1:] = torch.cumsum(cnt, 0)
    cols = ((rows * 7) % 1024).to(torch.int32)
    vals = torch.full((rows.numel(),), 0.25, dtype=torch.float16, device="cuda")
    sparse = torch.zeros(300, 512, device="cuda", dtype=torch.float64)
    sparse[:, rows] = 0.25 * x.double()[:, cols.long()]
    y = hip.lut_linear_outliers(x, qw, lut, bias, 4, rowptr, cols, vals)
    check(y, x, Q, lut, bias, torch.float16, addend=sparse)


def test_quantlinear_forward_prefill_uses_the_fused_kernel(hip, monkeypatch):
    """GanqHipQuantLinear.forward at M > 64 never reaches a library GEMM"""
    from ganq_amd.nn_modules.qlinear.ganq_hip import GanqHipQuantLinear

    def boom(*a, **k):
        raise AssertionError("F.linear called from GanqHipQuantLinear.forward")

    Q, lut, x, bias = make(4, 4 * 100, 256, 512, torch.float16, 9)
    layer = GanqHipQuantLinear(bits=4, group_size=128, sym=True, desc_act=True, in_features=512, out_features=256, bias=True,
                               pack_dtype=torch.int32).cuda()
    layer.qweight = hip.pack_indices(Q, 4)
    layer.lut = lut
    layer.bias = bias
    monkeypatch.setattr(torch.nn.functional, "linear", boom)
    y = layer(x.reshape(4, 100, 512))
    assert y.shape == (4, 100, 256)
    check(y.reshape(400, 256), x, Q, lut, bias, torch.float16)


@pytest.mark.parametrize("ks", [2, 3, 8])
def test_lut_gemm_split_k_is_deterministic_and_matches(hip, ks):
    """64 < M <= ~1024 splits in_features over workgroups (fp32 partial tiles + ticket, summed in split order by the last
    workgroup); forced split factors incl. a ragged one against the fp64 bound, and bitwise stable over repeats / after
    other shapes used the same workspace"""
    Q, lut, x, bias = make(4, 300, 640, 4096, torch.float16, 11)
    qw = hip.pack_indices(Q, 4)
    Q2, lut2, x2, _ = make(3, 100, 256, 2048, torch.float16, 12)
    qw2 = hip.pack_indices(Q2, 3)
    hip.debug_option("GANQ_LUT_GEMM_RM", ks)
    try:
        y = hip.lut_linear(x, qw, lut, bias, 4)
        check(y, x, Q, lut, bias, torch.float16)
        for _ in range(5):
            check(hip.lut_linear(x2, qw2, lut2, None, 3), x2, Q2, lut2, None, torch.float16)
            assert torch.equal(hip.lut_linear(x, qw, lut, bias, 4), y)
    finally:
        hip.debug_option("GANQ_LUT_GEMM_RM", None)
    # the shape-driven plan splits this one too (3 x 5 tiles)
    assert torch.allclose(hip.lut_linear(x, qw, lut, bias, 4).float(), y.float(), rtol=2 ** -9, atol=1e-3)


@pytest.mark.parametrize("bits,M,dtype,m,n", [(4, 300, torch.float16, 384, 1024), (3, 257, torch.bfloat16, 200, 2048),
                                              (2, 512, torch.float16, 128, 96), (4, 1000, torch.float16, 1025, 160)])
def test_lut_gemm_pipelined_kernel_forced(hip, bits, M, dtype, m, n):
    """the one-workgroup-per-CU pipelined kernel (taken by shape for >= 224 whole-K tiles) forced on small problems: odd stage
    counts, ragged last stage, ragged rows / features; equal to the other kernel to rounding, to the fp64 bound"""
    Q, lut, x, bias = make(bits, M, m, n, dtype, 77 + M)
    qw = hip.pack_indices(Q, bits)
    try:
        hip.debug_option("GANQ_LUT_GEMM_RM", 1)  # no split-K: the switch below only applies to whole-K launches
        hip.debug_option("GANQ_LUT_GEMM_PIPE", 1)
        y1 = hip.lut_linear(x, qw, lut, bias, bits)
        hip.debug_option("GANQ_LUT_GEMM_PIPE", 0)
        y0 = hip.lut_linear(x, qw, lut, bias, bits)
    finally:
        hip.debug_option("GANQ_LUT_GEMM_PIPE", None)
        hip.debug_option("GANQ_LUT_GEMM_RM", None)
    check(y1, x, Q, lut, bias, dtype)
    assert torch.equal(y0, y1)  # same products, same accumulation order per output: identical bits

"""csrc/gemm_h16.hip -- the dense fp16 / bf16 GEMM behind the LUT forward's prefill path (dequantise once + GEMM, M >= ~1024) --
against an fp64 product of the same 16-bit inputs.  No reference counterpart beyond fake.py:88-89 (F.linear); the LUT-forward
tests (test_hip_lut_gemm.py) cover the path end to end."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def ref(x, w, bias, addend):
    y = x.double() @ w.double().T
    if bias is not None:
        y = y + bias.double()
    if addend is not None:
        y = y + addend.double()
    return y


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M,N,K,bm", [(256, 256, 64, 256), (256, 256, 128, 128), (512, 768, 256, 256), (300, 512, 192, 128),
                                      (1000, 260, 320, 256), (77, 1028, 64, 128), (1024, 1024, 1024, 256), (640, 3072, 768, 0)])
def test_gemm_h16_vs_fp64(dtype, M, N, K, bm, lib_options):
    from ganq_amd import _lib

    if bm:
        lib_options(GANQ_GEMM_H16_BM=bm)
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    x = torch.randn(M, K, device="cuda", generator=g).to(dtype)
    w = (0.05 * torch.randn(N, K, device="cuda", generator=g)).to(dtype)
    bias = torch.randn(N, device="cuda", generator=g).to(dtype)
    addend = torch.randn(M, N, device="cuda", generator=g)
    for b, a in ((None, None), (bias, None), (bias, addend)):
        y = _lib.debug_gemm_h16(x, w, b, a)
        want = ref(x, w, b, a)
        tol = (2.0 ** -10 if dtype == torch.float16 else 2.0 ** -7)  # one rounding of the result + fp32 accumulation
        err = (y.double() - want).abs()
        bound = tol * want.abs() + 2e-5 * (K ** 0.5)  # + fp32 accumulation of K products of size ~0.05
        assert bool((err <= bound).all()), f"max err {float(err.max()):.3e} at {int(err.argmax())}, bound there {float(bound.flatten()[err.argmax()]):.3e}"


def test_gemm_h16_identity_and_asymmetric_operand():
    """x = I (exactly representable): y must be w^T exactly -- catches a swapped row / feature map or a wrong k order"""
    from ganq_amd import _lib

    K = N = 512
    w = (torch.arange(N * K, device="cuda", dtype=torch.float32).reshape(N, K) % 2039 - 1000).to(torch.float16)  # asymmetric, exact
    x = torch.eye(K, device="cuda", dtype=torch.float16)
    y = _lib.debug_gemm_h16(x, w)
    assert torch.equal(y, w.T.contiguous())

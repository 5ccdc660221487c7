"""nn.Conv2d through the quantizer: the reference flattens the weight to [out_channels, in_channels * kh * kw] (gptq.py:80-81) and
accumulates the Hessian over the unfolded patches (gptq.py:111-121).  Golden: tests/golden/conv/ (make_golden_conv.py, the
reference's own GANQ object on a Conv2d).  CPU: the oracle against it; `-m gpu`: the HIP quantizer object."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR, rel_fro

CASE = os.path.join(GOLDEN_DIR, "conv", "conv16x8x3x3_b4.npz")


def _unfolded(g):
    """[tokens, C kh kw] per calibration batch, in the reference's order (torch F.unfold: tensor plumbing)"""
    out = []
    for xb in g["X"]:
        u = torch.nn.functional.unfold(torch.from_numpy(xb).float(), int(g["k"]), padding=int(g["pad"]), stride=int(g["stride"]))
        out.append(u.permute(0, 2, 1).reshape(-1, u.shape[1]))
    return out


def test_oracle_conv2d_vs_reference():
    from oracle import c_oracle

    g = np.load(CASE)
    K, V = int(g["K"]), 2 ** int(g["bits"])
    # Hessian over the patches: H = (2 / N) sum X^T X with N = images (gptq.py:104 counts shape[0])
    toks = _unfolded(g)
    nimg = sum(x.shape[0] for x in g["X"])
    H = sum((t.double().T @ t.double()) for t in toks) * (2.0 / nimg)
    assert rel_fro(H.numpy(), g["H_raw"]) < 1e-5 and int(g["nsamples"]) == nimg
    W, L, Hd = g["W_perm"], g["L"], g["Xxt_damped"]
    assert np.array_equal(W, g["W"].reshape(W.shape[0], -1)[:, g["perm"]])  # flatten(1), then the act_sort permutation
    WH = c_oracle.matmul(W, Hd)
    for k in range(K):
        assert np.array_equal(c_oracle.solve_s(W, L, g["T"][k]), g["Q"][k])
        assert rel_fro(c_oracle.update_t(WH, Hd, g["Q"][k], V), g["T"][k + 1]) < 1e-5


@pytest.mark.gpu
def test_hip_conv2d_quantize_vs_reference():
    from ganq_amd.looper.named_module import NamedModule
    from ganq_amd.quantization import GANQ, QuantizeConfig

    g = np.load(CASE)
    conv = torch.nn.Conv2d(int(g["in_ch"]), int(g["out_ch"]), int(g["k"]), padding=int(g["pad"]), stride=int(g["stride"]), bias=True).half().cuda()
    with torch.no_grad():
        conv.weight.copy_(torch.from_numpy(g["W"]))
        conv.bias.copy_(torch.from_numpy(g["bias"]))
    qcfg = QuantizeConfig(bits=int(g["bits"]), quant_method="ganq", format="fake", act_sort="asc", l_damp_style="ganq", dead="mean",
                          desc_act=True, ganq_iterations=int(g["K"]), group_size=128, damp_percent=0.01)
    q = GANQ(NamedModule(conv, "conv", "model.layers.0.conv", 0), qcfg)
    q.quantizer.configure(perchannel=True)
    assert (q.rows, q.columns) == g["W_perm"].shape
    for xb in g["X"]:
        q.add_batch(torch.from_numpy(xb).cuda(), None)
    assert q.nsamples == int(g["nsamples"])
    assert rel_fro(q.hessian.cpu().numpy(), g["H_raw"]) < 1e-5
    wq, scale, zero, g_idx, duration, avg_loss, damp = q.quantize()
    assert wq.shape == conv.weight.shape and wq.dtype == torch.float16
    K = int(g["K"])
    Qref = g["Q"][K - 1][:, np.argsort(g["perm"])]  # aliased indices of the last iteration, original column order
    bad = int((q.ganq_indices.cpu().numpy() != Qref).sum())
    print(f"conv2d quantize(): {bad} of {Qref.size} indices differ from the reference's")
    assert bad <= 2
    best = int(np.argmin(g["dists"]))
    assert rel_fro(q.ganq_codebook.cpu().numpy(), g["T"][best + 1]) < 1e-5 or bad > 0
    diff = (wq.float().cpu().numpy() - g["Wq"].astype(np.float32)).reshape(Qref.shape)
    assert (np.abs(diff) > np.spacing(np.abs(g["Wq"]).astype(np.float16)).astype(np.float32).reshape(Qref.shape)).sum() <= bad
    assert abs(avg_loss - float(g["avg_loss"])) < 1e-3 * float(g["avg_loss"])

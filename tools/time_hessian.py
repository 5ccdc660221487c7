"""Hessian kernel per launch for the two hand-over sizes: one sequence (2048 tokens) and a staged group (16384 tokens,
QuantizeConfig.ganq_hessian_stage_tokens); plus the whole add_batch path (staging copies included) per 2048 tokens."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ganq_amd import _lib
for n in (4096, 2048, 8192, 14336):
    for rows in (2048, 16384):
        X = (torch.randn(rows, n, device="cuda") * 0.5).half()
        H = torch.zeros(n, n, device="cuda")
        ns = 0
        for _ in range(2): _lib.hessian_accum(H, X, ns, rows // 2048); ns += rows // 2048
        torch.cuda.synchronize()
        _lib.profile_enable(True)
        for _ in range(5): _lib.hessian_accum(H, X, ns, rows // 2048); ns += rows // 2048
        rep = _lib.profile_report(); _lib.profile_enable(False)
        ms, cnt = rep["hessian_kernel"]
        us = ms / cnt * 1e3
        print(f"n={n} rows={rows}: {us:.1f} us per launch = {us * 2048 / rows:.1f} us per 2048 tokens -> "
              f"{2.0 * rows * n * n / (us * 1e-6) / 1e12:.0f} TFLOP/s ({2.0 * rows * n * n / (us * 1e-6) / 2.5e15:.2f} of the 2.5 PF fp16 peak)", flush=True)
        if _lib.hessian_t_supported(n, rows):  # the transposed staging path (hessian_w4.hip): the group as Xt [n, rows]
            Xt = X.t().contiguous()
            for _ in range(2): _lib.hessian_accum_t(H, Xt, rows, ns, rows // 2048); ns += rows // 2048
            torch.cuda.synchronize()
            _lib.profile_enable(True)
            for _ in range(5): _lib.hessian_accum_t(H, Xt, rows, ns, rows // 2048); ns += rows // 2048
            rep = _lib.profile_report(); _lib.profile_enable(False)
            ms, cnt = rep["hessian_kernel"]
            ut = ms / cnt * 1e3
            print(f"n={n} rows={rows}, transposed staging: {ut:.1f} us per group = {ut * 2048 / rows:.1f} us per 2048 tokens -> "
                  f"{2.0 * rows * n * n / (ut * 1e-6) / 1e12:.0f} TFLOP/s nominal, {1.0 * rows * n * n / (ut * 1e-6) / 1e15:.2f} PF executed ({us / ut:.2f} x the row-major kernels)", flush=True)
            del Xt
        del X, H
import torch.nn as nn
from ganq_amd.looper.named_module import NamedModule
from ganq_amd.quantization import GANQ, QuantizeConfig
n = 4096
lin = nn.Linear(n, 64, bias=False).half().cuda()
xs = [(torch.randn(1, 2048, n, device="cuda") * 0.5).half() for _ in range(16)]
for stage in (0, 16384):
    q = GANQ(NamedModule(lin, "fc", "layers.0.fc", 0), QuantizeConfig(bits=4, ganq_hessian_stage_tokens=stage))
    q.quantizer.configure(perchannel=True)
    for x in xs: q.add_batch(x, None)
    q.hessian; torch.cuda.synchronize()
    t0 = time.perf_counter()
    for rep in range(4):
        for x in xs: q.add_batch(x, None)
    q.hessian; torch.cuda.synchronize()
    print(f"add_batch path, n=4096, stage_tokens={stage}: {(time.perf_counter() - t0) / 64 * 1e6:.1f} us per 2048-token sequence (wall)")

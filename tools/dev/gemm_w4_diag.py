#!/usr/bin/env python3
"""developer: where the four-wave GEMM goes wrong -- structured operands, map of wrong outputs"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ganq_amd import _lib
torch.manual_seed(0)
def show(tag, y, want):
    y = y.float(); bad = ~torch.isclose(y, want, rtol=2e-3, atol=1e-2)
    print(f"{tag}: bad {int(bad.sum())} of {bad.numel()}; nan {int(torch.isnan(y).sum())} inf {int(torch.isinf(y).sum())}", flush=True)
    if bad.any():
        rows = bad.any(1).nonzero().flatten(); cols = bad.any(0).nonzero().flatten()
        print("   bad rows:", rows[:40].tolist(), "... count", rows.numel())
        print("   bad cols:", cols[:40].tolist(), "... count", cols.numel())
        i, j = bad.nonzero()[0].tolist()
        print(f"   first bad y[{i},{j}] = {float(y[i, j])} want {float(want[i, j])}; y[{i},{j}:{j+8}] = {y[i, j:j+8].tolist()}")
for (M, N, K) in [(256, 256, 64), (256, 256, 128), (256, 256, 256), (512, 512, 512)]:
    for kind in ("ones", "rows", "rand"):
        if kind == "ones":
            x = torch.ones(M, K, device="cuda").half(); w = torch.ones(N, K, device="cuda").half()
        elif kind == "rows":
            x = (torch.arange(M, device="cuda").float()[:, None] % 16 + 1).expand(M, K).contiguous().half()
            w = (torch.arange(N, device="cuda").float()[:, None] % 8 + 1).expand(N, K).contiguous().half() / 8
        else:
            x = torch.randn(M, K, device="cuda").half(); w = (0.05 * torch.randn(N, K, device="cuda")).half()
        want = x.float() @ w.float().T
        _lib.debug_option("GANQ_GEMM_H16_BM", 512)
        y = _lib.debug_gemm_h16(x, w)
        torch.cuda.synchronize()
        show(f"{M}x{N}x{K} {kind}", y, want)
_lib.debug_option("GANQ_GEMM_H16_BM", None)

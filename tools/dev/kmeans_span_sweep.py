#!/usr/bin/env python3
"""developer: windowed k-means, the spacing from which the levels are solved span by span (GANQ_KMEANS_SPAN) -- device time per shape"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ganq_amd import _lib
shapes = [(2048, 8192, 16), (1024, 14336, 16), (1024, 11008, 16), (1024, 5120, 16)]
for m, n, V in shapes:
    g = torch.Generator(device="cuda").manual_seed(0)
    W = 0.02 * torch.randn(m, n, device="cuda", generator=g)
    cw = (torch.rand(n, device="cuda", generator=g, dtype=torch.float64) + 0.5) ** 4
    ref = None
    out = []
    for span in (0, 128, 256, 512, 1024, 2048, 4096, 8192):
        _lib.debug_option("GANQ_KMEANS_SPAN", span)
        _lib.kmeans_init(W, cw, V)
        torch.cuda.synchronize()
        _lib.profile_enable(True)
        for _ in range(2): T0 = _lib.kmeans_init(W, cw, V)
        torch.cuda.synchronize()
        rep = _lib.profile_report(); _lib.profile_enable(False)
        ms, cnt = rep["kmeans_kernels"]
        if ref is None: ref = T0.clone()
        assert torch.equal(T0, ref), (n, span)
        out.append(f"{span}: {ms / cnt:.2f}")
    print(f"m={m} n={n}: " + "  ".join(out) + "  ms (same bits)", flush=True)
_lib.debug_option("GANQ_KMEANS_SPAN", None)

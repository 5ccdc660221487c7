#!/usr/bin/env python3
"""developer: race screen of csrc/gemm_h16.hip -- many shapes, repeated launches, every output element against the fp32 product of
the same fp16 operands (an LDS-DMA read placed a phase too early passes single runs and fails rarely: DESIGN.md, kernel table)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ganq_amd import _lib
g = torch.Generator(device="cuda").manual_seed(1)
shapes = [(4096, 4096, 4096), (2048, 4096, 4096), (4096, 14336, 4096), (4096, 4096, 14336), (1000, 768, 3072), (3000, 3072, 768),
          (256, 256, 8192), (1280, 2048, 2048), (2048, 8192, 2048), (2048, 2048, 8192), (513, 1028, 192), (4096, 512, 2048)]
bad = 0
for (M, N, K) in shapes:
    x = torch.randn(M, K, device="cuda", generator=g).half()
    w = (0.05 * torch.randn(N, K, device="cuda", generator=g)).half()
    want = (x.float() @ w.float().T)
    scale = float(want.abs().max())
    for bm in (0, 128, 256, 512):
        _lib.debug_option("GANQ_GEMM_H16_BM", bm if bm else None)
        worst = 0.0
        for rep in range(8):
            y = _lib.debug_gemm_h16(x, w)
            err = float((y.float() - want).abs().max()) / scale
            worst = max(worst, err)
        ok = worst < 2e-3
        bad += 0 if ok else 1
        print(f"{M}x{N}x{K} bm={bm}: worst max-error / max|y| over 8 launches {worst:.2e} {'ok' if ok else 'FAIL'}", flush=True)
_lib.debug_option("GANQ_GEMM_H16_BM", None)
print("failures:", bad)
sys.exit(1 if bad else 0)

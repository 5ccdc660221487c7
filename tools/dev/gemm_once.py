#!/usr/bin/env python3
"""developer: a few launches of the dense GEMM (csrc/gemm_h16.hip) for counter runs: python tools/dev/gemm_once.py [M N K]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ganq_amd import _lib
M, N, K = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (4096, 4096, 4096)
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(M, K, device="cuda", generator=g).half()
w = (0.02 * torch.randn(N, K, device="cuda", generator=g)).half()
for _ in range(4):
    y = _lib.debug_gemm_h16(x, w)
torch.cuda.synchronize()

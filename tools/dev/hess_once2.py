#!/usr/bin/env python3
"""developer: a few staged-group Hessian launches (16384 x n tokens) for rocprofv3 runs: python tools/dev/hess_once2.py [n]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ganq_amd import _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
X = (torch.randn(16384, n, device="cuda") * 0.5).half()
H = torch.zeros(n, n, device="cuda")
ns = 0
for _ in range(6):
    _lib.hessian_accum(H, X, ns, 8); ns += 8
torch.cuda.synchronize()

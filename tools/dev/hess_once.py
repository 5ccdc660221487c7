#!/usr/bin/env python3
"""developer: a few launches of the Hessian kernel at one size (for counter runs): hess_once.py n rows"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ganq_amd import _lib
n, rows = int(sys.argv[1]), int(sys.argv[2])
X = (torch.randn(rows, n, device="cuda") * 0.5).half()
H = torch.zeros(n, n, device="cuda")
ns = 0
for _ in range(4):
    _lib.hessian_accum(H, X, ns, rows // 2048); ns += rows // 2048
torch.cuda.synchronize()
print(float(H.abs().mean()))

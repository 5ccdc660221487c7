#!/usr/bin/env python3
"""developer: a few staged-group Hessian launches at one shape (for the counter passes of tools/dev/sq_pmc.sh)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ganq_amd import _lib
rows, n = (int(v) for v in (sys.argv[1:3] if len(sys.argv) > 2 else (16384, 4096)))
X = (torch.randn(rows, n, device="cuda") * 0.5).half()
H = torch.zeros(n, n, device="cuda")
for i in range(3): _lib.hessian_accum(H, X, 8 * i, 8)
torch.cuda.synchronize()

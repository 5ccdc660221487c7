import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ganq_amd import _lib
n, rows = 4096, 2048
X = (torch.randn(rows, n, device="cuda") * 0.5).half()
H = torch.zeros(n, n, device="cuda")
for i in range(4): _lib.hessian_accum(H, X, i, 1)
torch.cuda.synchronize()

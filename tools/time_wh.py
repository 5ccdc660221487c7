"""Time the W @ H_fixed preparation of the T-update (t_prepare) via rocprofv3-free host timing of the debug entry, and report
its error against fp64.  usage: python tools/time_wh.py [m n]   (GANQ_WH_F64=1 selects the fp64 GEMM)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ganq_amd import _lib
m, n = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4096, 4096)
torch.manual_seed(0)
W = (0.02 * torch.randn(m, n, device="cuda")).half().float()
X = torch.randn(2 * n, n, device="cuda") * (0.1 + torch.rand(n, device="cuda"))
H = (X.T @ X) / X.shape[0]
WH, Hf = _lib.debug_wh_product(W, H)
ref = W.double() @ Hf
print("rel err vs fp64:", float((WH - ref).norm() / ref.norm()), " fp32 matmul:", float(((W @ Hf.float()).double() - ref).norm() / ref.norm()))
W2 = W + 1e-6 * torch.randn_like(W)
WH2, _ = _lib.debug_wh_product(W2, H)
ref2 = W2.double() @ Hf
print("fp32-valued W rel err:", float((WH2 - ref2).norm() / ref2.norm()))

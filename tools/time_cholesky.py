import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ganq_amd import _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
X = torch.randn(2 * n, n, device="cuda") * (0.1 + torch.rand(n, device="cuda"))
H = (X.T @ X) / X.shape[0]
H += 0.01 * H.diag().mean() * torch.eye(n, device="cuda")
for _ in range(3): _lib.cholesky(H, check=False)
torch.cuda.synchronize()

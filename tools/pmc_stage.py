"""Run each stage of the HIP path a few times on the synthetic 4096x4096 layer (target for rocprofv3 --pmc)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ganq_amd import _lib  # noqa: E402

m = n = int(os.environ.get("GANQ_N", "4096"))
V = 16
torch.manual_seed(0)
dev = "cuda"
W = (0.02 * torch.randn(m, n)).half().float().to(dev)
p = 2 * n
X = (torch.randn(p, n, device=dev) * (0.1 + torch.rand(n, device=dev)))
H = (2.0 / p) * (X.T @ X)
H += 0.01 * H.diag().mean() * torch.eye(n, device=dev)
off = (H.abs().sum(1) - 2 * H.diag()).clamp(min=1e-8)
L = torch.linalg.cholesky(H + torch.diag(off))
qs = (torch.arange(V, device=dev) + 0.5) / V
T0 = torch.quantile(W, qs, dim=1).T.contiguous()
_lib.selftest()
WH = _lib.matmul_f32(W, H)
for _ in range(int(os.environ.get("GANQ_REPS", "2"))):
    Q = _lib.solve_s(W, L, T0)
    T1 = _lib.update_t(WH, H, Q, V)
    d = _lib.quad_loss(W, H, T1, Q)
torch.cuda.synchronize()
print("done", float(d))

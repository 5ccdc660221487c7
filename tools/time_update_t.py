"""Device time of the T-update kernels at 4096x4096 (HIP events inside the library)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ganq_amd import _lib
m = n = int(os.environ.get("N", 4096)); V = 16
torch.manual_seed(0)
W = (0.02 * torch.randn(m, n)).cuda()
X = torch.randn(n, 2 * n, device="cuda") * 0.05
H = (X @ X.T / (2 * n)).contiguous()
Q = torch.randint(0, V, (m, n), device="cuda", dtype=torch.uint8)
for _ in range(2): _lib.update_t(W, H, Q, V)
torch.cuda.synchronize()
_lib.profile_enable(True)
for _ in range(5): _lib.update_t(W, H, Q, V)
torch.cuda.synchronize()
rep = _lib.profile_report(); _lib.profile_enable(False)
for k, (ms, cnt) in rep.items():
    if cnt: print(f"{k:24s} {ms / cnt:8.3f} ms x {cnt}")

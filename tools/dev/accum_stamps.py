"""developer: cycle stamps of onehot_accum_kernel (workgroup 0, wave 0) on a 4096 x 4096 layer (GANQ_ACCUM_DEBUG=1)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ganq_amd import _lib
_lib.selftest()
m = n = 4096; V = 16
torch.manual_seed(0)
W = (0.02 * torch.randn(m, n)).cuda()
X = torch.randn(2 * n, n, device="cuda") * (0.1 + torch.rand(n, device="cuda"))
H = (X.T @ X) / n; H += 0.01 * H.diag().mean() * torch.eye(n, device="cuda")
Q = torch.randint(0, V, (m, n), dtype=torch.uint8, device="cuda")
WH = _lib.matmul_f32(W, H)
for _ in range(2): _lib.update_t(WH, H, Q, V)
torch.cuda.synchronize()
_lib.debug_option("GANQ_ACCUM_DEBUG", 1)
_lib.update_t(WH, H, Q, V)
torch.cuda.synchronize()
_lib.debug_option("GANQ_ACCUM_DEBUG", None)
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(5): _lib.update_t(WH, H, Q, V)
e.record(); torch.cuda.synchronize()
print(f"update_t (prepare + accumulation + solve): {s.elapsed_time(e) / 5:.3f} ms")

import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ganq_amd import _lib
m = n = 4096; V = 16
torch.manual_seed(0)
W = (0.02 * torch.randn(m, n)).cuda()
L = torch.tril(torch.randn(n, n)).cuda() * 0.01 + torch.eye(n).cuda()
T0 = torch.sort(0.02 * torch.randn(m, V))[0].cuda()
_lib.selftest()
for _ in range(2): _lib.solve_s(W, L, T0)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(5): _lib.solve_s(W, L, T0)
e.record(); torch.cuda.synchronize()
print("mode", os.environ.get("GANQ_SOLVE_S_DEBUG_MODE", "0"), "solve_s ms", s.elapsed_time(e) / 5)

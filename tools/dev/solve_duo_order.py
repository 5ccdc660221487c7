"""developer: does asking for helper workgroups cost a launch that cannot have them (4096 rows = 256 tiles)?  alternating A/B"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ganq_amd import _lib
_lib.selftest()
m, n, V = 4096, 4096, 16
torch.manual_seed(0)
W = (0.02 * torch.randn(m, n)).cuda()
L = torch.tril(torch.randn(n, n)).cuda() * 0.01 + torch.eye(n).cuda()
T0 = torch.sort(0.02 * torch.randn(m, V))[0].cuda()
for duo in (0, 1, 0, 1, 1, 0):
    _lib.debug_option("GANQ_SOLVE_DUO", duo)
    for _ in range(2): q = _lib.solve_s(W, L, T0)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): q = _lib.solve_s(W, L, T0)
    e.record(); torch.cuda.synchronize()
    print(f"GANQ_SOLVE_DUO={duo}: {s.elapsed_time(e) / 10:.4f} ms", flush=True)

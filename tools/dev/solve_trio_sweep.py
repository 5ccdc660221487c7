"""developer sweep of the split policy with two helper workgroups per tile (GANQ_SOLVE_TRIO_XA / _XB)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ganq_amd import _lib
_lib.selftest()
V = 16
shapes = [(928, 4096), (1232, 4096), (1024, 8192), (768, 3072)]
data = {}
for (m, n) in shapes:
    torch.manual_seed(0)
    data[(m, n)] = ((0.02 * torch.randn(m, n)).cuda(), torch.tril(torch.randn(n, n)).cuda() * 0.01 + torch.eye(n).cuda(),
                    torch.sort(0.02 * torch.randn(m, V))[0].cuda())
def run(m, n):
    W, L, T0 = data[(m, n)]
    for _ in range(2): q = _lib.solve_s(W, L, T0)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5): q = _lib.solve_s(W, L, T0)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / 5
run(928, 4096)
_lib.debug_option("GANQ_SOLVE_TRIO", 0)
print("one helper:", "  ".join(f"{m}x{n} {run(m, n):.3f}" for (m, n) in shapes), flush=True)
_lib.debug_option("GANQ_SOLVE_TRIO", 1)
for pol in [(26, 8), (26, 4), (20, 4), (20, 8), (32, 8), (32, 12), (16, 2), (12, 0), (38, 12), (26, 12), (8, 0), (20, 0)]:
    _lib.debug_option("GANQ_SOLVE_TRIO_XA", pol[0]); _lib.debug_option("GANQ_SOLVE_TRIO_XB", pol[1])
    print(f"{pol}:", "  ".join(f"{m}x{n} {run(m, n):.3f}" for (m, n) in shapes), flush=True)

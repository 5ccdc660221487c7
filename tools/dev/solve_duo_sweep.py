"""developer sweep of the helper-workgroup split policy of the S-solve (GANQ_SOLVE_DUO_XA / XB / XMIN / CMIN)"""
import os, sys, itertools, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ganq_amd import _lib
_lib.selftest()
V = 16
shapes = [(928, 4096), (2048, 8192), (2048, 2048), (768, 3072)]
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
_lib.debug_option("GANQ_SOLVE_DUO", 0)
print("without:", "  ".join(f"{m}x{n} {run(m, n):.3f}" for (m, n) in shapes), flush=True)
_lib.debug_option("GANQ_SOLVE_DUO", 1)
pols = [(44, 12, 2, 8), (44, 10, 2, 8), (44, 8, 2, 8), (48, 12, 2, 8), (52, 16, 2, 8), (52, 12, 2, 8), (44, 10, 3, 8), (44, 10, 2, 12),
        (44, 10, 2, 5), (40, 8, 2, 8), (40, 6, 2, 8), (36, 4, 2, 8), (48, 10, 2, 8), (56, 16, 2, 8)]
for pol in pols:
    for name, v in zip(("GANQ_SOLVE_DUO_XA", "GANQ_SOLVE_DUO_XB", "GANQ_SOLVE_DUO_XMIN", "GANQ_SOLVE_DUO_CMIN"), pol):
        _lib.debug_option(name, v)
    print(f"{pol}:", "  ".join(f"{m}x{n} {run(m, n):.3f}" for (m, n) in shapes), flush=True)

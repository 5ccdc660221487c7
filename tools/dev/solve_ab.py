"""developer A/B: S-solve with the threshold fast path (default) vs the reduction path only (GANQ_SOLVE_VARIANT=1)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ganq_amd import _lib
_lib.selftest()
V = 16
for (m, n) in [(4096, 4096), (8192, 2048), (2048, 2048), (3072, 768), (768, 3072), (2048, 8192), (14336, 4096)]:
    torch.manual_seed(0)
    W = (0.02 * torch.randn(m, n)).cuda()
    L = torch.tril(torch.randn(n, n)).cuda() * 0.01 + torch.eye(n).cuda()
    T0 = torch.sort(0.02 * torch.randn(m, V))[0].cuda()
    res = {}
    for variant in (0, 1):
        _lib.debug_option("GANQ_SOLVE_VARIANT", variant)
        for _ in range(2): q = _lib.solve_s(W, L, T0)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5): q = _lib.solve_s(W, L, T0)
        e.record(); torch.cuda.synchronize()
        res[variant] = (s.elapsed_time(e) / 5, q)
    if not os.environ.get("GANQ_AB_NOCHECK"): assert torch.equal(res[0][1], res[1][1])
    print(f"{m}x{n}: thresholds {res[0][0]:.3f} ms, reductions only {res[1][0]:.3f} ms  ({res[1][0] / res[0][0]:.2f}x)")
_lib.debug_option("GANQ_SOLVE_VARIANT", None)

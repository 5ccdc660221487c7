"""developer A/B: S-solve with helper workgroups (default where the launch has at most half as many tiles as CUs) vs without
(GANQ_SOLVE_DUO=0); indices must be identical.  Optional args: XA XB XMIN CMIN of the split policy."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ganq_amd import _lib
_lib.selftest()
V = 16
pol = [int(a) for a in sys.argv[1:5]]
for name, v in zip(("GANQ_SOLVE_DUO_XA", "GANQ_SOLVE_DUO_XB", "GANQ_SOLVE_DUO_XMIN", "GANQ_SOLVE_DUO_CMIN"), pol):
    _lib.debug_option(name, v)
shapes = [(2048, 2048), (2048, 4096), (928, 4096), (1232, 4096), (512, 4096), (2048, 8192), (1024, 8192), (768, 3072), (512, 2048), (1024, 1024), (4096, 4096)]
for (m, n) in shapes:
    torch.manual_seed(0)
    W = (0.02 * torch.randn(m, n)).cuda()
    L = torch.tril(torch.randn(n, n)).cuda() * 0.01 + torch.eye(n).cuda()
    T0 = torch.sort(0.02 * torch.randn(m, V))[0].cuda()
    res = {}
    for duo in (1, 2, 0):  # helpers (two per tile where a third of the chip holds the tiles) / one helper only / none
        _lib.debug_option("GANQ_SOLVE_DUO", 1 if duo else 0)
        _lib.debug_option("GANQ_SOLVE_TRIO", 1 if duo == 1 else 0)
        for _ in range(2): q = _lib.solve_s(W, L, T0)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5): q = _lib.solve_s(W, L, T0)
        e.record(); torch.cuda.synchronize()
        res[duo] = (s.elapsed_time(e) / 5, q)
    same = torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][1], res[2][1])
    print(f"{m}x{n}: helpers {res[1][0]:.3f} ms (one per tile: {res[2][0]:.3f}), without {res[0][0]:.3f} ms  ({res[0][0] / res[1][0]:.2f}x)  identical={same}", flush=True)
    assert same
_lib.debug_option("GANQ_SOLVE_DUO", None); _lib.debug_option("GANQ_SOLVE_TRIO", None)

"""developer: Hessian accumulation of one staged group (16384 tokens) under forced cut plans: bulk (whole tiles) / parts"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ganq_amd import _lib
_lib.selftest()
n = int(sys.argv[1]); rows = 16384
torch.manual_seed(0)
X = (torch.randn(rows, n, device="cuda") * 0.5).half()
def run(wide, bulk, parts):
    _lib.debug_option("GANQ_HESS_WIDE", wide); _lib.debug_option("GANQ_HESS_BULK", bulk); _lib.debug_option("GANQ_HESS_PARTS", parts)
    H = torch.zeros(n, n, device="cuda"); ns = 0
    for _ in range(2): _lib.hessian_accum(H, X, ns, 8); ns += 8
    torch.cuda.synchronize()
    _lib.profile_enable(True)
    for _ in range(5): _lib.hessian_accum(H, X, ns, 8); ns += 8
    rep = _lib.profile_report(); _lib.profile_enable(False)
    ms, cnt = rep["hessian_kernel"]
    return ms / cnt * 1e3
for wide, bulk, parts in [(0, -1, 0), (2, -1, 0), (2, 0, 2), (2, 0, 3), (2, 0, 4), (2, 256, 2), (0, 0, 2), (0, 256, 2), (0, 256, 3), (0, 0, 3)]:
    print(f"n={n} wide={wide} bulk={bulk} parts={parts}: {run(wide, bulk, parts):.1f} us", flush=True)

import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ganq_amd import _lib
for n in (4096, 2048, 8192):
    rows = 2048
    X = (torch.randn(rows, n, device="cuda") * 0.5).half()
    H = torch.zeros(n, n, device="cuda")
    ns = 0
    for _ in range(3): _lib.hessian_accum(H, X, ns, 1); ns += 1
    torch.cuda.synchronize()
    _lib.profile_enable(True)
    for _ in range(10): _lib.hessian_accum(H, X, ns, 1); ns += 1
    rep = _lib.profile_report(); _lib.profile_enable(False)
    ms, cnt = rep["hessian_kernel"]
    us = ms / cnt * 1e3
    print(f"n={n} rows={rows}: {us:.1f} us  -> {2.0 * rows * n * n / (us * 1e-6) / 1e12:.0f} TFLOP/s nominal", flush=True)

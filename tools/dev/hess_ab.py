import os, sys, torch
sys.path.insert(0, os.getcwd())
from ganq_amd import _lib
for n in (4096, 8192, 11008, 14336):
    rows = 16384
    X = (torch.randn(rows, n, device="cuda") * 0.5).half()
    H = torch.zeros(n, n, device="cuda")
    ns = 0
    for _ in range(2): _lib.hessian_accum(H, X, ns, 8); ns += 8
    torch.cuda.synchronize()
    _lib.profile_enable(True)
    for _ in range(5): _lib.hessian_accum(H, X, ns, 8); ns += 8
    rep = _lib.profile_report(); _lib.profile_enable(False)
    ms, cnt = rep["hessian_kernel"]
    print(f"n={n}: {ms / cnt * 1e3:.1f} us per 16384 tokens", flush=True)
    del X, H

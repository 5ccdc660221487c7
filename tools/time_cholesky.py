#!/usr/bin/env python3
"""device time of ganq_cholesky (the prologue runs it twice per layer: gptq.py:289-308 of the reference) -- `python tools/time_cholesky.py [n ...]`;
under rocprofv3 --kernel-trace --stats it gives the per-kernel split (chol_diag / chol_panel / chol_update)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ganq_amd import _lib
for n in ([int(a) for a in sys.argv[1:]] or [2048, 4096, 8192]):
    X = torch.randn(2 * n, n, device="cuda") * (0.1 + torch.rand(n, device="cuda"))
    H = (X.T @ X) / X.shape[0]
    H += 0.01 * H.diag().mean() * torch.eye(n, device="cuda")
    for _ in range(3): _lib.cholesky(H, check=False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record()
    for _ in range(reps): L, _info = _lib.cholesky(H, check=False)
    e1.record(); torch.cuda.synchronize()
    err = float((L.double() @ L.double().T - H.double()).abs().max() / H.double().abs().max())
    print(f"n={n}: {e0.elapsed_time(e1) / reps:.3f} ms per factorisation; max |L L^T - H| / max |H| = {err:.2e}", flush=True)

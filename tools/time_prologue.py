"""Time the torch ops of the inherited GPTQ prologue (gptq.py:quantize) on the GPU."""
import sys, time, torch
for n in (768, 2048, 4096, 8192):
    X = torch.randn(4 * n if n <= 4096 else 2 * n, n, device="cuda") * (0.1 + torch.rand(n, device="cuda"))
    H = (X.T @ X) / X.shape[0]
    H += 0.01 * H.diag().mean() * torch.eye(n, device="cuda")
    def t(f, reps=3):
        f(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps): r = f()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3, r
    t_ch, L = t(lambda: torch.linalg.cholesky(H))
    t_inv, Hi = t(lambda: torch.cholesky_inverse(L))
    t_chu, U = t(lambda: torch.linalg.cholesky(Hi, upper=True))
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from ganq_amd import _lib
    t_hip, Lh = t(lambda: _lib.cholesky(H, check=False)[0])
    err = float((Lh - L).norm() / L.norm())
    print(f"n={n}: ganq_cholesky {t_hip:.2f} ms (vs torch {err:.1e})", flush=True)
    print(f"n={n}: cholesky {t_ch:.2f} ms, cholesky_inverse {t_inv:.2f} ms, cholesky(upper) {t_chu:.2f} ms", flush=True)

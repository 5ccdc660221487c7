"""Per-stage device timing of the HIP path on the synthetic 4096x4096 layer (developer tool)."""
import argparse
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ganq_amd import _lib  # noqa: E402


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        out = fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=4096)
    ap.add_argument("--n", type=int, default=4096)
    ap.add_argument("--bits", type=int, default=4)
    ap.add_argument("--K", type=int, default=10)
    a = ap.parse_args()
    m, n, V = a.m, a.n, 2 ** a.bits
    torch.manual_seed(0)
    dev = "cuda"
    W = (0.02 * torch.randn(m, n)).half().float().to(dev)
    p = 4 * n
    X = (torch.randn(p, n, device=dev) * (0.1 + torch.rand(n, device=dev)))
    H = (2.0 / p) * (X.T @ X)
    H += 0.01 * H.diag().mean() * torch.eye(n, device=dev)
    off = (H.abs().sum(1) - 2 * H.diag()).clamp(min=1e-8)
    L = torch.linalg.cholesky(H + torch.diag(off))
    qs = (torch.arange(V, device=dev) + 0.5) / V
    T0 = torch.quantile(W, qs, dim=1).T.contiguous()
    _lib.selftest()
    t_mm, WH = timed(lambda: _lib.matmul_f32(W, H))
    print(f"matmul_f32 {m}x{n}x{n}: {t_mm:.3f} ms  ({2*m*n*n/t_mm/1e9:.1f} TFLOP/s)  err vs torch "
          f"{(WH - W @ H).norm() / (W @ H).norm():.2e}")
    t_s, Q = timed(lambda: _lib.solve_s(W, L, T0))
    print(f"solve_s: {t_s:.3f} ms  ({m*n*n/t_s/1e9:.1f} TFLOP/s of residual GEMM)")
    t_u, T1 = timed(lambda: _lib.update_t(WH, H, Q, V))
    print(f"update_t: {t_u:.3f} ms")
    t_l, d = timed(lambda: _lib.quad_loss(W, H, T1, Q))
    print(f"quad_loss: {t_l:.3f} ms  dist={float(d):.6g}")
    ws = _lib.run_layer_workspace(m, n, V, dev)
    t_r, out = timed(lambda: _lib.run_layer(W, H, L, T0, a.K, workspace=ws), reps=2)
    print(f"run_layer K={a.K}: {t_r:.3f} ms -> {n / (t_r / 1e3):.1f} columns/s;  dists={out[2].cpu().numpy()} "
          f"best_k={int(out[3])}")


if __name__ == "__main__":
    main()

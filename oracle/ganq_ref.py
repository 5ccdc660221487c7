"""torch-CPU restatement of the reference loop that keeps the reference's OWN op sequence.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  ``c_oracle`` fixes one summation order so
that it can be matched bit-for-bit on the GPU; this module instead does what the reference's
torch branch does op for op -- per column: gather the already-assigned codebook values and
take one gemv against a column of L (ganq.py:536-565); per iteration: batched
``lstsq(driver="gelsd")`` on S H S^T (ganq.py:589-591) and the quadratic loss
(ganq.py:392-395, :621-626) -- so its wall time on the GPU box's host cores is the
"reference CPU path" number that bench.py reports beside the HIP path, and so that the
golden vectors can be cross-checked against a second, independent restatement.
"""
import time

import torch


@torch.no_grad()
def solve_s(W: torch.Tensor, L: torch.Tensor, T: torch.Tensor) -> torch.Tensor:
    """ganq.py:533-565.  W [m,n] f32, L [n,n] lower, T [m,V] -> Q [m,n] int64."""
    m, n = W.shape
    Q = torch.zeros(m, n, dtype=torch.long)
    resid = torch.zeros(m, 1)
    for j in range(n - 1, -1, -1):
        target = W[:, j : j + 1] + resid / L[j, j]
        Q[:, j] = (target - T).abs().argmin(dim=1)
        tail_err = W[:, j:] - T.gather(1, Q[:, j:])
        resid = tail_err @ L[j:, j - 1].unsqueeze(-1)  # column j-1 (wraps to n-1 at j == 0; unused)
    return Q


@torch.no_grad()
def update_t(W: torch.Tensor, H: torch.Tensor, Q: torch.Tensor, V: int) -> torch.Tensor:
    """ganq.py:576-591 (CPU branch): T = lstsq(S H S^T, S (W H)^T, gelsd)."""
    m, n = W.shape
    S = torch.zeros(m, V, n)
    S.scatter_(1, Q.unsqueeze(1), 1.0)
    lhs = S @ H @ S.mT
    rhs = S @ (W @ H).unsqueeze(1).mT
    return torch.linalg.lstsq(lhs, rhs, driver="gelsd").solution.mT.squeeze(-2)


@torch.no_grad()
def quad_loss(W: torch.Tensor, H: torch.Tensor, T: torch.Tensor, Q: torch.Tensor) -> float:
    """ganq.py:392-395 on Wq = T.gather(1, Q)."""
    E = W - T.gather(1, Q)
    return float((E.mm(H) * E).sum())


@torch.no_grad()
def run_layer(W, H, L, T0, K, alias_q=True, timings=None):
    """ganq.py:516-634.  Returns (T_best, Q_out, dists, best_k); see c_oracle.run_layer for alias_q."""
    V = T0.shape[1]
    T = T0.clone()
    best = (float("inf"), None, None, -1)
    dists = []
    Q = None
    for k in range(K):
        t0 = time.perf_counter()
        Q = solve_s(W, L, T)
        t1 = time.perf_counter()
        T = update_t(W, H, Q, V)
        t2 = time.perf_counter()
        d = quad_loss(W, H, T, Q)
        t3 = time.perf_counter()
        if timings is not None:
            timings.append((t1 - t0, t2 - t1, t3 - t2))
        dists.append(d)
        if d < best[0]:
            best = (d, T, Q.clone(), k)
    Q_out = Q if alias_q else best[2]
    return best[1], Q_out, dists, best[3]


@torch.no_grad()
def outlier_split(W: torch.Tensor, ratio: float):
    """Outlier extraction of the paper (Appendix A, Algorithm 2; paper.md:884-899 -- the reference repository has no
    code for it, so this restatement is pinned by the published pseudocode only: "parity unpinned").
    W [m,n] f32 -> (W_sparse, W_dense, mask, c_lower [m], c_upper [m]); ties with a cut-off value are outliers."""
    import math

    m, n = W.shape
    p = 1.0 - 0.5 * ratio
    upper = min(max(int(math.floor(n * p)), 0), n - 1)
    lower = min(max(int(math.ceil(n * (1.0 - p))), 0), n - 1)
    srt = torch.sort(W, dim=1).values
    c_upper, c_lower = srt[:, upper], srt[:, lower]
    mask = (W >= c_upper[:, None]) | (W <= c_lower[:, None])
    W_sparse = torch.where(mask, W, torch.zeros_like(W))
    return W_sparse, W - W_sparse, mask, c_lower, c_upper

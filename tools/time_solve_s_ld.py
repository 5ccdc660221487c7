"""S-solve time against the leading dimension of L (power-of-two row strides vs padded ones)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ganq_amd import _lib
m = n = 4096; V = 16
torch.manual_seed(0)
W = (0.02 * torch.randn(m, n)).cuda()
L0 = torch.tril(torch.randn(n, n)).cuda() * 0.01 + torch.eye(n).cuda()
T0 = torch.sort(0.02 * torch.randn(m, V))[0].cuda()
_lib.selftest()
Qref = None
for pad in (0, 16, 32, 64, 128, 1056):
    Lp = torch.zeros(n, n + pad, device="cuda")
    Lp[:, :n] = L0
    L = Lp[:, :n]
    for _ in range(2): Q = _lib.solve_s(W, L, T0)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5): _lib.solve_s(W, L, T0)
    e.record(); torch.cuda.synchronize()
    if Qref is None: Qref = Q.clone()
    print(f"ldl = n + {pad:5d}: solve_s {s.elapsed_time(e) / 5:.3f} ms  same Q: {bool(torch.equal(Q, Qref))}", flush=True)

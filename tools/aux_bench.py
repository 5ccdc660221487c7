"""Everything around the loop once, for `rocprofv3 --kernel-trace --stats`: Hessian batches, the two Cholesky
factorisations of the prologue, k-means, packing and the LUT forward at decode sizes (4096 x 4096 layer, 4 bit)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ganq_amd import _lib
m = n = 4096; V = 16
g = torch.Generator(device="cuda").manual_seed(0)
W = (0.02 * torch.randn(m, n, device="cuda", generator=g)).half().float()
H = torch.zeros(n, n, device="cuda")
for i in range(16):
    X = (torch.randn(2048, n, device="cuda", generator=g) * 0.5).half()
    _lib.hessian_accum(H, X, i, 1)
H += 0.01 * H.diag().mean() * torch.eye(n, device="cuda")
for _ in range(2):
    L = _lib.cholesky(H)
    Lr = _lib.cholesky(torch.flip(H, dims=(0, 1)))
cw = (torch.rand(n, device="cuda", generator=g, dtype=torch.float64) + 0.5) ** 4
T0 = _lib.kmeans_init(W, cw, V)
Q = torch.randint(0, V, (m, n), device="cuda", generator=g, dtype=torch.uint8)
qw = _lib.pack_indices(Q, 4)
lut = T0.half()
for M in (1, 16):
    x = torch.randn(M, n, device="cuda", generator=g).half()
    for _ in range(50):
        _lib.lut_linear(x, qw, lut, None, 4)
torch.cuda.synchronize()
print("done")

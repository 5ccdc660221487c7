import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ganq_amd import _lib
_lib.selftest()
m = n = 4096; V = 16
g = torch.Generator(device="cuda").manual_seed(0)
X = torch.randn(2 * n, n, device="cuda", generator=g) * (0.1 + torch.rand(n, device="cuda", generator=g))
H = (2.0 / X.shape[0]) * (X.T @ X); H = 0.5 * (H + H.T)
Q = torch.randint(0, V, (m, n), device="cuda", generator=g, dtype=torch.uint8)
WH = torch.randn(m, n, device="cuda", generator=g)
_lib.update_t(WH, H, Q, V); torch.cuda.synchronize()
_lib.debug_option("GANQ_ACCUM_DEBUG", 1)
_lib.update_t(WH, H, Q, V); torch.cuda.synchronize()
_lib.debug_option("GANQ_ACCUM_DEBUG", None)

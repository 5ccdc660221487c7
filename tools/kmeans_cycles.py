"""Debug build only (make EXTRA=-DGANQ_KMEANS_DEBUG): where the resident k-means kernel spends its cycles."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ganq_amd import _lib
m, n, V = 4096, 4096, 16
g = torch.Generator(device="cuda").manual_seed(0)
W = 0.02 * torch.randn(m, n, device="cuda", generator=g)
cw = (torch.rand(n, device="cuda", generator=g, dtype=torch.float64) + 0.5) ** 4
out = (ctypes.c_ulonglong * 32)()
L = _lib.lib()
L.ganq_debug_kmeans_cycles(out)
_lib.kmeans_init(W, cw, V)
L.ganq_debug_kmeans_cycles(out)
names = {0: "sort", 1: "prefix", 2: "dprev fill", 3: "full scan", 20: "backtrack"}
tot = sum(out)
for k in range(32):
    if out[k]:
        nm = names.get(k, f"level hs=2^{k - 4}")
        print(f"{nm:18s} {out[k] / 1e6:10.1f} Mcycles  {100.0 * out[k] / tot:5.1f}%")

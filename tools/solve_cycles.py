"""Debug build only (make EXTRA=-DGANQ_SOLVE_DEBUG): per-role cycle totals of the S-solve, averaged per workgroup."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ganq_amd import _lib
m = n = 4096; V = 16
torch.manual_seed(0)
W = (0.02 * torch.randn(m, n)).cuda()
L = torch.tril(torch.randn(n, n)).cuda() * 0.01 + torch.eye(n).cuda()
T0 = torch.sort(0.02 * torch.randn(m, V))[0].cuda()
_lib.selftest()
out = (ctypes.c_ulonglong * 8)()
lib = _lib.lib()
wgs = m // 16
names = {0: "as shipped", 1: "no B loads", 2: "no A reads", 3: "no MFMA (one VALU fma per group)", 5: "no panel steps (G alone)"}
for mode in (0, 1, 2, 3, 5):
    os.environ["GANQ_SOLVE_DBG"] = str(mode)
    _lib.solve_s(W, L, T0)
    lib.ganq_debug_solve_cycles(out)
    _lib.solve_s(W, L, T0)
    lib.ganq_debug_solve_cycles(out)
    print(f"mode {mode} ({names[mode]}): " + "  ".join(f"{nm} {out[k] / wgs / 1e6:.3f}" for k, nm in enumerate(["P", "G1", "toA"])), flush=True)

import os, sys, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ganq_amd import _lib
m = n = 4096; V = 16
torch.manual_seed(0)
W = (0.02 * torch.randn(m, n)).cuda()
L = torch.tril(torch.randn(n, n)).cuda() * 0.01 + torch.eye(n).cuda()
T0 = torch.sort(0.02 * torch.randn(m, V))[0].cuda()
_lib.selftest()
h = ctypes.CDLL(_lib.LIB_PATH)
out = (ctypes.c_ulonglong * 4)()
_lib.solve_s(W, L, T0); torch.cuda.synchronize()
h.ganq_debug_solve_counters(out)
_lib.solve_s(W, L, T0); torch.cuda.synchronize()
h.ganq_debug_solve_counters(out)
print("fast panels", out[0], "redone", out[1], "waves-panels without fast path", out[2], "| memtime ticks in the fast steps of wg0 wave0:", out[3], "=", out[3] / 4096.0, "per step (100 MHz ticks x 24 = core cycles at 2.4 GHz)")

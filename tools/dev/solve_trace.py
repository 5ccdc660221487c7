"""developer: per-panel timeline of one workgroup of the S-solve (build with -DGANQ_SOLVE_TRACE, see tools/dev/build_variant.sh).
P = wave 0 (column steps), G = wave 4 (residual chain).  Prints, per group of panels, the average time each role spends in
its phases and waiting at the two barriers of a step."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ganq_amd import _lib
m = int(os.environ.get("M", 4096)); n = int(os.environ.get("N", 4096)); V = 16
torch.manual_seed(0)
W = (0.02 * torch.randn(m, n)).cuda()
X = torch.randn(2 * n, n, device="cuda") * (0.1 + torch.rand(n, device="cuda"))
H = (X.T @ X) / n; H += 0.01 * H.diag().mean() * torch.eye(n, device="cuda")
L = torch.linalg.cholesky(H + torch.diag((H.abs().sum(1) - 2 * H.diag()).clamp(min=1e-8)))
T0 = torch.quantile(W[:, ::8], torch.linspace(0.03, 0.97, V, device="cuda"), dim=1).T.contiguous()
_lib.selftest()
if os.environ.get("VARIANT"): _lib.debug_option("GANQ_SOLVE_VARIANT", int(os.environ["VARIANT"]))
h = ctypes.CDLL(_lib.LIB_PATH)
for _ in range(2): _lib.solve_s(W, L, T0)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record(); _lib.solve_s(W, L, T0); e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e)
buf = (ctypes.c_ulonglong * (2 * 320 * 6))()
h.ganq_debug_solve_trace(buf)
t = np.frombuffer(buf, dtype=np.uint64).reshape(2, 320, 6).astype(np.float64)
SBW = int(os.environ.get("SB", 64)); nb = (n + SBW - 1) // SBW
P, G = t[0, :nb + 1], t[1, :nb + 1]
total = G[nb, 5] - P[0, 0]
tick = ms * 1e3 / total  # us per tick, assuming the traced workgroup spans the kernel
print(f"kernel {ms:.3f} ms; traced workgroup spans {total:.0f} ticks -> {tick * 1e3:.2f} ns per tick")
print("steps      | P: steps  store  wait1  wait2 | G: chain  stage  wait1  part2  wait2 | step total (us)")
for lo in range(0, nb + 1, 8):
    hi = min(nb + 1, lo + 8)
    p, g = P[lo:hi], G[lo:hi]
    f = lambda a: a.mean() * tick
    print(f"{lo:3d}..{hi - 1:3d}   | {f(p[:, 1] - p[:, 0]):8.2f} {f(p[:, 2] - p[:, 1]):6.2f} {f(p[:, 3] - p[:, 2]):6.2f} {f(p[:, 4] - p[:, 3]):6.2f} |"
          f" {f(g[:, 1] - g[:, 0]):8.2f} {f(g[:, 2] - g[:, 1]):6.2f} {f(g[:, 3] - g[:, 2]):6.2f} {f(g[:, 4] - g[:, 3]):6.2f} {f(g[:, 5] - g[:, 4]):6.2f} |"
          f" {f(g[:, 5] - g[:, 0]):6.2f}")
hw = [int(t[0, 300 + w, 0]) for w in range(16)]
if any(hw):
    print("HW_ID per wave (SIMD_ID = bits 5:4, CU_ID = bits 11:8):", [f"w{w}:simd{(v >> 4) & 3}/cu{(v >> 8) & 15}" for w, v in enumerate(hw) if v])

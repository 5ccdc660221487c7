#!/usr/bin/env python3
"""developer: staged-group Hessian -- row-major staging + ganq_hessian_accum (round-3 kernels) against transposed staging +
ganq_hessian_accum_t (hessian_w4.hip); device time per group of 8 x 2048 tokens, staging copies included and apart"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ganq_amd import _lib
def timed(fn, reps=6):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
rows, seq = 16384, 2048
_lib.debug_option("GANQ_HESS_W4", 2)  # (the product takes the transposed path from 3072 in_features on; here every size is timed)
for n in [int(a) for a in sys.argv[1:]] or [4096, 2048, 8192, 14336, 3072, 1024]:
    xs = [(torch.randn(seq, n, device="cuda") * 0.5).half() for _ in range(rows // seq)]
    H = torch.zeros(n, n, device="cuda")
    S = torch.empty(rows, n, dtype=torch.float16, device="cuda")
    St = torch.empty(n, rows, dtype=torch.float16, device="cuda")
    def stage_rm():
        for i, x in enumerate(xs): S[i * seq:(i + 1) * seq].copy_(x)
    def stage_t():
        for i, x in enumerate(xs): _lib.hessian_stage_t(St, x, i * seq)
    stage_rm(); stage_t()
    t_srm, t_st = timed(stage_rm), timed(stage_t)
    t_rm = timed(lambda: _lib.hessian_accum(H, S, 8, 8))
    t_t = timed(lambda: _lib.hessian_accum_t(H, St, rows, 8, 8))
    fl = 2.0 * rows * n * n
    print(f"n={n}: row-major staging {t_srm:.1f} + kernels {t_rm:.1f} = {t_srm + t_rm:.1f} us | transposed staging {t_st:.1f} + kernel {t_t:.1f} = {t_st + t_t:.1f} us "
          f"({(t_srm + t_rm) / (t_st + t_t):.2f} x; kernel alone {t_rm / t_t:.2f} x, {fl / t_t * 1e-9:.0f} TFLOP/s nominal)", flush=True)
    del xs, H, S, St

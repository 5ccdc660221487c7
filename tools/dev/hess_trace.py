"""developer: where a workgroup of the Hessian kernel spends a slab step (build with -DGANQ_HESS_TRACE)"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ganq_amd import _lib
h = ctypes.CDLL(_lib.LIB_PATH)
out = (ctypes.c_ulonglong * 8)()
for n, rows in ((4096, 16384), (8192, 16384), (4096, 2048)):
    X = (torch.randn(rows, n, device="cuda") * 0.5).half()
    H = torch.zeros(n, n, device="cuda")
    _lib.hessian_accum(H, X, 0, rows // 2048); torch.cuda.synchronize()
    h.ganq_debug_hess_trace(out)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); _lib.hessian_accum(H, X, 8, rows // 2048); e.record(); torch.cuda.synchronize()
    h.ganq_debug_hess_trace(out)
    nslab = rows // 32
    tot = sum(out[:4])
    print(f"n={n} rows={rows}: kernel {s.elapsed_time(e) * 1e3:.0f} us; workgroup 0: {tot} ticks over {nslab} slabs = {tot / nslab:.0f} per slab: "
          f"issue loads {out[0] / nslab:.0f}, frag reads + MFMAs {out[1] / nslab:.0f}, wait loads + LDS store {out[2] / nslab:.0f}, barrier {out[3] / nslab:.0f}")

"""developer: one launch per Hessian kernel variant, synchronised and checked after each (a cautious first run of new variants)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ganq_amd import _lib
_lib.selftest()
rows, n = int(sys.argv[1]), int(sys.argv[2])
torch.manual_seed(0)
X = (torch.randn(rows, n, device="cuda") * 0.5).half()
ref = (2.0 / 2.0) * (X.double().T @ X.double())
for wide, split in [(0, 0), (0, 1), (1, 0), (1, 1)]:
    _lib.debug_option("GANQ_HESS_WIDE", wide); _lib.debug_option("GANQ_HESS_SPLIT", split)
    H = torch.full((n, n), 7.0, device="cuda")
    print(f"wide={wide} split={split}: launching", flush=True)
    _lib.hessian_accum(H, X, 0, 2)
    torch.cuda.synchronize()
    err = float((H.double() - ref).norm() / ref.norm())
    print(f"wide={wide} split={split}: rel err {err:.2e} symmetric={torch.equal(H, H.T)}", flush=True)

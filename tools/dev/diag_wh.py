"""developer diagnostic: error of the W @ H_fixed product on a two-massive-feature Hessian, split by source"""
import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from ganq_amd import _lib
from test_hip_massive import hessian_with_scales, scale_cases, dev

m, n = 128, 1024
H, _ = hessian_with_scales(n, scale_cases(n, 5)["two_1000x_massive"], seed=5)
rng = np.random.default_rng(1)
for wkind in ("fp32", "fp16"):
    W = (0.02 * rng.standard_normal((m, n))).astype(np.float32)
    if wkind == "fp16":
        W = W.astype(np.float16).astype(np.float32)
    for f64 in (0, 1):
        _lib.debug_option("GANQ_WH_F64", f64)
        WH, Hf = _lib.debug_wh_product(dev(W), dev(H))
        WH, Hf = WH.cpu().numpy(), Hf.cpu().numpy()
        ref_true = W.astype(np.float64) @ H.astype(np.float64)
        ref_fix = W.astype(np.float64) @ Hf
        e_true = np.linalg.norm(WH - ref_true, axis=0) / np.linalg.norm(ref_true, axis=0)
        e_fix = np.linalg.norm(WH - ref_fix, axis=0) / np.linalg.norm(ref_fix, axis=0)
        e_h = np.linalg.norm(ref_fix - ref_true, axis=0) / np.linalg.norm(ref_true, axis=0)
        d = np.diag(H)
        top = np.argsort(e_fix)[-3:]
        print(wkind, "f64gemm" if f64 else "split", "vs true %.2e  vs W@Hfixed %.2e  fixed-vs-true %.2e" % (e_true.max(), e_fix.max(), e_h.max()),
              "worst cols", top, "H_uu", d[top], "max H_uu", d.max())
        dsq = np.sqrt(d.astype(np.float64))
        print("   Hfixed corr-relative err %.2e" % (np.abs(Hf - H.astype(np.float64)) / np.outer(dsq, dsq)).max())
_lib.debug_option("GANQ_WH_F64", None)

import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from ganq_amd import _lib
from test_hip_massive import hessian_with_scales, scale_cases, dev
n = 1024
H, _ = hessian_with_scales(n, scale_cases(n, 5)["two_1000x_massive"], seed=5)
W = np.ones((128, n), np.float32)
for ext in (-1, 0, 1):
    _lib.debug_option("GANQ_H_EXT", ext)
    WH, Hf = _lib.debug_wh_product(dev(W), dev(H))
    Hf = Hf.cpu().numpy()
    scale = float(np.abs(H).max()) / 2**30
    x = Hf / scale
    frac = x - np.rint(x)
    f16 = frac * 65536
    print("ext", ext, "scale", scale, "max|Hf-H|/scale", np.abs(Hf - H.astype(np.float64)).max() / scale,
          "frac nonzero share", float((np.abs(frac) > 1e-9).mean()), "frac*65536 integrality", float(np.abs(f16 - np.rint(f16)).max()))
    i, j = 5, 7
    print("   H", H[i, j], "Hf", Hf[i, j], "x", x[i, j], "true x", H[i, j] / scale)
_lib.debug_option("GANQ_H_EXT", None)

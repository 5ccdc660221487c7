"""developer diagnostic: where the codebook / loss error of the wide-range Hessian cases sits (per row, per entry)"""
import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from ganq_amd import _lib
from oracle import c_oracle
from test_hip_massive import hessian_with_scales, scale_cases, dev

m, n, V, K = 64, 1024, 16, 2
for case in ("log_uniform_1e3", "five_100x_outliers", "two_1000x_massive"):
    H, L = hessian_with_scales(n, scale_cases(n, 5)[case], seed=77, corr=0.05)
    rng = np.random.default_rng(9)
    W = (0.02 * rng.standard_normal((m, n))).astype(np.float16).astype(np.float32)
    T0 = c_oracle.kmeans_init(W, None, V)
    tr = c_oracle.run_layer_trace(W, H, L, T0, K)
    rec = _lib.run_layer_rows(dev(W), dev(H), dev(L), dev(T0), K, want_q_all=True)
    Tg = rec["T_all"].cpu().numpy(); lg = rec["loss_rows_all"].cpu().numpy()
    Q0 = tr["Q_all"][0]
    WHo = c_oracle.matmul(W, H)
    To, Ao, bo = c_oracle.update_t(WHo, H, Q0, V, want_ab=True)
    WHg, Hf = _lib.debug_wh_product(dev(W), dev(H))
    WHg = WHg.cpu().numpy()
    ref = W.astype(np.float64) @ H.astype(np.float64)
    print(case, "WH col err max", (np.linalg.norm(WHg - ref, axis=0) / np.linalg.norm(ref, axis=0)).max(),
          "entry max rel", (np.abs(WHg - ref) / (np.abs(ref) + 1e-30)).max())
    errT = np.linalg.norm(Tg[0] - tr["T_all"][0], axis=1) / np.linalg.norm(tr["T_all"][0], axis=1)
    worst = int(np.argmax(errT))
    print("  T row err: max %.3e median %.3e worst row %d" % (errT.max(), np.median(errT), worst))
    A = Ao[worst].astype(np.float64); lam = np.linalg.eigvalsh(0.5 * (A + A.T))
    print("  worst row eigen ratio", (lam / lam.max())[:4], "cut", 1.19e-7 * V, "counts", np.bincount(Q0[worst], minlength=V))
    print("  T gpu ", Tg[0][worst]); print("  T orc ", tr["T_all"][0][worst])
    el = np.abs(lg[0] - tr["loss_rows_all"][0]) / np.abs(tr["loss_rows_all"][0])
    print("  loss row rel err: max %.3e median %.3e" % (el.max(), np.median(el)))

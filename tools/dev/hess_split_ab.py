"""developer A/B: Hessian accumulation with the token-split launches (GANQ_HESS_SPLIT=1, default) vs whole tiles only (0)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ganq_amd import _lib
_lib.selftest()
caps = [int(a) for a in sys.argv[1:]] or [1]   # GANQ_HESS_SPLIT values; each is run with 128 x 128 tiles and with 256 x 128 tiles
for n in (768, 2048, 3072, 4096, 8192, 14336):
    for rows in (2048, 16384):
        torch.manual_seed(0)
        X = (torch.randn(rows, n, device="cuda") * 0.5).half()
        out = {}
        for split in [0] + caps + [-c for c in caps] + [-1000]:
            _lib.debug_option("GANQ_HESS_WIDE", 2 if split < 0 else 0)
            _lib.debug_option("GANQ_HESS_SPLIT", 0 if split == -1000 else abs(split))
            H = torch.zeros(n, n, device="cuda")
            ns = 0
            for _ in range(2): _lib.hessian_accum(H, X, ns, rows // 2048); ns += rows // 2048
            torch.cuda.synchronize()
            _lib.profile_enable(True)
            for _ in range(5): _lib.hessian_accum(H, X, ns, rows // 2048); ns += rows // 2048
            rep = _lib.profile_report(); _lib.profile_enable(False)
            ms, cnt = rep["hessian_kernel"]
            out[split] = (ms / cnt * 1e3, H)
        ref = out[0][1]
        msg = []
        for split in caps + [-c for c in caps] + [-1000]:
            err = ((out[split][1] - ref).abs().max() / ref.abs().max()).item()
            sym = torch.equal(out[split][1], out[split][1].T)
            msg.append(f"{'wide ' if split < 0 else ''}split={0 if split == -1000 else abs(split)}: {out[split][0]:.1f} us ({out[0][0] / out[split][0]:.2f}x, max rel diff {err:.1e}, symmetric={sym})")
        print(f"n={n} rows={rows}: whole tiles {out[0][0]:.1f} us; " + "; ".join(msg), flush=True)
_lib.debug_option("GANQ_HESS_SPLIT", None); _lib.debug_option("GANQ_HESS_WIDE", None)

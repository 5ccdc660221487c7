"""How many indices change between consecutive GANQ iterations (bench.py's synthetic layer; also a trained-like case)?"""
import argparse, os, sys, types, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from ganq_amd import _lib
from ganq_amd import distributed as gdist
a = argparse.Namespace(m=int(os.environ.get("M", 4096)), n=int(os.environ.get("N", 4096)), bits=int(os.environ.get("BITS", 4)),
                       iters=10, nseq=128, seqlen=int(os.environ.get("SEQLEN", 2048)))
dev = torch.device("cuda:0")
dist = gdist.Dist(0, 1, dev)
cap, setup = bench.build_workload(a, dist, dev)
W, H, L, T = cap["W"], cap["H"], cap["L"], cap["T0"].clone()
V = T.shape[1]
WH = _lib.matmul_f32(W, H)
Qp = None
for k in range(10):
    Q = _lib.solve_s(W, L, T)
    if Qp is not None:
        ch = (Q != Qp)
        per_row = ch.float().mean(dim=1)
        cnt = ch.sum(dim=1).float()
        qs = torch.quantile(cnt, torch.tensor([0.5, 0.9, 0.99, 0.999], device=cnt.device)).tolist()
        print(f"iter {k}: changed {ch.float().mean().item() * 100:6.2f}% of indices; rows: median {per_row.median().item() * 100:.2f}% max {per_row.max().item() * 100:.2f}%;"
              f" changes per row: p50 {qs[0]:.0f} p90 {qs[1]:.0f} p99 {qs[2]:.0f} p99.9 {qs[3]:.0f} max {cnt.max().item():.0f} (rows with > 64: {(cnt > 64).sum().item()})", flush=True)
    Qp = Q
    T = _lib.update_t(WH, H, Q, V)

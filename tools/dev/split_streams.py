#!/usr/bin/env python3
"""Experiment: the benchmark layer's loop as P row parts on P streams (ganq_run_layer_rows per part + ganq_select_best), so the
T-update kernels of one part run beside the S-solve of the others.  Prints ms per layer for P = 1, 2, 4 and checks that the
result equals the single-launch loop's bit for bit."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from ganq_amd import _lib  # noqa: E402
from ganq_amd import distributed as gdist  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=4096)
    ap.add_argument("--n", type=int, default=4096)
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    args = argparse.Namespace(m=a.m, n=a.n, bits=4, iters=10, nseq=32, seqlen=2048, mode="layers")
    dist = gdist.init_from_env()
    cap, _ = bench.build_workload(args, dist, dist.device)
    W, H, L, T0 = cap["W"], cap["H"], cap["L"], cap["T0"]
    K = 10
    ws = _lib.run_layer_workspace(a.m, a.n, 16, W.device)
    T1, Q1, d1, b1 = _lib.run_layer(W, H, L, T0, K, workspace=ws)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.reps):
        _lib.run_layer(W, H, L, T0, K, workspace=ws)
    torch.cuda.synchronize()
    print(f"single launch sequence: {(time.perf_counter() - t0) / a.reps * 1e3:.3f} ms")
    for P in (1, 2, 4, 8):
        sl = gdist.row_slices(a.m, P, align=128)
        streams = [torch.cuda.Stream() for _ in range(P)]
        wss = [_lib.run_layer_workspace(hi - lo, a.n, 16, W.device) for lo, hi in sl]
        Ws = [W[lo:hi].contiguous() for lo, hi in sl]
        Ts = [T0[lo:hi].contiguous() for lo, hi in sl]

        def run():
            main_s = torch.cuda.current_stream()
            recs = []
            for p in range(P):
                streams[p].wait_stream(main_s)
                with torch.cuda.stream(streams[p]):
                    recs.append(_lib.run_layer_rows(Ws[p], H, L, Ts[p], K, alias_q=True, workspace=wss[p]))
            for p in range(P):
                main_s.wait_stream(streams[p])
            loss = torch.cat([r["loss_rows_all"] for r in recs], dim=1)
            dists, bk = _lib.select_best(loss)
            return recs, dists, bk

        recs, dists, bk = run()
        torch.cuda.synchronize()
        ok = torch.equal(dists, d1) and int(bk) == int(b1) and torch.equal(torch.cat([r["Q_last"] for r in recs]), Q1)
        t0 = time.perf_counter()
        for _ in range(a.reps):
            run()
        torch.cuda.synchronize()
        print(f"P = {P}: {(time.perf_counter() - t0) / a.reps * 1e3:.3f} ms  identical={ok}")


if __name__ == "__main__":
    main()

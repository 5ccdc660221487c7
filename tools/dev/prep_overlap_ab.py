#!/usr/bin/env python3
"""A/B: the T-update's preparation beside the first S-solve (helper stream, GANQ_PREP_OVERLAP=1) against the single-stream
sequence; ms per layer of the fused loop on the benchmark workload and other shapes, results compared bit for bit."""
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
    dist = gdist.init_from_env()
    for (m, n) in [(4096, 4096), (1024, 4096), (8192, 2048), (768, 3072)]:
        args = argparse.Namespace(m=m, n=n, bits=4, iters=10, nseq=16, seqlen=2048, mode="layers")
        cap, _ = bench.build_workload(args, dist, dist.device)
        W, H, L, T0 = cap["W"], cap["H"], cap["L"], cap["T0"]
        ws = _lib.run_layer_workspace(m, n, 16, W.device)
        res = {}
        for ov in (0, 1, 0, 1):
            _lib.debug_option("GANQ_PREP_OVERLAP", ov)
            out = _lib.run_layer(W, H, L, T0, 10, workspace=ws)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                _lib.run_layer(W, H, L, T0, 10, workspace=ws)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 5 * 1e3
            same = True
            if ov in res:
                same = all(torch.equal(a, b) for a, b in zip(out, res[ov][1]))
            elif res:
                ref = next(iter(res.values()))[1]
                same = all(torch.equal(a, b) for a, b in zip(out, ref))
            res[ov] = (ms, out)
            print(f"{m}x{n} overlap={ov}: {ms:.3f} ms  identical={same}", flush=True)
        _lib.debug_option("GANQ_PREP_OVERLAP", None)
        del cap, ws
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""GanqHipQuantLinear forward vs torch fp16 F.linear (BASELINE.json configs[2]): decode (M = 1..16) and prefill shapes."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ganq_amd import _lib  # noqa: E402


def timeit(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3  # us


def main():
    out = []
    for (m, n) in [(4096, 4096), (8192, 2048), (2048, 8192), (14336, 4096)]:
        for bits in (4, 3):
            V = 2 ** bits
            g = torch.Generator(device="cuda").manual_seed(0)
            Q = torch.randint(0, V, (m, n), device="cuda", generator=g, dtype=torch.uint8)
            lut = (0.02 * torch.randn(m, V, device="cuda", generator=g)).half()
            qw = _lib.pack_indices(Q, bits)
            Wd = _lib.lut_dequant(qw, lut, n, bits)
            for M in (1, 4, 16):
                x = torch.randn(M, n, device="cuda", generator=g).half()
                t_lut = timeit(lambda: _lib.lut_linear(x, qw, lut, None, bits))
                t_fp = timeit(lambda: torch.nn.functional.linear(x, Wd))
                gbs = (m * n * bits / 8 + m * V * 2) / (t_lut * 1e-6) / 1e9
                out.append(dict(m=m, n=n, bits=bits, M=M, lut_us=round(t_lut, 2), fp16_us=round(t_fp, 2),
                                speedup=round(t_fp / t_lut, 2), lut_GBs=round(gbs, 1)))
                print(out[-1], flush=True)
            if bits == 4:
                x = torch.randn(2048, n, device="cuda", generator=g).half()
                t_deq = timeit(lambda: _lib.lut_dequant(qw, lut, n, bits), reps=20)
                t_fp = timeit(lambda: torch.nn.functional.linear(x, Wd), reps=20)
                out.append(dict(m=m, n=n, bits=bits, M=2048, dequant_us=round(t_deq, 2), fp16_gemm_us=round(t_fp, 2)))
                print(out[-1], flush=True)
    json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "lut_linear_bench.json"), "w"), indent=1)


if __name__ == "__main__":
    main()

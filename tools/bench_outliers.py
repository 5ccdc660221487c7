#!/usr/bin/env python3
"""Outlier split (ganq_outlier_ratio): time of the split kernels at 4096x4096 and of the LUT forward with and without the
sparse addend, vs torch fp16 F.linear."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ganq_amd import _lib  # noqa: E402
from bench_lut_linear import timeit  # noqa: E402

m = n = 4096
g = torch.Generator(device="cuda").manual_seed(0)
W = (0.02 * torch.randn(m, n, device="cuda", generator=g)).half().float()
for ratio in (0.005, 0.0045):
    t = timeit(lambda: _lib.outlier_split(W.clone(), ratio), reps=10)
    t0 = timeit(lambda: W.clone(), reps=10)
    rowptr, cols, vals, _ = _lib.outlier_split(W.clone(), ratio)
    print(f"ratio {ratio}: split {t - t0:.1f} us (incl. the host read of nnz), nnz {cols.numel()} = {cols.numel() / (m * n):.4%}")
bits, V = 4, 16
Q = torch.randint(0, V, (m, n), device="cuda", generator=g, dtype=torch.uint8)
lut = (0.02 * torch.randn(m, V, device="cuda", generator=g)).half()
qw = _lib.pack_indices(Q, bits)
Wd = _lib.lut_dequant(qw, lut, n, bits)
vh = vals.half()
for M in (1, 4, 16, 64):
    x = torch.randn(M, n, device="cuda", generator=g).half()
    t_plain = timeit(lambda: _lib.lut_linear(x, qw, lut, None, bits))
    t_sp = timeit(lambda: _lib.outlier_matmul(x, rowptr, cols, vh, m))
    t_both = timeit(lambda: _lib.lut_linear(x, qw, lut, None, bits, addend=_lib.outlier_matmul(x, rowptr, cols, vh, m)))
    t_one = timeit(lambda: _lib.lut_linear_outliers(x, qw, lut, None, bits, rowptr, cols, vh))
    t_fp = timeit(lambda: torch.nn.functional.linear(x, Wd))
    print(f"M={M}: LUT {t_plain:.1f} us, sparse product {t_sp:.1f} us, LUT + outliers {t_both:.1f} us (one call: {t_one:.1f} us), fp16 F.linear {t_fp:.1f} us", flush=True)

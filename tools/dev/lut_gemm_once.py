#!/usr/bin/env python3
"""developer: a few launches of the LUT GEMM at one shape (for rocprofv3 --pmc / --kernel-trace runs)
usage: lut_gemm_once.py m n M [pipe]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ganq_amd import _lib  # noqa: E402

m, n, M = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
if len(sys.argv) > 4:
    _lib.debug_option("GANQ_LUT_GEMM_PIPE", int(sys.argv[4]))
g = torch.Generator(device="cuda").manual_seed(0)
Q = torch.randint(0, 16, (m, n), device="cuda", generator=g, dtype=torch.int32).to(torch.uint8)
lut = (0.02 * torch.randn(m, 16, device="cuda", generator=g)).half()
x = torch.randn(M, n, device="cuda", generator=g).half()
qw = _lib.pack_indices(Q, 4)
for _ in range(5):
    y = _lib.lut_linear(x, qw, lut, None, 4)
torch.cuda.synchronize()
print(float(y.float().abs().mean()))

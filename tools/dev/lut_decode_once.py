#!/usr/bin/env python3
"""developer: a ring of cold LUT decode calls at one shape (target of counter passes: rocprofv3 --pmc FETCH_SIZE / tools/dev/sq_pmc.sh)
usage: lut_decode_once.py m n M"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ganq_amd import _lib
m, n, M = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (4096, 14336, 1)))
g = torch.Generator(device="cuda").manual_seed(0)
nl = max(2, int(600e6 // (m * n // 2)))
x = torch.randn(M, n, device="cuda", generator=g).half()
Q = torch.randint(0, 16, (m, n), device="cuda", generator=g, dtype=torch.uint8)
lut = (0.02 * torch.randn(m, 16, device="cuda", generator=g)).half()
qw0 = _lib.pack_indices(Q, 4)
qws = [qw0.clone() for _ in range(nl)]
for rep in range(2):
    for i in range(nl):
        y = _lib.lut_linear(x, qws[i], lut, None, 4)
torch.cuda.synchronize()
print("layers in the ring", nl, "packed bytes per layer", m * n // 2)

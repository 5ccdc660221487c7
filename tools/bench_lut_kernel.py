import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ganq_amd import _lib
for (m, n) in [(4096, 4096), (8192, 2048), (2048, 8192), (14336, 4096)]:
    for bits in (4, 3):
        V = 2 ** bits
        g = torch.Generator(device="cuda").manual_seed(0)
        Q = torch.randint(0, V, (m, n), device="cuda", generator=g, dtype=torch.uint8)
        lut = (0.02 * torch.randn(m, V, device="cuda", generator=g)).half()
        qw = _lib.pack_indices(Q, bits)
        for M in (1, 16, 64):
            x = torch.randn(M, n, device="cuda", generator=g).half()
            for _ in range(3): _lib.lut_linear(x, qw, lut, None, bits)
            _lib.profile_enable(True)
            for _ in range(20): _lib.lut_linear(x, qw, lut, None, bits)
            rep = _lib.profile_report(); _lib.profile_enable(False)
            ms, cnt = rep["lut_gemv_kernel"]
            us = ms / cnt * 1e3
            print(f"m={m} n={n} bits={bits} M={M}: device {us:.2f} us  -> {(m*n*bits/8)/(us*1e-6)/1e9:.0f} GB/s of packed weights")

#!/usr/bin/env python3
"""developer: where the four-wave GEMM's cycles go (GANQ_HIP_LIB=build_variants/libganq_probe.so, built by tools/dev/gemm_probe.sh)"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ganq_amd import _lib
M, N, K = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (4096, 4096, 4096)))
x = torch.randn(M, K, device="cuda").half(); w = (0.05 * torch.randn(N, K, device="cuda")).half()
_lib.debug_option("GANQ_GEMM_H16_BM", 512)
for _ in range(5):
    y = _lib.debug_gemm_h16(x, w)
torch.cuda.synchronize()
h = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * 32)()
assert h.ganq_debug_gemm_probe(buf) == 0
names = ["k half 0 (64 mfma + 16 reads + 16 dma)", "wait lds", "k half 1 a (32 mfma)", "wait dma", "barrier", "k half 1 b (32 mfma + 16 reads)", "wait lds"]
nt = K // 64
for wv in range(4):
    v = [int(buf[wv * 8 + k]) for k in range(8)]
    print(f"wave {wv}: K loop {v[7]} cycles = {v[7] / nt:.0f} per K tile (128 matrix instructions = 2048 pipe cycles)")
    for k in range(7):
        print(f"    {names[k]:45s} {v[k] / nt:8.1f} cycles per K tile")

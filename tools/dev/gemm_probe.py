#!/usr/bin/env python3
"""developer: where the four-wave GEMM's cycles go (GANQ_HIP_LIB=build_variants/libganq_probe.so, built by tools/dev/probe_build.sh gemm_h16 HG_PROBE)"""
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
names = ["wait dma (vmcnt 16)", "barrier", "64 mfma + 16 reads + 8 dma", "wait lds"]
nt = K // 32
for wv in range(4):
    v = [int(buf[wv * 8 + k]) for k in range(8)]
    print(f"wave {wv}: K loop {v[7]} cycles = {v[7] / nt:.0f} per slice (64 matrix instructions = 1024 pipe cycles)")
    print(f"    100 MHz clock: K loop {v[4] / 100:.1f} us ({v[7] / (v[4] * 10.0):.2f} GHz), K loop + epilogue {v[5] / 100:.1f} us")
    for k in range(4):
        print(f"    {names[k]:45s} {v[k] / nt:8.1f} cycles per slice")

#!/usr/bin/env python3
"""developer: where the stream-K Hessian kernel's time goes (GANQ_HIP_LIB=build_variants/libganq_probe.so, built by
tools/dev/probe_build.sh hessian_w4 HW_PROBE): per workgroup, slice loops against segment ends, on the 100 MHz clock"""
import ctypes, os, sys, torch
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ganq_amd import _lib
rows, n = (int(v) for v in (sys.argv[1:3] if len(sys.argv) > 2 else (16384, 4096)))
Xt = (torch.randn(n, rows, device="cuda") * 0.5).half()
H = torch.zeros(n, n, device="cuda")
_lib.hessian_accum_t(H, Xt, rows, 0, 8)
torch.cuda.synchronize()
h = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * 8)(); allb = (ctypes.c_ulonglong * 4096)()
assert h.ganq_debug_hess_w4_probe(buf, 1) == 0 and h.ganq_debug_hess_w4_probe_all(allb) == 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); _lib.hessian_accum_t(H, Xt, rows, 8, 8); e1.record(); torch.cuda.synchronize()
assert h.ganq_debug_hess_w4_probe(buf, 1) == 0 and h.ganq_debug_hess_w4_probe_all(allb) == 0
v = [int(buf[k]) for k in range(8)]
a = np.array(list(allb), dtype=np.float64).reshape(1024, 4)
a = a[a[:, 0] > 0]
t0 = a[:, 0].min()
print(f"n={n} rows={rows}: {e0.elapsed_time(e1) * 1e3:.1f} us per call; workgroup 0: {v[1]} slices, {v[0] / max(v[1], 1):.0f} cycles per slice (1024 matrix cycles), "
      f"{v[0] / max(v[3], 1) / 10:.2f} GHz; per slice: wait dma {v[5] / max(v[1], 1):.0f}, barrier {v[6] / max(v[1], 1):.0f}, wait lds {v[7] / max(v[1], 1):.0f} cycles")
print(f"  {len(a)} workgroups: first slice loop starts {(a[:, 0] - t0).min() / 100:.1f} .. {(a[:, 0] - t0).max() / 100:.1f} us; slice loops {a[:, 1].min() / 100:.1f} / "
      f"{a[:, 1].mean() / 100:.1f} / {a[:, 1].max() / 100:.1f} us (min / mean / max); segment ends {a[:, 2].min() / 100:.1f} / {a[:, 2].mean() / 100:.1f} / {a[:, 2].max() / 100:.1f} us; "
      f"last workgroup done {(a[:, 3] - t0).max() / 100:.1f} us after the first started")

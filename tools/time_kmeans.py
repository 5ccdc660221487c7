"""Device time of ganq_kmeans_init (HIP events inside the library) for a few shapes."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ganq_amd import _lib
shapes = [(4096, 4096, 16), (4096, 4096, 8), (4096, 11008, 16), (1024, 2048, 16)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
for m, n, V in shapes:
    g = torch.Generator(device="cuda").manual_seed(0)
    W = 0.02 * torch.randn(m, n, device="cuda", generator=g)
    cw = (torch.rand(n, device="cuda", generator=g, dtype=torch.float64) + 0.5) ** 4
    _lib.kmeans_init(W, cw, V)
    torch.cuda.synchronize()
    _lib.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(3):
        T0 = _lib.kmeans_init(W, cw, V)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 3
    rep = _lib.profile_report(); _lib.profile_enable(False)
    ms, cnt = rep["kmeans_kernels"]
    print(f"m={m} n={n} V={V}: device {ms / cnt:.2f} ms  wall {wall * 1e3:.2f} ms  checksum {float(T0.double().sum()):.9e}", flush=True)

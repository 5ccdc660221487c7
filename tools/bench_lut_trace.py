"""One LUT-linear configuration run N times next to torch fp16 F.linear, for `rocprofv3 --kernel-trace --stats`.
usage: python3 tools/bench_lut_trace.py m n bits M [iters]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ganq_amd import _lib
m, n, bits, M = (int(a) for a in sys.argv[1:5])
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 200
V = 2 ** bits
g = torch.Generator(device="cuda").manual_seed(0)
Q = torch.randint(0, V, (m, n), device="cuda", generator=g, dtype=torch.uint8)
lut = (0.02 * torch.randn(m, V, device="cuda", generator=g)).half()
qw = _lib.pack_indices(Q, bits)
x = torch.randn(M, n, device="cuda", generator=g).half()
W = torch.gather(lut, 1, Q.long())
for _ in range(iters):
    _lib.lut_linear(x, qw, lut, None, bits)
torch.cuda.synchronize()
for _ in range(iters):
    torch.nn.functional.linear(x, W)
torch.cuda.synchronize()

"""Warm wall time of a full GANQ.quantize() (prologue + k-means + K iterations + epilogue) and of its phases.
usage: python tools/time_quantize.py [m n]"""
import os, sys, time
import torch, torch.nn as nn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ganq_amd.quantization import GANQ, QuantizeConfig
from ganq_amd.looper.named_module import NamedModule
from ganq_amd import _lib

m, n = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4096, 4096)
dev = "cuda"
torch.manual_seed(0)
lin = nn.Linear(n, m, bias=False).half().to(dev)
lin.weight.data = (0.02 * torch.randn(m, n, device=dev)).half()
qcfg = QuantizeConfig(bits=4, act_sort="asc", l_damp_style="ganq", dead="mean", damp_percent=0.01, desc_act=True,
                      group_size=128, ganq_iterations=10)
scale = 0.1 + torch.rand(n, device=dev)
xs = [(torch.randn(2048, n, device=dev) * scale).half() for _ in range(8)]

marks = {}
def timed(name, fn):
    def w(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = fn(*a, **k)
        torch.cuda.synchronize(); marks[name] = marks.get(name, 0.0) + time.perf_counter() - t0
        return r
    return w

for rep in range(4):
    q = GANQ(NamedModule(lin, "proj", "model.layers.0.proj", 0), qcfg)
    q.quantizer.configure(perchannel=True)
    for x in xs:
        q.add_batch(x.unsqueeze(0), None)
    if rep == 3:  # instrumented pass (adds syncs)
        q._perform_quantization_loop = timed("loop_total", q._perform_quantization_loop)
        q._initialize_codebook_kmeans = timed("kmeans", q._initialize_codebook_kmeans)
        q._hip_cholesky = timed("cholesky", q._hip_cholesky)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = q.quantize()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"rep {rep}: quantize() {dt*1e3:.2f} ms  avg_loss {out[5]:.4f}", flush=True)
print({k: round(v * 1e3, 2) for k, v in marks.items()})

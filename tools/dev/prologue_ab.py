#!/usr/bin/env python3
"""developer: warm GANQ.quantize() with the prologue as HIP passes ("hip") against the reference's op sequence on torch
("torch"), phase by phase: python tools/dev/prologue_ab.py [n ...]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch.nn as nn
from ganq_amd import _lib
from ganq_amd.looper.named_module import NamedModule
from ganq_amd.quantization import GANQ, QuantizeConfig

for n in [int(a) for a in sys.argv[1:]] or [4096, 2048, 8192]:
    m = 256
    lin = nn.Linear(n, m, bias=False).half().cuda()
    X = (torch.randn(4096, n, device="cuda") * (0.1 + torch.rand(n, device="cuda"))).half()
    for mode in ("hip", "torch", "hip", "torch"):
        qcfg = QuantizeConfig(bits=4, act_sort="asc", l_damp_style="ganq", dead="mean", ganq_iterations=1, ganq_prologue=mode)
        q = GANQ(NamedModule(lin, "fc", "layers.0.fc", 0), qcfg)
        q.quantizer.configure(perchannel=True)
        q.add_batch(X.unsqueeze(0), None)
        H = q.hessian
        W = q.module_copy.clone()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if mode == "hip":
            out = q._prologue_hip(W, H.clone())
        else:
            out = q._prologue_reference_ops(W, H.clone())
        torch.cuda.synchronize()
        print(f"n={n} prologue={mode}: {(time.perf_counter() - t0) * 1e3:.2f} ms (incl. one 4n^2-byte clone of H)", flush=True)
        q.free()

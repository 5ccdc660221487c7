#!/usr/bin/env python3
"""Fused LUT-dequant GEMM (csrc/lut_gemm.hip) against the library fp16 GEMM on the dequantised weight (torch F.linear ->
hipBLASLt) and against dequant + GEMM, the path it replaces.  Device time per call from HIP-graph replays.

    python tools/bench_lut_gemm.py [--shapes 4096x4096,14336x4096] [--M 128,512,2048,4096]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ganq_amd import _lib  # noqa: E402


def graph_time(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                fn()
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3  # us


def bench(m, n, M, bits=4, dtype=torch.float16):
    g = torch.Generator(device="cuda").manual_seed(0)
    Q = torch.randint(0, 2 ** bits, (m, n), device="cuda", generator=g, dtype=torch.int32).to(torch.uint8)
    lut = (0.02 * torch.randn(m, 2 ** bits, device="cuda", generator=g)).to(dtype)
    x = torch.randn(M, n, device="cuda", generator=g).to(dtype)
    qw = _lib.pack_indices(Q, bits)
    Wq = _lib.lut_dequant(qw, lut, n, bits)
    t_lut = graph_time(lambda: _lib.lut_linear(x, qw, lut, None, bits))
    t_v = {}
    for pipe in (0, 1):  # the two whole-K kernels, forced (split-K launches ignore the switch)
        _lib.debug_option("GANQ_LUT_GEMM_PIPE", pipe)
        t_v[pipe] = graph_time(lambda: _lib.lut_linear(x, qw, lut, None, bits))
    _lib.debug_option("GANQ_LUT_GEMM_PIPE", None)
    t_lib = graph_time(lambda: torch.nn.functional.linear(x, Wq))
    t_deq = graph_time(lambda: torch.nn.functional.linear(x, _lib.lut_dequant(qw, lut, n, bits)))
    flop = 2.0 * M * m * n
    return {"out_x_in": f"{m}x{n}", "M": M, "bits": bits, "lut_gemm_us": round(t_lut, 1), "two_per_cu_us": round(t_v[0], 1), "pipelined_us": round(t_v[1], 1), "lib_fp16_gemm_us": round(t_lib, 1),
            "dequant_plus_lib_us": round(t_deq, 1), "lut_gemm_TFLOPs": round(flop / t_lut / 1e6, 1),
            "lib_TFLOPs": round(flop / t_lib / 1e6, 1), "vs_lib": round(t_lib / t_lut, 3), "vs_dequant_plus_lib": round(t_deq / t_lut, 3)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="4096x4096,14336x4096,4096x14336")
    ap.add_argument("--M", default="128,512,2048,4096")
    ap.add_argument("--bits", type=int, default=4)
    a = ap.parse_args()
    for sh in a.shapes.split(","):
        m, n = (int(v) for v in sh.split("x"))
        for M in (int(v) for v in a.M.split(",")):
            print(json.dumps(bench(m, n, M, a.bits)), flush=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""csrc/gemm_h16.hip against the library fp16 GEMM (torch F.linear -> hipBLASLt), same random operands, device time from HIP-graph
replays:  python tools/bench_gemm_h16.py [--shapes 4096x4096,...] [--M 1024,2048,4096]"""
import argparse, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ganq_amd import _lib
from bench_lut_gemm import graph_time


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="4096x4096,14336x4096,4096x14336")
    ap.add_argument("--M", default="1024,2048,4096")
    ap.add_argument("--bm", default="0,128,256")
    a = ap.parse_args()
    for sh in a.shapes.split(","):
        N, K = (int(v) for v in sh.split("x"))
        for M in (int(v) for v in a.M.split(",")):
            g = torch.Generator(device="cuda").manual_seed(0)
            x = torch.randn(M, K, device="cuda", generator=g).half()
            w = (0.02 * torch.randn(N, K, device="cuda", generator=g)).half()
            row = {"out_x_in": sh, "M": M}
            t_lib = graph_time(lambda: torch.nn.functional.linear(x, w))
            row["lib_us"] = round(t_lib, 1)
            for bm in (int(v) for v in a.bm.split(",")):
                _lib.debug_option("GANQ_GEMM_H16_BM", bm if bm else None)
                t = graph_time(lambda: _lib.debug_gemm_h16(x, w))
                row[f"h16_bm{bm}_us"] = round(t, 1)
                row[f"h16_bm{bm}_TF"] = round(2.0 * M * N * K / t / 1e6, 1)
            _lib.debug_option("GANQ_GEMM_H16_BM", None)
            row["lib_TF"] = round(2.0 * M * N * K / t_lib / 1e6, 1)
            row["best_vs_lib"] = round(t_lib / min(row[k] for k in row if k.endswith("_us") and k.startswith("h16")), 3)
            print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()

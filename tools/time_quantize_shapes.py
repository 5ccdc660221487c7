"""Warm quantize() phase times for the module shapes of a Llama-3.2-1B / Llama-3-8B layer.
usage: python tools/time_quantize_shapes.py"""
import os, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
for m, n in [(2048, 2048), (512, 2048), (8192, 2048), (2048, 8192), (4096, 4096), (14336, 4096), (4096, 14336)]:
    out = subprocess.run([sys.executable, os.path.join(here, "time_quantize.py"), str(m), str(n)], capture_output=True, text=True).stdout
    lines = [l for l in out.splitlines() if l.startswith("rep 2") or l.startswith("{")]
    print(f"{m}x{n}: " + " | ".join(lines), flush=True)

import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import bench_lut_decode as bl
from ganq_amd import _lib
_lib.selftest()
for (m, n) in [(512, 2048), (1024, 4096), (768, 768), (1024, 2048)]:
    print("default", bl.bench(m, n, 4, 1, True))
    for ks in (1, 2, 4):
        _lib.debug_option("GANQ_LUT_INWG", 1); _lib.debug_option("GANQ_LUT_KS", ks)
        r = bl.bench(m, n, 4, 1, True)
        print("decode kernel ks", ks, r["lut_us"], r["torch_fp16_us"])
        _lib.debug_option("GANQ_LUT_INWG", None); _lib.debug_option("GANQ_LUT_KS", None)

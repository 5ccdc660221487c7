"""LUT decode forward vs torch fp16 F.linear, hot (one layer re-used: weights stay in the 256 MB Infinity Cache) and
cold (a ring of layers larger than the cache, as in a real decode step).  Device time per call from back-to-back
launches between two events (launch gaps included for both sides alike).
usage: python tools/bench_lut_decode.py [--json]"""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ganq_amd import _lib


def timed(fn, n_layers, iters):
    """device time per call: the calls are captured into one HIP graph (no host launch cost in the timed region) and the
    graph is replayed between two events"""
    for i in range(n_layers):
        fn(i)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        fn(0)
        torch.cuda.synchronize()
        with torch.cuda.graph(graph, stream=side):
            for it in range(iters):
                fn(it % n_layers)
    torch.cuda.synchronize()
    graph.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(3):
        graph.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / (3 * iters) * 1e3  # us


def launch_floor():
    """what ANY kernel costs in this harness: a 256-thread fill of 1 KB, 200 dependent launches in the same HIP graph -- the time from
    one kernel's start to the next one's start when nothing is computed (dispatch, completion signal, the barrier between the nodes)"""
    t = torch.zeros(256, device="cuda")
    return round(timed(lambda i: t.fill_(1.0), 1, 200), 2)


def bench(m, n, bits, M, cold, outliers=0.0):
    V = 2 ** bits
    g = torch.Generator(device="cuda").manual_seed(0)
    nl = max(1, int(600e6 // (m * n * 2))) if cold else 1  # ring: > 512 MB of fp16 weights (and > 256 MB... of packed ones x4)
    nl_q = max(1, int(600e6 // (m * n * bits // 8))) if cold else 1
    x = torch.randn(M, n, device="cuda", generator=g).half()
    Q = torch.randint(0, V, (m, n), device="cuda", generator=g, dtype=torch.uint8)
    lut = (0.02 * torch.randn(m, V, device="cuda", generator=g)).half()
    qw0 = _lib.pack_indices(Q, bits)
    qws = [qw0.clone() for _ in range(nl_q)]
    luts = [lut.clone() for _ in range(nl_q)]
    W0 = torch.gather(lut, 1, Q.long())
    Ws = [W0.clone() for _ in range(nl)]
    del Q
    csr = None
    if outliers > 0:
        k = max(1, int(n * outliers))
        rowptr = (torch.arange(m + 1, device="cuda") * k).to(torch.int32)
        cols = torch.stack([torch.randperm(n, device="cuda")[:k].sort().values for _ in range(8)]).repeat((m + 7) // 8, 1)[:m].reshape(-1).to(torch.int32)
        vals = (0.1 * torch.randn(m * k, device="cuda")).half()
        csr = (rowptr, cols.contiguous(), vals)
    if csr is None:
        t_lut = timed(lambda i: _lib.lut_linear(x, qws[i], luts[i], None, bits), nl_q, 200)
    else:
        t_lut = timed(lambda i: _lib.lut_linear_outliers(x, qws[i], luts[i], None, bits, *csr), nl_q, 200)
    t_f16 = timed(lambda i: torch.nn.functional.linear(x, Ws[i]), nl, 200)
    nbytes = m * n * bits / 8
    return {"m": m, "n": n, "bits": bits, "M": M, "cold": cold, "outlier_ratio": outliers, "lut_us": round(t_lut, 2),
            "torch_fp16_us": round(t_f16, 2), "speedup": round(t_f16 / t_lut, 2), "lut_GBs": round(nbytes / t_lut / 1e3, 1),
            "frac_of_8TBs": round(nbytes / t_lut / 1e3 / 8000, 4)}


if __name__ == "__main__":
    _lib.selftest()
    if "--nt" in sys.argv:  # developer: force 16 / 32 features per workgroup and time the key shapes only
        for nt in (1, 2):
            _lib.debug_option("GANQ_LUT_NT", nt)
            for (m, n, M) in [(4096, 4096, 1), (4096, 4096, 16), (14336, 4096, 1), (14336, 4096, 16), (2048, 8192, 16)]:
                print("nt", nt, bench(m, n, 4, M, True))
        sys.exit(0)
    if "--ks" in sys.argv:  # developer: cross-workgroup split factor of the decode kernel on the layers with few feature blocks
        for ks in (1, 2, 4, 8):
            _lib.debug_option("GANQ_LUT_KS", ks)
            for (m, n, M) in [(2048, 8192, 1), (2048, 8192, 16), (2048, 2048, 1), (2048, 2048, 16), (1024, 4096, 1), (512, 2048, 1)]:
                print("ks", ks, bench(m, n, 4, M, True))
        sys.exit(0)
    rows = []
    print({"launch_floor_us": launch_floor()})
    for (m, n) in [(4096, 4096), (14336, 4096), (4096, 14336), (2048, 2048), (8192, 2048), (2048, 8192)]:
        for M in (1, 16):
            for cold in (False, True):
                rows.append(bench(m, n, 4, M, cold))
    rows.append(bench(4096, 4096, 3, 1, True))
    rows.append(bench(4096, 4096, 4, 1, True, outliers=0.005))
    rows.append(bench(4096, 4096, 4, 1, False, outliers=0.005))
    if "--json" in sys.argv:
        print(json.dumps(rows))
    else:
        for r in rows:
            print(r)

"""CPU-side checks of the C-ABI boundary: the shared library builds for gfx950, loads, and exports every
symbol include/ganq_hip.h declares (no compute calls without a GPU)."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "ganq_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ganq_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    from ganq_amd import _lib

    assert declared_symbols() == sorted(_lib.SIGNATURES)


def test_library_builds_loads_and_exports_everything():
    from ganq_amd import _lib

    _lib.build()
    handle = _lib.lib()
    for name in declared_symbols():
        assert hasattr(handle, name), name
    assert handle.ganq_hip_version() == 4
    # size queries are pure host functions
    # Err scratch + packed L, plus what the helper workgroups of small launches need (accumulators and their own Err copy for at
    # most 128 tiles and two helpers each, flags): 64 MiB + 64 MiB + 2 x (32.5 MiB + 32 MiB) + 5 KiB at 4096 x 4096
    need = handle.ganq_solve_s_workspace_bytes(4096, 4096, 16)
    assert 2 * 4096 * 4096 * 4 <= need <= 2 * 4096 * 4096 * 4 + 4 * 128 * 65 * 4096 + 8192
    assert handle.ganq_run_layer_workspace_bytes(4096, 4096, 16) > handle.ganq_update_t_workspace_bytes(4096, 4096, 16)


def test_no_cpu_fallback():
    import torch
    from ganq_amd import _lib

    if torch.cuda.is_available():
        pytest.skip("CPU-only check")
    W = torch.zeros(16, 64)
    with pytest.raises(_lib.GanqHipError):
        _lib.solve_s(W, torch.eye(64), torch.zeros(16, 16))


def test_product_does_not_import_oracle():
    """the oracle is test infrastructure: nothing under ganq_amd/ may import, link or call it (comments may cite it)"""
    pkg = os.path.join(ROOT, "ganq_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            path = os.path.join(dirpath, f)
            if f.endswith(".py"):
                src = open(path).read()
                assert not re.search(r"^\s*(import|from)\s+oracle\b", src, flags=re.M), f
                assert "libganq_oracle" not in src, f
            elif f.endswith((".hip", ".h", ".cc", ".cpp")) or f == "Makefile":
                src = open(path).read()
                src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
                src = re.sub(r"//[^\n]*", "", src)
                assert "ganq_oracle" not in src and "oracle/" not in src, f


@pytest.mark.parametrize("ncu", [8, 31, 64, 104, 256, 304])
def test_hessian_stream_k_cut_covers_every_slice_once(ncu):
    """The cut of ganq_hessian_accum_t (hessian_w4.hip), enumerated on the host BY THE CODE THE KERNELS RUN (ganq_debug_hessian_t_cut;
    no GPU): for every layer width / group length / workgroup count -- CU counts of other parts and partition modes included --
    each (tile, token slice) pair belongs to exactly one segment, a workgroup leaves at most three partial tiles in distinct slots,
    every segment of a tile reports the same part count, and the fix kernel's (workgroup, slot) list of a tile's parts names exactly
    the segments of that tile, in the order of the token slices."""
    import ctypes

    from ganq_amd import _lib

    h = _lib.lib()
    for n in (1024, 1288, 2048, 3072, 4096, 5120, 6144, 8192, 11008, 14336):
        for rows in (32, 64, 2048, 4128 // 32 * 32, 16384):
            nt = (n + 255) // 256
            T, Ks = nt * (nt + 1) // 2, rows // 32
            hdr = (ctypes.c_int * 7)()
            assert h.ganq_debug_hessian_t_cut(n, rows, ncu, hdr, None, 0, None, 0) == 0
            _ks, G, W, R, nprim, P, Sh = list(hdr)
            assert (_ks, G, W, R) == (Ks, ncu, T // ncu, T % ncu)
            per, maxp = W + 3, 64
            segs = (ctypes.c_int * (ncu * per * 5))()
            loc = (ctypes.c_int * (max(R, 1) * maxp * 2))()
            assert h.ganq_debug_hessian_t_cut(n, rows, ncu, hdr, segs, len(segs), loc, maxp) == 0
            S = np.frombuffer(segs, dtype=np.int32).reshape(ncu, per, 5)
            L = np.frombuffer(loc, dtype=np.int32).reshape(max(R, 1), maxp, 2)
            cover = np.zeros((T, Ks), dtype=np.int32)
            by_tile = {}
            for c in range(ncu):
                slots = []
                for t, s0, s1, np_, slot in S[c]:
                    if t < 0:
                        continue
                    assert 0 <= t < T and 0 <= s0 < s1 <= Ks, (n, rows, c, t, s0, s1)
                    cover[t, s0:s1] += 1
                    by_tile.setdefault(int(t), []).append((int(s0), int(s1), c, int(slot), int(np_)))
                    if np_ > 1:
                        assert 0 <= slot < 3
                        slots.append(int(slot))
                    else:
                        assert (s0, s1) == (0, Ks)  # a lone part is a whole tile (finished from the registers)
                assert len(slots) == len(set(slots)), (n, rows, c, slots)
            assert (cover == 1).all(), (n, rows, ncu, np.argwhere(cover != 1)[:4])
            for t, parts in by_tile.items():
                parts.sort()
                assert all(p[4] == len(parts) for p in parts), (n, rows, t, parts)
                if len(parts) > 1:
                    r = t - W * ncu
                    assert 0 <= r < R and len(parts) <= maxp
                    want = [(c, slot) for _, _, c, slot, _ in parts]
                    got = [tuple(int(v) for v in L[r, i]) for i in range(len(parts))]
                    assert got == want, (n, rows, t, got, want)
                    assert tuple(L[r, len(parts)]) == (-1, -1) if len(parts) < maxp else True

"""Multi-GPU dispatch for the GANQ path: one process per GPU, torch.distributed ("nccl" == RCCL on ROCm,
over xGMI inside a node).  The reference has no distributed code at all; the design
follows SURVEY.md section 8(e).

What shards, and the exchange step each way needs:
  * module level -- the modules of one group (q/k/v, up/gate, or the independent layers of a benchmark) share
    their calibration activations and are otherwise independent: rank r quantizes the modules `assign()` gives
    it.  Exchange: the calibration activations (or the finished Hessian) reach every owner by broadcast /
    all-reduce; the owner broadcasts the quantized result back.
  * row level -- rows of W are independent in S-solve, T-update and k-means (algo.md:10), only best-of-K is
    global: `run_layer_row_sharded` gives rank r a contiguous row slice, all-reduces the K per-iteration
    distances (K doubles) and all-gathers the rows.
Layers of a transformer are sequentially dependent (module_looper.py:354-407) and are never run concurrently.

The compute calls go through `solver` (default: the HIP library); tests inject a CPU solver to exercise the
collective logic under gloo.
"""
import os
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch
import torch.distributed as td


@dataclass
class Dist:
    rank: int = 0
    world: int = 1
    device: Optional[torch.device] = None

    @staticmethod
    def current() -> "Dist":
        if td.is_available() and td.is_initialized():
            dev = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
            return Dist(td.get_rank(), td.get_world_size(), dev)
        return Dist(0, 1, None)


def init_from_env(backend: Optional[str] = None) -> Dist:
    """RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT as set by torch.distributed.run.

    One process per GPU, backend "nccl" (= RCCL).  For rehearsing the multi-rank code path on a box with fewer GPUs
    than ranks, GANQ_DIST_SHARE_DEVICE=1 maps every rank to cuda:0 and uses gloo (RCCL refuses two ranks on one GPU)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        if torch.cuda.is_available():
            torch.cuda.set_device(0)
        return Dist(0, 1, torch.device("cuda", 0) if torch.cuda.is_available() else torch.device("cpu"))
    local = int(os.environ.get("LOCAL_RANK", os.environ.get("RANK", "0")))
    use_cuda = torch.cuda.is_available()
    share = os.environ.get("GANQ_DIST_SHARE_DEVICE", "") == "1"
    if use_cuda:
        torch.cuda.set_device(0 if share else local)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # the host driver only supports dmabuf IPC
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if not td.is_initialized():
        td.init_process_group(backend or ("gloo" if (share or not use_cuda) else "nccl"))
    return Dist.current()


def broadcast_tensor(t: torch.Tensor, src: int) -> torch.Tensor:
    """in-place broadcast; under gloo (rehearsal / CPU tests) device tensors are staged through the host"""
    if td.get_backend() == "gloo" and t.is_cuda:
        host = t.detach().cpu()
        td.broadcast(host, src=src)
        if td.get_rank() != src:
            t.copy_(host)
    else:
        td.broadcast(t, src=src)
    return t


def module_cost(m: int, n: int) -> float:
    return float(m) * float(n) * float(n)  # S-solve / T-update / loss all scale as m * n^2


def assign(shapes: Dict[str, Tuple[int, int]], world: int) -> Dict[str, int]:
    """name -> owner rank; longest-processing-time greedy over cost m*n^2, deterministic on every rank"""
    load = [0.0] * world
    owners = {}
    for name in sorted(shapes, key=lambda k: (-module_cost(*shapes[k]), k)):
        r = min(range(world), key=lambda i: (load[i], i))
        owners[name] = r
        load[r] += module_cost(*shapes[name])
    return owners


def row_slices(m: int, world: int, align: int = 16) -> List[Tuple[int, int]]:
    """contiguous row ranges, multiples of `align` rows (the S-solve workgroup tile) except the last"""
    tiles = (m + align - 1) // align
    base, extra = divmod(tiles, world)
    out, start = [], 0
    for r in range(world):
        cnt = (base + (1 if r < extra else 0)) * align
        end = min(m, start + cnt)
        out.append((start, end))
        start = end
    return out


def broadcast_activations(x: Optional[torch.Tensor], shape, dtype, src: int, dist: Dist) -> torch.Tensor:
    """rank `src` forwards the calibration batch, every rank receives it (north-star variant: RCCL broadcast of X)"""
    if dist.world == 1:
        return x
    buf = x if dist.rank == src else torch.empty(shape, dtype=dtype, device=dist.device)
    return broadcast_tensor(buf, src)


def allreduce_sum(t: torch.Tensor) -> torch.Tensor:
    if td.get_backend() == "gloo" and t.is_cuda:
        host = t.cpu()
        td.all_reduce(host, op=td.ReduceOp.SUM)
        t.copy_(host)
    else:
        td.all_reduce(t, op=td.ReduceOp.SUM)
    return t


def allreduce_hessian(H: torch.Tensor, dist: Dist) -> torch.Tensor:
    """data-parallel calibration: every rank accumulated H over ITS share of the sequences with the global
    sample count as normaliser (H = (2/N) sum_b X_b^T X_b is a plain sum, gptq.py:122-131)"""
    if dist.world > 1:
        allreduce_sum(H)
    return H


def share_module_result(processor, named_module, owner: int, dist: Dist):
    """owner broadcasts (wq, indices, codebook) of a finished module; the other ranks install them"""
    lin = named_module.module
    dev = lin.weight.device
    m, n = named_module.state["out_features"], named_module.state["in_features"]
    meta = torch.zeros(2, dtype=torch.int64, device=dev)
    res = processor.results().get(named_module.full_name) if dist.rank == owner else None
    if dist.rank == owner:
        meta[0] = 1 if res is not None else 0
        meta[1] = res["bits"] if res is not None else 0
    broadcast_tensor(meta, owner)
    if int(meta[0]) == 0:
        return  # module was skipped by the owner
    bits = int(meta[1])
    wq = lin.weight.data if dist.rank == owner else torch.empty_like(lin.weight.data)
    q = res["ganq_q"] if dist.rank == owner else torch.empty((m, n), dtype=torch.uint8, device=dev)
    lut = res["ganq_lut"] if dist.rank == owner else torch.empty((m, 2 ** bits), dtype=torch.float32, device=dev)
    if dist.rank == owner:
        wq, q, lut = wq.contiguous(), q.contiguous(), lut.contiguous()
    for t in (wq, q, lut):
        broadcast_tensor(t, owner)
    if dist.rank != owner:
        lin.weight.data = wq
        named_module.state.update({"wq": wq, "ganq_q": q, "ganq_lut": lut})
        processor.results()[named_module.full_name] = {"scale": None, "zero": None, "g_idx": None, "ganq_q": q,
                                                       "ganq_lut": lut, "bits": bits}


class HipSolver:
    """the product path: every call goes to libganq_hip.so"""

    def __init__(self):
        from . import _lib

        self._lib = _lib

    def matmul(self, A, B):
        return self._lib.matmul_f32(A, B)

    def solve_s(self, W, L, T):
        return self._lib.solve_s(W, L, T)

    def update_t(self, WH, H, Q, V, rcond):
        return self._lib.update_t(WH, H, Q, V, rcond)

    def quad_loss(self, W, H, T, Q):
        return self._lib.quad_loss(W, H, T, Q)


def run_layer_row_sharded(W, H, L, T0, K: int, alias_q: bool = True, rcond: float = -1.0, dist: Optional[Dist] = None,
                          solver=None):
    """ganq.py:516-634 with the rows of W split over the ranks.  Every rank passes the FULL W / T0 (replicated) and
    gets the FULL (T_best, Q, dists, best_k) back.  Exchange: one all-reduce of K doubles, one all-gather of rows."""
    dist = dist or Dist.current()
    solver = solver or HipSolver()
    m, n = W.shape
    V = T0.shape[1]
    lo, hi = row_slices(m, dist.world)[dist.rank]
    Wl, T = W[lo:hi].contiguous(), T0[lo:hi].contiguous()
    WH = solver.matmul(Wl, H) if hi > lo else Wl
    Ts, Qs, ds = [], [], []
    for _ in range(K):
        if hi > lo:
            Q = solver.solve_s(Wl, L, T)
            T = solver.update_t(WH, H, Q, V, rcond)
            d = solver.quad_loss(Wl, H, T, Q).reshape(1).to(torch.float64)
        else:
            Q = torch.empty((0, n), dtype=torch.uint8, device=W.device)
            d = torch.zeros(1, dtype=torch.float64, device=W.device)
        Ts.append(T)
        Qs.append(Q)
        ds.append(d)
    dists = torch.cat(ds)
    if dist.world > 1:
        allreduce_sum(dists)  # best-of-K is global over rows (ganq.py:622-626)
    best_k = int(torch.argmin(dists))  # first minimum == the reference's strict `<` scan
    T_loc = Ts[best_k]
    Q_loc = Qs[K - 1] if alias_q else Qs[best_k]
    if dist.world == 1:
        return T_loc, Q_loc, dists, best_k
    T_full = torch.empty((m, V), dtype=T0.dtype, device=W.device)
    Q_full = torch.empty((m, n), dtype=torch.uint8, device=W.device)
    for r, (a, b) in enumerate(row_slices(m, dist.world)):  # ragged slices: one broadcast per owner
        if b > a:
            tt = T_loc if r == dist.rank else torch.empty((b - a, V), dtype=T0.dtype, device=W.device)
            qq = Q_loc if r == dist.rank else torch.empty((b - a, n), dtype=torch.uint8, device=W.device)
            broadcast_tensor(tt, r)
            broadcast_tensor(qq, r)
            T_full[a:b], Q_full[a:b] = tt, qq
    return T_full, Q_full, dists, best_k

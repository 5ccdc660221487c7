"""Multi-GPU dispatch for the GANQ path: one process per GPU, torch.distributed ("nccl" == RCCL on ROCm,
over xGMI inside a node).  The reference has no distributed code at all; the design
follows SURVEY.md section 8(e).

What shards, and the exchange step each way needs:
  * module level -- the modules of one group (q/k/v, up/gate, or the independent layers of a benchmark) share
    their calibration activations and are otherwise independent: rank r quantizes the modules `assign()` gives
    it.  Exchange: the calibration activations (or the finished Hessian) reach every owner by broadcast /
    all-reduce; the owner broadcasts the quantized result back.
  * row level -- rows of W are independent in S-solve, T-update and k-means (algo.md:10), only best-of-K is
    global: `run_layer_row_sharded` gives rank r a contiguous row slice, runs the fused loop on it (no collective
    inside), all-gathers the K x m per-row losses (the decision) and then the chosen rows.
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
    """owner broadcasts everything `GPTQProcessor.finalize` / `pack` need of a finished module -- the quantized weight,
    indices, codebook, the compat scale / zero / g_idx and, with the outlier split on, the CSR outliers -- and the
    other ranks install them, so every rank (the one that saves the checkpoint included) holds identical results."""
    lin = named_module.module
    dev = lin.weight.device
    m, n = named_module.state["out_features"], named_module.state["in_features"]
    # meta: present, bits, nnz (-1: no outlier split), scale columns, zero columns
    meta = torch.zeros(5, dtype=torch.int64, device=dev)
    res = processor.results().get(named_module.full_name) if dist.rank == owner else None
    if dist.rank == owner and res is not None:
        out = res.get("ganq_outliers")
        meta[0], meta[1] = 1, res["bits"]
        meta[2] = -1 if out is None else int(out[1].numel())
        meta[3] = 0 if res.get("scale") is None else res["scale"].reshape(m, -1).shape[1]
        meta[4] = 0 if res.get("zero") is None else res["zero"].reshape(m, -1).shape[1]
    broadcast_tensor(meta, owner)
    if int(meta[0]) == 0:
        return  # module was skipped by the owner
    bits, nnz, sc_cols, ze_cols = int(meta[1]), int(meta[2]), int(meta[3]), int(meta[4])
    is_owner = dist.rank == owner

    def take(t, shape, dtype):
        """the owner's tensor (on the device, contiguous) or a receive buffer of the same shape"""
        if is_owner:
            return t.to(device=dev, dtype=dtype).reshape(shape).contiguous()
        return torch.empty(shape, dtype=dtype, device=dev)

    wq = take(lin.weight.data if is_owner else None, tuple(lin.weight.shape), lin.weight.dtype)
    q = take(res["ganq_q"] if is_owner else None, (m, n), torch.uint8)
    lut = take(res["ganq_lut"] if is_owner else None, (m, 2 ** bits), torch.float32)
    g_idx = take(res["g_idx"] if is_owner else None, (n,), torch.int32)
    payload = [wq, q, lut, g_idx]
    scale = zero = None
    if sc_cols:
        scale = take(res["scale"] if is_owner else None, (m, sc_cols), torch.float32)
        payload.append(scale)
    if ze_cols:
        zero = take(res["zero"] if is_owner else None, (m, ze_cols), torch.float32)
        payload.append(zero)
    outliers = None
    if nnz >= 0:
        rowptr = take(res["ganq_outliers"][0] if is_owner else None, (m + 1,), torch.int32)
        cols = take(res["ganq_outliers"][1] if is_owner else None, (nnz,), torch.int32)
        vals = take(res["ganq_outliers"][2] if is_owner else None, (nnz,), torch.float32)
        outliers = (rowptr, cols, vals)
        payload += [rowptr] + ([cols, vals] if nnz else [])
    for t in payload:
        broadcast_tensor(t, owner)
    if not is_owner:
        lin.weight.data = wq
        named_module.state.update({"wq": wq, "ganq_q": q, "ganq_lut": lut})
        processor.results()[named_module.full_name] = {"scale": scale, "zero": zero, "g_idx": g_idx, "ganq_q": q,
                                                       "ganq_lut": lut, "bits": bits, "ganq_outliers": outliers}


class HipSolver:
    """the product path: every call goes to libganq_hip.so"""

    def __init__(self, helpers: bool = True):
        from . import _lib

        self._lib = _lib
        self.helpers = helpers  # False: the S-solve launches no helper workgroups (modules of a group run side by side)

    def run_layer_rows(self, W, H, L, T0, K, alias_q, rcond):
        """fused K-iteration loop on a row slice -> dict(T_all [K,m,V], loss_rows_all [K,m], Q_last, Q_all or None)"""
        return self._lib.run_layer_rows(W, H, L, T0, K, alias_q=alias_q, rcond=rcond, helpers=self.helpers)

    def select_best(self, loss_rows_all):
        """[K, m] per-row losses of the whole layer -> (dists [K], best_k) in the single-GPU loop's summation order"""
        return self._lib.select_best(loss_rows_all)


def allgather_rows(local: torch.Tensor, slices: List[Tuple[int, int]], dim: int, dist: Dist) -> torch.Tensor:
    """concatenate every rank's rows (ragged slices) along `dim`: ONE all-gather of equal-sized padded blocks"""
    if dist.world == 1:
        return local
    pad = max(b - a for a, b in slices)
    shape = list(local.shape)
    shape[dim] = pad
    block = torch.zeros(shape, dtype=local.dtype, device=local.device)
    if local.shape[dim]:
        block.narrow(dim, 0, local.shape[dim]).copy_(local)
    if td.get_backend() == "gloo" and block.is_cuda:
        parts = [torch.empty(shape, dtype=local.dtype) for _ in range(dist.world)]
        td.all_gather(parts, block.cpu())
        parts = [p.to(local.device) for p in parts]
    else:
        parts = [torch.empty_like(block) for _ in range(dist.world)]
        td.all_gather(parts, block)
    return torch.cat([parts[r].narrow(dim, 0, b - a) for r, (a, b) in enumerate(slices)], dim=dim)


class _Timer:
    """wall time of the exchange steps, only when the caller asks for it (a synchronize on both sides of every
    collective would otherwise serialise the host with the device)"""

    def __init__(self, stats, device):
        self.stats, self.on = stats, bool(stats is not None and stats.get("timing"))
        self.device = device

    def __call__(self, key):
        return _TimerCtx(self, key)


class _TimerCtx:
    def __init__(self, t, key):
        self.t, self.key = t, key

    def __enter__(self):
        if self.t.on:
            import time

            if self.t.device is not None and self.t.device.type == "cuda":
                torch.cuda.synchronize(self.t.device)
            self.t0 = time.perf_counter()

    def __exit__(self, *exc):
        if self.t.on:
            import time

            if self.t.device is not None and self.t.device.type == "cuda":
                torch.cuda.synchronize(self.t.device)
            self.t.stats[self.key] = self.t.stats.get(self.key, 0.0) + time.perf_counter() - self.t0


class CollectiveTurns:
    """Several modules of a group are quantized at the same time (worker threads, side streams) while every module's
    exchange is a sequence of collectives: those must be ISSUED in the same order on every rank.  Module i's thread enters
    `turn(i)` before its first collective and leaves it after the last; turns are granted in index order, so the order of
    the collectives is the group's module order on every rank, whichever thread finishes its local work first.  A thread
    that fails marks the object broken, which raises in the waiting ones instead of leaving them (and the other ranks)
    hanging."""

    def __init__(self):
        import threading

        self._cv = threading.Condition()
        self._next = 0
        self._broken = None

    def turn(self, index: int):
        return _Turn(self, index)

    def fail(self, exc):
        with self._cv:
            if self._broken is None:
                self._broken = exc
            self._cv.notify_all()


class _Turn:
    def __init__(self, owner, index):
        self.o, self.i = owner, index

    def __enter__(self):
        with self.o._cv:
            while self.o._next != self.i and self.o._broken is None:
                self.o._cv.wait()
            if self.o._broken is not None:
                raise RuntimeError(f"another module of the group failed: {self.o._broken!r}")
        return self

    def __exit__(self, et, ev, tb):
        with self.o._cv:
            if et is not None and self.o._broken is None:
                self.o._broken = ev
            self.o._next = self.i + 1
            self.o._cv.notify_all()
        return False


def _raise_together(err: Optional[BaseException], what: str, device, dist: Dist) -> None:
    """one int per rank, MAX-reduced: a rank-local failure (`err`) raises on EVERY rank instead of leaving the others in the
    next collective until the RCCL timeout"""
    flag = torch.tensor([0 if err is None else 1 + dist.rank], dtype=torch.int64, device=device)
    if dist.world > 1:
        if td.get_backend() == "gloo" and flag.is_cuda:
            host = flag.cpu()
            td.all_reduce(host, op=td.ReduceOp.MAX)
            flag = host
        else:
            td.all_reduce(flag, op=td.ReduceOp.MAX)
    bad = int(flag)
    if err is not None:
        raise err
    if bad:
        raise RuntimeError(f"{what}: rank {bad - 1} failed in its local part; every rank stops")


def run_layer_row_sharded(W, H, L, T0, K: int, alias_q: bool = True, rcond: float = -1.0, dist: Optional[Dist] = None,
                          solver=None, t0_fn=None, stats: Optional[dict] = None, turn=None, V: Optional[int] = None):
    """ganq.py:501-634 with the rows of W split over the ranks.  Every rank passes the FULL W (replicated) and gets the
    FULL (T_best, Q, dists, best_k) back.  T0: the full initial codebook (replicated), or None with
    `t0_fn(W_rows) -> T0_rows`: then the codebook initialisation (ganq.py:423-438, rows are independent) is sharded as
    well -- every rank clusters its own rows only and the initial codebook is never exchanged.

    Each rank runs the FUSED loop (ganq_run_layer_rows: incremental bucket sums, closed-form loss, packed L -- the same
    kernels as the single-GPU path) on its row slice with no collective inside; afterwards ONE all-gather of the K x m
    per-row losses lets every rank form the K distances in the single-GPU summation order (ganq_select_best), so the
    best-of-K decision (ganq.py:621-626) and every returned bit equal the unsharded run's; then one all-gather of the
    chosen codebook rows and one of the index rows.
    stats: optional dict; with stats["timing"] set, "kmeans_s" / "loop_s" / "collective_s" are added up in it.
    turn: optional context manager (CollectiveTurns.turn(i)) entered around the exchange, when the modules of a group run
    their local parts concurrently.
    V: codebook width (2^bits); needed by a rank that owns no rows when T0 is None.
    A failure in a rank's local part (k-means, fused loop) is exchanged before the first all-gather: every rank raises."""
    import contextlib

    dist = dist or Dist.current()
    solver = solver or HipSolver()
    m, n = W.shape
    timer = _Timer(stats, W.device)
    # 128-row alignment keeps the per-128-row decisions of the W @ H kernel identical to the unsharded run
    slices = row_slices(m, dist.world, align=128 if m >= 128 * dist.world else 16)
    lo, hi = slices[dist.rank]
    if T0 is not None:
        V = T0.shape[1]
    err = None
    loss_loc = T_all = Q_last = Q_all = None
    try:
        if hi > lo:
            W_loc = W[lo:hi].contiguous()
            with timer("kmeans_s"):
                T0_loc = T0[lo:hi].contiguous() if T0 is not None else t0_fn(W_loc)
            V = T0_loc.shape[1]
            with timer("loop_s"):
                rec = solver.run_layer_rows(W_loc, H, L, T0_loc, K, alias_q, rcond)
            loss_loc, T_all, Q_last, Q_all = rec["loss_rows_all"], rec["T_all"], rec["Q_last"], rec["Q_all"]
        else:
            if V is None:
                raise ValueError("run_layer_row_sharded: a rank without rows needs the codebook width: pass V= (2 ** bits) or T0")
            loss_loc = torch.zeros((K, 0), dtype=torch.float64, device=W.device)
            T_all = torch.zeros((K, 0, V), dtype=torch.float32, device=W.device)
            Q_last = torch.zeros((0, n), dtype=torch.uint8, device=W.device)
            Q_all = None if alias_q else torch.zeros((K, 0, n), dtype=torch.uint8, device=W.device)
    except BaseException as e:  # exchanged below, inside this module's turn (the collectives keep their order on every rank)
        err = e
    with (turn if turn is not None else contextlib.nullcontext()):
        with timer("collective_s"):
            _raise_together(err, "run_layer_row_sharded", W.device, dist)
            loss_all = allgather_rows(loss_loc, slices, 1, dist)  # K x m doubles: the only exchange the decision needs
        dists, best_k_t = solver.select_best(loss_all)
        best_k = int(best_k_t)
        kk = best_k if best_k >= 0 else K - 1  # no iteration won (all NaN): the last codebook, like the single-GPU loop
        T_loc = T_all[kk]
        Q_loc = Q_last if (alias_q or best_k < 0) else Q_all[kk]
        with timer("collective_s"):
            T_full = allgather_rows(T_loc.contiguous(), slices, 0, dist)
            Q_full = allgather_rows(Q_loc.contiguous(), slices, 0, dist)
    return T_full, Q_full, dists, best_k


def reduce_group_statistics(tasks, dist: Dist, stats: Optional[dict] = None) -> None:
    """Data-parallel calibration (SURVEY 8(e) axis 1): every rank forwarded ITS share of the calibration sequences and
    holds, per hooked task, the partial Hessian H_r = (2/N_r) sum_{b in r} X_b^T X_b (gptq.py:122-131) with its own
    sample count N_r.  The full statistic is the sample-weighted sum, H = sum_r (N_r / N) H_r: one all-reduce of the
    counters, then one of n^2 floats per task (64 MiB at n = 4096).  Afterwards every rank holds the same H, nsamples
    and fwd_counter -- the looper's "was this module ever invoked" decision (module_looper.py:332-343) included."""
    if dist.world <= 1 or not tasks:
        return
    dev = dist.device or torch.device("cpu")
    timer = _Timer(stats, dev)
    local = torch.tensor([[t.fwd_counter, t.nsamples] for t in tasks], dtype=torch.float64, device=dev)
    total = local.clone()
    with timer("collective_s"):
        allreduce_sum(total)
    for i, t in enumerate(tasks):
        n_loc, n_all = int(local[i, 1]), int(total[i, 1])
        if n_all == 0:
            t.fwd_counter = 0
            continue
        if hasattr(t, "H"):
            H = t.hessian  # flushes the staged batches
            H.mul_(n_loc / n_all)
        else:
            H = torch.zeros((t.columns, t.columns), dtype=torch.float32, device=t.device)
        with timer("collective_s"):
            allreduce_sum(H)
        t.H, t.nsamples, t.fwd_counter = H, n_all, int(total[i, 0])


def broadcast_calibration_batch(x: Optional[torch.Tensor], owner: int, dist: Dist, like_dtype=None) -> Optional[torch.Tensor]:
    """north-star variant of the exchange: the rank that forwarded a calibration batch broadcasts the activations entering
    a hooked module; every rank then accumulates the SAME batches in the SAME order as a single-GPU run, so the Hessian
    -- and everything after it -- is bit-identical to world = 1.  x is None on the receivers (and on the owner when the
    module was not invoked for this batch: then None comes back everywhere)."""
    dev = dist.device or torch.device("cpu")
    codes = {torch.float16: 0, torch.bfloat16: 1, torch.float32: 2}
    # meta: state (0 none, 1 tensor follows, -1 the owner cannot send this batch), ndim, up to 4 dims, dtype code
    meta = torch.zeros(7, dtype=torch.int64, device=dev)
    problem = None
    if dist.rank == owner and x is not None:
        if x.dtype not in codes:
            problem = f"activations of dtype {x.dtype} (fp16 / bf16 / fp32 are exchanged)"
        elif not 1 <= x.dim() <= 4:
            problem = f"activations with {x.dim()} dimensions"
        if problem is None:
            shp = list(x.shape)
            meta[0], meta[1], meta[6] = 1, len(shp), codes[x.dtype]
            for i, d in enumerate(shp):
                meta[2 + i] = d
        else:
            meta[0] = -1  # every rank raises together instead of the receivers waiting in the broadcast
    broadcast_tensor(meta, owner)
    if int(meta[0]) < 0:
        raise ValueError("broadcast_calibration_batch: rank %d cannot send its batch%s" % (owner, f": {problem}" if problem else ""))
    if int(meta[0]) == 0:
        return None
    shape = tuple(int(v) for v in meta[2:2 + int(meta[1])])
    dtype = {v: k for k, v in codes.items()}[int(meta[6])]
    buf = x.contiguous() if dist.rank == owner else torch.empty(shape, dtype=dtype, device=dev)
    return broadcast_tensor(buf, owner)

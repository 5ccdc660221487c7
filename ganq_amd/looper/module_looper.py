"""Layer-by-layer calibration driver: the caller of the hot path (reference gptqmodel/looper/module_looper.py:
129-452), kept on the device end to end.

For every decoder layer and every module group inside it (`layer_modules`, e.g. [[q,k,v],[o],[fc1],[fc2]],
models/definitions/opt.py:36-41): install forward hooks -> run all calibration batches through the layer
(hooks stream activations into GANQ.add_batch) -> processor.process(module) for every module of the group ->
after the last group, re-run the batches through the now-quantized layer to produce the next layer's inputs
(module_looper.py:354-396).  Unlike the reference, activations never bounce through host memory
(module_looper.py:289-302).

With torch.distributed initialised ("looper dispatches layers over RCCL ranks") the default is
`dist_mode="rows"`: BOTH axes on which the path shards are used, for every module of every group --
  * calibration is data-parallel: rank r forwards the calibration batches b = r (mod world) through its replica of the
    layer (the layer's outputs, i.e. the next layer's inputs, stay with the rank that produced them);
    `calibration="allreduce"` (default): every rank accumulates a partial Hessian over its batches and the group's
    statistics meet in ONE all-reduce of n^2 floats (ganq_amd.distributed.reduce_group_statistics);
    `calibration="broadcast"` (the variant BASELINE.json's north star names): the forwarding rank broadcasts the
    activations entering the hooked module and every rank accumulates all batches in single-GPU order -- 33x the bytes
    at n = 4096, but every bit of the result equals the single-GPU run's;
  * the quantization of a module is row-parallel: k-means, S-solve and T-update of rank r's row slice, one exchange of
    the K x m row losses for best-of-K, one all-gather of the chosen codebook rows / indices
    (GANQ.row_dist -> ganq_amd.distributed.run_layer_row_sharded).  The prologue (two factorisations) is replicated.
    The followers of a group's shared prologue run their local parts beside the leader's as on one GPU; the exchanges are
    issued in the group's module order on every rank (ganq_amd.distributed.CollectiveTurns).
`dist_mode="modules"` is the coarser scheme of round 1: the modules of a group are dealt to the ranks
(ganq_amd.distributed.assign), every rank forwards everything, owners broadcast their results.
Either way every rank ends with identical quantized layers.
"""
from typing import Callable, Dict, List, Optional, Sequence

import warnings

import torch
import torch.nn as nn

from .. import distributed as gdist
from .named_module import NamedModule


class _StopForward(Exception):
    """raised by a hook once every module of the current group has seen the batch: the rest of the layer's forward
    only produces an output that the calibration passes throw away"""


def find_modules(layer: nn.Module, names: Sequence[str]) -> Dict[str, nn.Module]:
    named = dict(layer.named_modules())
    return {n: named[n] for n in names if n in named}


def _result_tensors(processor, named_module):
    """the device tensors a finished module leaves behind (weight, indices, codebook, compat values, outliers)"""
    out = []
    res = processor.results().get(named_module.full_name, {})
    vals = list(res.values()) + [named_module.state.get("wq")]
    for v in vals:
        for t in (v if isinstance(v, (tuple, list)) else (v,)):
            if isinstance(t, torch.Tensor) and t.is_cuda:
                out.append(t)
    return out


class ModuleLooper:
    def __init__(self, processor, layers: Sequence[nn.Module], layer_modules: List[List[str]],
                 layers_prefix: str = "model.layers", share_group_hessian: bool = False, early_exit: bool = True,
                 cache_outputs: bool = True, cache_budget_bytes: int = 48 << 30, concurrent_group: bool = True,
                 dist_mode: str = "rows", calibration: str = "allreduce"):
        # share_group_hessian: the modules of one group ([q,k,v], [gate,up]) receive the same inputs, so the first
        # one accumulates the Hessian and runs the prologue (permutation, factor, damping) for all of them; the
        # reference does both once per module.  Same numbers, less work.  Only valid when the groups really share
        # their inputs (dense q/k/v, gate/up): the experts of a mixture-of-experts group do not -- hence opt-in; the
        # layer maps of ganq_amd.models say which it is.
        self.share_group_hessian = share_group_hessian
        # early_exit: a calibration pass of a group stops the layer's forward as soon as the group's hooked modules
        # have all been called (the reference runs the whole layer every time, module_looper.py:287-316, and discards
        # the output); same statistics, about a third less forward work per layer.  Turn it off for a layer that calls
        # one of its Linear modules more than once per forward.
        self.early_exit = early_exit
        # cache_outputs: once a module is quantized its weights are final, and (the groups being in forward order) so
        # are its inputs: the later passes over the same layer reuse its per-batch outputs instead of recomputing the
        # GEMM.  Same tensors, about a third less forward work again; the outputs of one layer stay on the device until
        # the layer is done (up to cache_budget_bytes).  Turn it off together with early_exit for unusual layers.
        self.cache_outputs = cache_outputs
        self.cache_budget_bytes = int(cache_budget_bytes)
        # concurrent_group (with share_group_hessian, on a GPU): once the group's leader has finished the shared prologue,
        # its followers are quantized on side streams beside the leader's own loop.  The solve of a small module is a
        # latency chain that leaves most of the chip idle (k / v projections: 32 workgroups on 256 CUs), and the modules
        # of a group are independent given the prologue.  Same numbers: every module's work is the same sequence of
        # launches on its own stream.
        self.concurrent_group = concurrent_group
        # multi-GPU scheme (module docstring); ignored when torch.distributed is not initialised
        if dist_mode not in ("rows", "modules", "none") or calibration not in ("allreduce", "broadcast"):
            raise ValueError(f"ModuleLooper: dist_mode={dist_mode!r} / calibration={calibration!r}")
        self.dist_mode, self.calibration = dist_mode, calibration
        self._turns = None  # the CollectiveTurns of the row-sharded group being processed
        self.dist_stats = {"timing": False}  # set ["timing"] = True to have the exchange steps timed (adds synchronisation)
        self.processor = processor
        self.layers = layers
        self.layer_modules = layer_modules
        self.layers_prefix = layers_prefix

    def _process_group(self, todo, named):
        """processor.process() for the modules of one group: one after the other, or -- followers of a shared prologue on a
        GPU -- the followers on side streams in worker threads, started by the leader the moment its prologue is shared."""
        tasks = self.processor.tasks
        followers = [n for n in todo if getattr(tasks[n], "_group_leader", None) is not None]
        leaders = [n for n in todo if n not in followers]
        dev = next(iter(named.values())).module.weight.device if named else None
        if not (self.concurrent_group and followers and len(leaders) == 1 and dev is not None and dev.type == "cuda"
                and all(tasks[f]._group_leader is tasks[leaders[0]] for f in followers)):
            for n in todo:
                self.processor.process(named[n])
            return
        import threading

        main = torch.cuda.current_stream(dev)
        errors, threads, streams = [], [], []
        log_rows = {}  # processor.log rows per module, re-assembled in `todo` order after the join
        log_before = len(self.processor.log)

        def run(n, stream):
            try:
                # grad mode is thread-local: the caller's no_grad() does not reach a worker thread
                with torch.no_grad(), torch.cuda.device(dev), torch.cuda.stream(stream):
                    self.processor.process(named[n])
                    # the follower's results were allocated under its side stream and are read (and freed) under the
                    # main stream from here on: tell the caching allocator
                    for t in _result_tensors(self.processor, named[n]):
                        t.record_stream(main)
            except BaseException as e:  # re-raised on the caller's thread
                errors.append(e)
                if self._turns is not None:  # row-sharded group: do not leave the other modules waiting for this one's exchange
                    self._turns.fail(e)

        def start_followers():
            shared = getattr(tasks[followers[0]], "_leader_prologue", None) or {}
            for n in followers:
                st = torch.cuda.Stream(device=dev)
                st.wait_stream(main)  # the shared prologue was enqueued on the leader's stream
                # ... and allocated under it: a block the leader drops must not be reused while a side stream reads it
                for t in shared.values():
                    if isinstance(t, torch.Tensor) and t.is_cuda:
                        t.record_stream(st)
                th = threading.Thread(target=run, args=(n, st), name=f"ganq-{n}")
                streams.append(st)
                threads.append(th)
                th.start()

        for n in todo:  # their S-solves share the CUs: no helper workgroups (solve_s.hip; the bits do not depend on it)
            if hasattr(tasks[n], "solve_helpers"):
                tasks[n].solve_helpers = False
        tasks[leaders[0]]._on_prologue_shared = start_followers
        try:
            self.processor.process(named[leaders[0]])
        except BaseException as e:
            if self._turns is not None:
                self._turns.fail(e)  # the followers must not wait for the leader's turn at the exchange
            raise
        finally:
            for th in threads:
                th.join()
            for st in streams:
                main.wait_stream(st)
        if followers and not threads:  # the leader never reached the hand-over (it keeps its hook only while it lives)
            for n in followers:
                self.processor.process(named[n])
        # the worker threads appended their log rows in completion order: restore the group's module order
        new_rows = self.processor.log[log_before:]
        for row in new_rows:
            log_rows.setdefault(row.get("module"), []).append(row)
        ordered = [r for n in todo for r in log_rows.pop(n, [])] + [r for rows in log_rows.values() for r in rows]
        self.processor.log[log_before:] = ordered
        if errors:
            raise errors[0]

    @torch.no_grad()
    def loop(self, layer_inputs: List[torch.Tensor], layer_kwargs: Optional[List[dict]] = None,
             forward: Optional[Callable] = None, progress: Optional[Callable] = None, inputs_are_local: bool = False):
        """layer_inputs: hidden states entering layer 0, one tensor per calibration batch ([b, seq, hidden]);
        layer_kwargs: per-batch keyword arguments of the layer forward (attention mask, position ids, ...).
        Returns the hidden states leaving the last layer.
        Several ranks, dist_mode="rows": rank r works on the batches r, r + world, ... -- pass all batches (every rank
        picks its own) or, with inputs_are_local=True, only those; the returned list holds the rank's own batches."""
        layer_kwargs = layer_kwargs or [{} for _ in layer_inputs]
        fwd = forward or (lambda layer, x, kw: layer(x, **kw))
        # dist_mode="none": this process quantizes on its own even if torch.distributed is initialised (A/B runs)
        dist = gdist.Dist.current() if self.dist_mode != "none" else gdist.Dist(0, 1, None)
        sharded = dist.world > 1 and self.dist_mode == "rows"
        n_global = len(layer_inputs)
        if sharded:
            if inputs_are_local:
                cnt = torch.tensor([len(layer_inputs)], dtype=torch.float64, device=dist.device or "cpu")
                n_global = int(gdist.allreduce_sum(cnt))
                if len(layer_inputs) != len(range(dist.rank, n_global, dist.world)):
                    raise ValueError("ModuleLooper: with inputs_are_local the ranks must hold the batches r, r + world, ...")
            else:
                layer_inputs = list(layer_inputs[dist.rank::dist.world])
                layer_kwargs = list(layer_kwargs[dist.rank::dist.world])
        for li, layer in enumerate(self.layers):
            cur = {"batch": 0, "pass": 0}
            wrapped, cached_bytes = [], [0]

            def cache_module(mod):
                store = [None] * len(layer_inputs)
                served = [-1] * len(layer_inputs)
                orig = mod.forward

                def forward_cached(*a, **k):
                    b = cur["batch"]
                    if served[b] == cur["pass"]:
                        raise RuntimeError("ModuleLooper: a quantized module is called twice in one forward of its layer; "
                                           "construct the looper with cache_outputs=False, early_exit=False")
                    served[b] = cur["pass"]
                    y = store[b]
                    if y is None:
                        y = orig(*a, **k)
                        if isinstance(y, torch.Tensor) and cached_bytes[0] + y.numel() * y.element_size() <= self.cache_budget_bytes:
                            store[b] = y
                            cached_bytes[0] += y.numel() * y.element_size()
                    return y

                mod.forward = forward_cached  # instance attribute: shadows the class method until it is deleted
                wrapped.append(mod)

            for names in self.layer_modules:
                mods = find_modules(layer, names)
                if not mods:
                    continue
                named = {n: NamedModule(m, name=n, full_name=f"{self.layers_prefix}.{li}.{n}", layer_index=li)
                         for n, m in mods.items()}
                if sharded:  # every rank works on every module (its share of the batches, its slice of the rows)
                    owners = {n: dist.rank for n in named}
                else:
                    owners = gdist.assign({n: (nm.state["out_features"], nm.state["in_features"])
                                           for n, nm in named.items()}, dist.world)
                mine = [n for n in named if owners[n] == dist.rank]
                handles = []
                hooked_names = []
                captured = {}  # calibration="broadcast": what the hooks of this rank saw in the current batch
                leader = None  # the modules of a group see the same inputs: one Hessian / prologue for all of them
                for n in mine:
                    self.processor.preprocess(named[n], buffered_fwd=False)
                    if self.processor.is_skipped(named[n]):
                        continue
                    task = self.processor.tasks[n]
                    if sharded:
                        task.row_dist = dist
                        task.time_collectives = bool(self.dist_stats.get("timing"))
                    # a follower takes the leader's Hessian AND prologue: only when every setting the prologue depends on
                    # is the same (a `dynamic` override of damp_percent / act_sort for one module makes it its own leader)
                    if (self.share_group_hessian and leader is not None
                            and self.processor.tasks[leader]._prologue_key() == task._prologue_key()):
                        task.follow(self.processor.tasks[leader])
                        continue
                    if leader is None:
                        leader = n
                    hooked_names.append(n)
                    if sharded and self.calibration == "broadcast":
                        def capture_hook(_mod, inp, _out, _n=n):
                            captured[_n] = inp[0].data
                        handles.append(mods[n].register_forward_hook(capture_hook))
                    else:
                        handles.append(mods[n].register_forward_hook(self.processor.preprocess_fwd_hook(n)))
                hooked, fired = len(handles), set()
                if self.early_exit and hooked:
                    def stop_hook(mod, _inp, _out):
                        if id(mod) in fired:
                            raise RuntimeError("ModuleLooper: a module of the group is called twice in one forward of its "
                                               "layer; construct the looper with early_exit=False, cache_outputs=False")
                        fired.add(id(mod))
                        if len(fired) == hooked:
                            raise _StopForward

                    # registered after the statistics hooks, so it runs after them
                    handles += [mods[n].register_forward_hook(stop_hook) for n in hooked_names]

                def forward_local(bi):
                    fired.clear()
                    cur["batch"] = bi
                    cur["pass"] += 1
                    try:
                        fwd(layer, layer_inputs[bi], layer_kwargs[bi])
                    except _StopForward:
                        pass

                if sharded and self.calibration == "broadcast":
                    # every rank accumulates every batch, in the order a single GPU would: the rank that owns batch b
                    # forwards it and broadcasts what entered each hooked module
                    for b in range(n_global):
                        owner = b % dist.world
                        captured.clear()
                        if owner == dist.rank:
                            forward_local(b // dist.world)
                        for n in hooked_names:
                            x = gdist.broadcast_calibration_batch(captured.get(n), owner, dist)
                            if x is not None:
                                self.processor.tasks[n].add_batch(x, None)
                else:
                    for bi in range(len(layer_inputs)):
                        forward_local(bi)
                    if sharded:  # partial Hessians of the ranks' shares -> the group's statistics, on every rank
                        gdist.reduce_group_statistics([self.processor.tasks[n] for n in hooked_names], dist, self.dist_stats)
                for h in handles:
                    h.remove()
                for n in hooked_names:  # staged calibration batches -> Hessian kernel, staging buffers freed
                    end = getattr(self.processor.tasks[n], "end_of_calibration", None)
                    if end is not None:
                        end()
                todo = []
                for n in mine:
                    if self.processor.is_skipped(named[n]):
                        continue
                    task = self.processor.tasks[n]
                    lead = getattr(task, "_group_leader", None)
                    if lead is not None:
                        task.fwd_counter, task.nsamples = lead.fwd_counter, lead.nsamples
                    if task.fwd_counter == 0:
                        # never invoked (an expert no calibration token was routed to): like the reference
                        # (module_looper.py:332-343) report it and leave the module as it is
                        warnings.warn(f"`{named[n].full_name}` was not invoked during calibration and stays unquantized "
                                      f"(a MoE expert may lack calibration tokens routed to it)")
                        self.processor.skip(named[n])
                        continue
                    todo.append(n)
                if sharded:
                    # every module's row-sharded loop ends in collectives, which must be issued in ONE order on every rank:
                    # the modules of a group still run their local parts (k-means + fused loop on the rank's rows) side by
                    # side like on one GPU, and take turns, in `todo` order, for the exchange
                    turns = gdist.CollectiveTurns()
                    for ti, n in enumerate(todo):
                        self.processor.tasks[n]._collective_turn = turns.turn(ti)
                    self._turns = turns
                    try:
                        self._process_group(todo, named)
                    finally:
                        self._turns = None
                    for n in todo:
                        for k, v in getattr(named[n], "state", {}).get("ganq_stats", {}).items():
                            if k.endswith("_s"):
                                self.dist_stats[k] = self.dist_stats.get(k, 0.0) + v
                else:
                    self._process_group(todo, named)
                for n in todo:
                    if self.cache_outputs:
                        cache_module(mods[n])
                    if progress:
                        progress(named[n])
                if dist.world > 1 and not sharded:
                    for n in named:  # owner broadcasts its result so every rank holds the quantized group
                        gdist.share_module_result(self.processor, named[n], owners[n], dist)
                        if self.cache_outputs and n not in mine:
                            cache_module(mods[n])
            outs = []
            for bi, (x, kw) in enumerate(zip(layer_inputs, layer_kwargs)):
                cur["batch"] = bi
                cur["pass"] += 1
                y = fwd(layer, x, kw)
                outs.append(y[0] if isinstance(y, (tuple, list)) else y)
            for mod in wrapped:
                del mod.forward  # back to the class method; the cached outputs go with the closure
            layer_inputs = outs
        return layer_inputs

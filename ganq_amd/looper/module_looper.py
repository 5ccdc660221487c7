"""Layer-by-layer calibration driver: the caller of the hot path (reference gptqmodel/looper/module_looper.py:
129-452), kept on the device end to end.

For every decoder layer and every module group inside it (`layer_modules`, e.g. [[q,k,v],[o],[fc1],[fc2]],
models/definitions/opt.py:36-41): install forward hooks -> run all calibration batches through the layer
(hooks stream activations into GANQ.add_batch) -> processor.process(module) for every module of the group ->
after the last group, re-run the batches through the now-quantized layer to produce the next layer's inputs
(module_looper.py:354-396).  Unlike the reference, activations never bounce through host memory
(module_looper.py:289-302).

With torch.distributed initialised, the modules of a group -- which share their calibration inputs -- are
dealt to the ranks (ganq_amd.distributed.assign) and their results broadcast back, so every rank continues
with identical quantized layers ("looper dispatches layers over RCCL ranks").
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


class ModuleLooper:
    def __init__(self, processor, layers: Sequence[nn.Module], layer_modules: List[List[str]],
                 layers_prefix: str = "model.layers", share_group_hessian: bool = False, early_exit: bool = True,
                 cache_outputs: bool = True, cache_budget_bytes: int = 48 << 30, concurrent_group: bool = True):
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

        def run(n, stream):
            try:
                with torch.cuda.device(dev), torch.cuda.stream(stream):
                    self.processor.process(named[n])
            except BaseException as e:  # re-raised on the caller's thread
                errors.append(e)

        def start_followers():
            for n in followers:
                st = torch.cuda.Stream(device=dev)
                st.wait_stream(main)  # the shared prologue was enqueued on the leader's stream
                th = threading.Thread(target=run, args=(n, st), name=f"ganq-{n}")
                streams.append(st)
                threads.append(th)
                th.start()

        tasks[leaders[0]]._on_prologue_shared = start_followers
        try:
            self.processor.process(named[leaders[0]])
        finally:
            for th in threads:
                th.join()
            for st in streams:
                main.wait_stream(st)
        if followers and not threads:  # the leader never reached the hand-over (it keeps its hook only while it lives)
            for n in followers:
                self.processor.process(named[n])
        if errors:
            raise errors[0]

    @torch.no_grad()
    def loop(self, layer_inputs: List[torch.Tensor], layer_kwargs: Optional[List[dict]] = None,
             forward: Optional[Callable] = None, progress: Optional[Callable] = None):
        """layer_inputs: hidden states entering layer 0, one tensor per calibration batch ([b, seq, hidden]);
        layer_kwargs: per-batch keyword arguments of the layer forward (attention mask, position ids, ...).
        Returns the hidden states leaving the last layer."""
        layer_kwargs = layer_kwargs or [{} for _ in layer_inputs]
        fwd = forward or (lambda layer, x, kw: layer(x, **kw))
        dist = gdist.Dist.current()
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
                owners = gdist.assign({n: (nm.state["out_features"], nm.state["in_features"])
                                       for n, nm in named.items()}, dist.world)
                mine = [n for n in named if owners[n] == dist.rank]
                handles = []
                leader = None  # the modules of a group see the same inputs: one Hessian / prologue for all of them
                for n in mine:
                    self.processor.preprocess(named[n], buffered_fwd=False)
                    if self.processor.is_skipped(named[n]):
                        continue
                    task = self.processor.tasks[n]
                    # a follower takes the leader's Hessian AND prologue: only when every setting the prologue depends on
                    # is the same (a `dynamic` override of damp_percent / act_sort for one module makes it its own leader)
                    if (self.share_group_hessian and leader is not None
                            and self.processor.tasks[leader]._prologue_key() == task._prologue_key()):
                        task.follow(self.processor.tasks[leader])
                        continue
                    if leader is None:
                        leader = n
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
                    handles += [mods[n].register_forward_hook(stop_hook) for n in mine
                                if not self.processor.is_skipped(named[n])
                                and getattr(self.processor.tasks[n], "_group_leader", None) is None]
                for bi, (x, kw) in enumerate(zip(layer_inputs, layer_kwargs)):
                    fired.clear()
                    cur["batch"] = bi
                    cur["pass"] += 1
                    try:
                        fwd(layer, x, kw)
                    except _StopForward:
                        pass
                for h in handles:
                    h.remove()
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
                self._process_group(todo, named)
                for n in todo:
                    if self.cache_outputs:
                        cache_module(mods[n])
                    if progress:
                        progress(named[n])
                if dist.world > 1:
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

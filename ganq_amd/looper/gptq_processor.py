"""LoopProcessor plugin for GANQ on the HIP path -- the caller-side contract of
gptqmodel/looper/gptq_processor.py:68-227 (preprocess / is_skipped / preprocess_fwd_hook / process /
submodule_finalize / finalize / results), reduced to what the path needs.

Differences from the reference processor:
  * `process` also stores the assignment indices and codebook (`ganq_q`, `ganq_lut`) next to scale/zero/g_idx in
    the per-module result, because the LUT layer is built from them (the reference result only carries
    {"scale","zero","g_idx"}, gptq_processor.py:172-176);
  * quantized weights stay on the GPU (the reference moves every `wq` to the CPU, :196-199);
  * `finalize` packs through ganq_amd/nn_modules/backend.py, the reference's pack_model -> make_quant ->
    select_quant_linear -> create_quant_layer path (utils/model.py:573-639) with BACKEND.GANQ_HIP registered.
"""
import copy
import time
from typing import Callable, Dict, Tuple

import torch
import torch.nn as nn

from ..quantization.config import FORMAT, QUANT_METHOD, QuantizeConfig
from ..quantization.ganq import GANQ
from .named_module import NamedModule


class GPTQProcessor:
    def __init__(self, qcfg: QuantizeConfig, logger_board: str = ""):
        if qcfg.quant_method != QUANT_METHOD.GANQ:
            raise ValueError(f"this processor implements quant_method=`ganq` only, got `{qcfg.quant_method}`")
        self.qcfg = qcfg
        self.tasks: Dict[str, GANQ] = {}
        self._results: Dict[str, Dict[str, torch.Tensor]] = {}
        self.log = []  # one row per module, like the reference's quant_log (writer.py:54-60)
        self.unquantized = []  # modules that saw no calibration data (MoE experts) and were left as they are

    def preprocess(self, module: NamedModule, buffered_fwd: bool = False):
        if self.qcfg.dynamic_get(layer_name=module.full_name) is False:  # '-:' pattern: module skipped
            return
        qcfg_clone = copy.deepcopy(self.qcfg)
        if self.qcfg.dynamic is not None:
            for key in ("bits", "sym", "mse", "group_size", "desc_act", "damp_percent", "static_groups"):
                setattr(qcfg_clone, key, self.qcfg.dynamic_get(module.full_name, key, getattr(qcfg_clone, key)))
        tmp = GANQ(module=module, qcfg=qcfg_clone)
        if buffered_fwd:
            tmp.fwd_inputs_buffered = True
        tmp.quantizer.configure(perchannel=True)
        self.tasks[module.name] = tmp

    def skip(self, module: NamedModule):
        """Drop a prepared module without quantizing it (it saw no calibration data)."""
        g = self.tasks.pop(module.name, None)
        if g:
            g.free()
        self.tasks[module.name] = False
        self.unquantized.append(module.full_name)

    def is_skipped(self, module: NamedModule) -> bool:
        return self.tasks.get(module.name, False) is False

    def preprocess_fwd_hook(self, name: str) -> Callable[[nn.Module, Tuple[torch.Tensor, ...], torch.Tensor], None]:
        def hook(_, inp: Tuple[torch.Tensor, ...], out: torch.Tensor):
            self.tasks[name].add_batch(inp[0].data, out.data if out is not None else None)

        return hook

    def process(self, module: NamedModule):
        g = self.tasks[module.name]
        t0 = time.time()
        wq, scale, zero, g_idx, duration, avg_loss, damp_percent = g.quantize()
        self.log.append({"layer": module.layer_index, "module": module.name, "loss": f"{avg_loss:.5f}",
                         "damp": f"{damp_percent:.5f}", "time": f"{duration:.3f}"})
        self._results[module.full_name] = {
            "scale": scale, "zero": zero, "g_idx": g_idx,
            "ganq_q": g.ganq_indices, "ganq_lut": g.ganq_codebook, "bits": g.qcfg.bits,
            "ganq_outliers": getattr(g, "ganq_outliers", None), "avg_loss": avg_loss,
            # a convolution keeps its dequantised weight (there is no LUT layer for it): finalize() does not pack it
            "packable": not isinstance(module.module, nn.Conv2d),
        }
        module.state.update({"wq": wq, "ganq_q": g.ganq_indices, "ganq_lut": g.ganq_codebook,
                             "quant_time": time.time() - t0, "avg_loss": avg_loss,
                             "ganq_stats": {k: v for k, v in getattr(g, "ganq_stats", {}).items() if isinstance(v, float)}})
        g.free()
        del self.tasks[module.name]
        module.weight.data = wq  # the next modules / layers are calibrated on the quantized weight

    def submodule_finalize(self, module: NamedModule):
        pass  # results stay on the device

    def results(self):
        return self._results

    def finalize(self, model: nn.Module, backend=None, **kwargs):
        """FORMAT.GANQ_LUT: replace every quantized nn.Linear by a packed QuantLinear, through the same steps as the
        reference's pack path (pack_model -> make_quant -> select_quant_linear -> create_quant_layer -> pack_module,
        utils/model.py:573-639): ganq_amd/nn_modules/backend.py.  FORMAT.FAKE: the dequantised weights already sit in
        the nn.Linear modules."""
        from ..nn_modules.backend import BACKEND, pack_model

        packable = {k: v for k, v in self._results.items() if v.get("packable", True)}
        pack_model(model, quant_result=packable, qcfg=self.qcfg, backend=backend or BACKEND.AUTO)
        return model

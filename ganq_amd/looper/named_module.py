"""NamedModule -- the wrapper the looper hands to quantizers (gptqmodel/looper/named_module.py:24-76)."""
from typing import Any

import torch
from torch import nn


class NamedModule(torch.nn.Module):
    def __init__(self, module: torch.nn.Module, name: str, full_name: str, layer_index: int) -> None:
        super().__init__()
        self.module = module
        self.name = name
        self.full_name = full_name
        self.layer_index = layer_index
        self.state = {}  # per-module work state of the LoopProcessors ("wq", "ganq_q", "ganq_lut", ...)
        if isinstance(module, nn.Linear):
            in_features, out_features = module.in_features, module.out_features
        elif type(module).__name__ == "Conv1D":  # transformers.pytorch_utils.Conv1D
            in_features, out_features = module.weight.shape[0], module.weight.shape[1]
        elif isinstance(module, nn.Conv2d):  # named_module.py:44-46 of the reference
            in_features, out_features = module.in_channels, module.out_channels
        else:
            raise NotImplementedError(f"Unsupported module.module type: `{type(module)}`")
        self.state.update({"in_features": in_features, "out_features": out_features})

    def __getattr__(self, name: str):
        return getattr(self.module, name)

    def __setattr__(self, name: str, value: Any) -> None:
        if name in ["module", "name", "full_name", "layer_index", "state"]:
            self.__dict__[name] = value
        else:
            self.module.__dict__[name] = value

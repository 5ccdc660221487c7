"""QuantizeConfig for the GANQ path -- same field names, defaults and validation as the reference
(gptqmodel/quantization/config.py:157-289) for every field the path reads; fields of unrelated
methods (adapters, rotation, marlin, ...) are not carried.

One addition: FORMAT.GANQ_LUT.  The reference saves GANQ output as FORMAT.FAKE (dequantised fp16 weights,
config.py:75,103-105); the LUT layer needs the indices and codebooks, so QUANT_METHOD.GANQ maps to
{FAKE, GANQ_LUT} here.
"""
import copy
import re
from dataclasses import dataclass, field, fields
from typing import Any, Dict, Optional, Union

import torch


class FORMAT:
    FAKE = "fake"          # reference: dequantised fp16 weights in a FakeQuantLinear
    GANQ_LUT = "ganq_lut"  # packed indices + per-row codebook, GanqHipQuantLinear


class QUANT_METHOD:
    GANQ = "ganq"


QUANT_METHOD_FORMAT_MAPPING = {QUANT_METHOD.GANQ: {FORMAT.FAKE, FORMAT.GANQ_LUT}}


def dynamic_get(dynamic, module_name, key=None, default=None, sub_key=None):
    """regex overrides per module (config.py:131-154): '-:pattern' excludes, '+:pattern' or bare pattern includes"""
    if dynamic is None:
        return default
    for pattern, overrides in dynamic.items():
        if pattern.startswith("-:"):
            if re.match(pattern.removeprefix("-:"), module_name):
                return False
        elif re.match(pattern.removeprefix("+:"), module_name):
            if key is None:
                return overrides
            if sub_key:
                sub = overrides.get(key, None)
                return sub.get(sub_key, default) if isinstance(sub, dict) else default
            return overrides.get(key, default)
    return default


@dataclass
class QuantizeConfig:
    bits: int = field(default=4, metadata={"choices": [2, 3, 4, 8]})
    dynamic: Optional[Dict[str, Dict[str, Union[int, bool]]]] = field(default=None)
    group_size: int = field(default=128)  # ignored by GANQ except for the compat g_idx (gptq.py:332-339)
    damp_percent: float = field(default=0.01)
    damp_auto_increment: float = field(default=0.0025)
    l_damp_style: str = field(default="gptq", metadata={"choices": ["gptq", "ganq"]})
    dead: str = field(default="zero", metadata={"choices": ["zero", "mean"]})
    desc_act: bool = field(default=True)
    act_sort: str = field(default="auto", metadata={"choices": ["auto", "none", "desc", "asc"]})
    static_groups: bool = field(default=False)
    sym: bool = field(default=True)
    true_sequential: bool = field(default=True)
    lm_head: bool = field(default=False)
    quant_method: str = field(default=QUANT_METHOD.GANQ)
    format: str = field(default=FORMAT.GANQ_LUT)
    mse: float = field(default=0.0)
    meta: Optional[Dict] = field(default=None)
    device: Optional[Union[str, torch.device]] = field(default=None)
    pack_dtype: Optional[Union[str, torch.dtype]] = field(default=torch.int32)
    ganq_iterations: int = field(default=5)
    # not in the reference: False pairs the best codebook with its OWN indices instead of reproducing the
    # torch-branch aliasing of ganq.py:487,550,625-626 (see DESIGN.md "reference quirks")
    ganq_reference_q_alias: bool = field(default=True)
    # not in the reference code (paper section 3.3 / Appendix A, "GANQ*"): fraction r of every weight row kept exactly
    # as sparse fp16 outliers (row-wise, beyond the 1 - r/2 and r/2 quantiles); GANQ quantizes the rest.  0 = off.
    ganq_outlier_ratio: float = field(default=0.0)
    # not in the reference: calibration batches are handed to the Hessian kernel in groups of up to this many tokens
    # (one read-modify-write of H per group instead of per batch; same H up to fp32 summation order).  0 = per batch.
    # Memory: one buffer of this many tokens x in_features activations per hooked module while its group's calibration
    # passes run (134 MB at in_features 4096, 470 MB at 14336; at most quantization.gptq.STAGE_MAX_LIVE per device, freed
    # when the passes end).
    ganq_hessian_stage_tokens: int = field(default=16384)

    def __post_init__(self):
        info = fields(self)
        if self.pack_dtype is None:
            self.pack_dtype = torch.int32
        elif isinstance(self.pack_dtype, str):
            name = self.pack_dtype.lower()
            if name not in ["int64", "int32", "int16", "int8"]:
                raise ValueError(f"QuantizeConfig: Unsupported `pack_dtype`: {self.pack_dtype}")
            self.pack_dtype = getattr(torch, name)
        elif self.pack_dtype not in [torch.int64, torch.int32, torch.int16, torch.int8]:
            raise ValueError(f"QuantizeConfig: Unsupported `pack_dtype`: {self.pack_dtype}")
        valid = QUANT_METHOD_FORMAT_MAPPING.get(self.quant_method, None)
        if valid is None:
            raise ValueError(f"QuantizeConfig: Unsupported `quant_method`: {self.quant_method}")
        if self.format not in valid:
            raise ValueError(f"QuantizeConfig: checkpoint `format` used is {self.format}, and the quantization method "
                             f"is {self.quant_method}. ")
        if self.bits not in info[0].metadata["choices"]:
            raise ValueError(f"QuantizeConfig: `bits` must be in the set of `{info[0].metadata['choices']}`.")
        if not (0.0 <= self.ganq_outlier_ratio < 1.0):
            raise ValueError("QuantizeConfig: `ganq_outlier_ratio` must be in [0, 1)")
        if self.dynamic is not None:
            self.dynamic = {**{k: v for k, v in self.dynamic.items() if k.startswith("-")},
                            **{k: v for k, v in self.dynamic.items() if not k.startswith("-")}}
            for layer, layer_dict in self.dynamic.items():
                for key, value in layer_dict.items():
                    if key == "bits" and value not in info[0].metadata["choices"]:
                        raise ValueError(f"QuantizeConfig: Layer `{layer}` only support quantization of "
                                         f"`{info[0].metadata['choices']}` bits.")
                    if key == "group_size" and value != -1 and value <= 0:
                        raise ValueError("QuantizeConfig: `group_size` must in the value set of `[-1, 16, 32, 64, 128]`.")
        if self.group_size != -1 and self.group_size <= 0:
            raise ValueError("QuantizeConfig: `group_size` must in the value set of `[-1, 16, 32, 64, 128]`.")
        if not (0 < self.damp_percent < 1):
            raise ValueError("QuantizeConfig: `damp_percent` must between 0 and 1.")
        if self.damp_auto_increment < 0:
            raise ValueError("QuantizeConfig:: `damp_auto_increment` must greater than 0.")
        if self.act_sort == "auto":
            self.act_sort = "desc" if self.desc_act else "none"
        if self.act_sort not in ["none", "desc", "asc"]:
            raise ValueError(f"QuantizeConfig: unknown `act_sort`: {self.act_sort}")
        if self.l_damp_style not in ["gptq", "ganq"]:
            raise ValueError(f"QuantizeConfig: unknown `l_damp_style`: {self.l_damp_style}")
        if self.dead not in ["zero", "mean"]:
            raise ValueError(f"QuantizeConfig: unknown `dead`: {self.dead}")
        if self.meta is None:
            self.meta = {}
        elif not isinstance(self.meta, dict) or not all(isinstance(k, str) for k in self.meta):
            raise ValueError("QuantizeConfig: `meta` must be a dictionary with string keys")

    def dynamic_get(self, layer_name: str, key: str = None, default: Any = None, sub_key: str = None):
        return dynamic_get(self.dynamic, layer_name, key, default, sub_key)

    def clone(self) -> "QuantizeConfig":
        return copy.deepcopy(self)

    def to_dict(self) -> Dict[str, Any]:
        out = {f.name: getattr(self, f.name) for f in fields(self) if f.name not in ("device",)}
        out["pack_dtype"] = str(self.pack_dtype).split(".")[-1]
        return out

    @classmethod
    def from_dict(cls, d: Dict[str, Any]) -> "QuantizeConfig":
        names = {f.name for f in fields(cls)}
        return cls(**{k: v for k, v in d.items() if k in names})

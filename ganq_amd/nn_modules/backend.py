"""Backend registration for the LUT layer -- the machinery a QuantLinear plugs into in the reference, as code:
`BACKEND` (utils/backend.py:20-42), `AUTO_SELECT_BACKEND_ORDER` / `FORMAT_DICT` / `select_quant_linear`
(utils/importer.py:45-68,157-262), `make_quant` / `create_quant_layer` / `pack_model` (utils/model.py:150-370,552-639),
reduced to the one format this path adds (FORMAT.GANQ_LUT -> BACKEND.GANQ_HIP -> GanqHipQuantLinear) next to the
reference's FORMAT.FAKE (dequantised weights stay in the nn.Linear modules; nothing to pack).

What differs from the reference's pack path, and why: a LUT layer cannot be rebuilt from the dequantised `linear` alone, so
`quant_result[name]` carries `ganq_q` / `ganq_lut` (/ `ganq_outliers`) beside scale / zero / g_idx and `pack_module` hands
them to `pack()`; packing runs on the device the module lives on (the reference moves every
module to the CPU first, utils/model.py:557-563 -- its packers are CPU loops, this one is a HIP kernel).
"""
from collections import OrderedDict
from enum import Enum
from typing import Any, Dict, List, Optional, Type, Union

import torch
import torch.nn as nn

from ..quantization.config import FORMAT
from .qlinear import BaseQuantLinear
from .qlinear.ganq_hip import GanqHipQuantLinear


class BACKEND(str, Enum):
    AUTO = "auto"          # the best kernel the format and the settings allow
    GANQ_HIP = "ganq_hip"  # LUT-dequant linear on the MI355X (libganq_hip.so)
    FAKE = "fake"          # reference: dequantised fp16 weights, no kernel


AUTO_SELECT_BACKEND_ORDER = OrderedDict({
    BACKEND.GANQ_HIP: GanqHipQuantLinear,
})

FORMAT_DICT = {
    FORMAT.GANQ_LUT: [BACKEND.GANQ_HIP],
    FORMAT.FAKE: [],  # the quantized weights already sit in the nn.Linear modules
}


def select_quant_linear(bits: int, group_size: int, desc_act: bool, sym: bool, device=None, backend: BACKEND = BACKEND.AUTO,
                        format: str = FORMAT.GANQ_LUT, pack: bool = False, dynamic=None, pack_dtype: torch.dtype = None,
                        multi_select: bool = False, adapter=None
                        ) -> Union[Type[BaseQuantLinear], List[Type[BaseQuantLinear]]]:
    """utils/importer.py:157-262: the QuantLinear class (or, with multi_select, every valid one in preference order)"""
    backend = BACKEND.AUTO if backend is None else BACKEND(backend)
    pack_dtype = torch.int32 if pack_dtype is None else pack_dtype
    if format not in FORMAT_DICT:
        raise ValueError(f"select_quant_linear: unknown format `{format}`")
    if backend == BACKEND.AUTO:
        valid, err = [], None
        for k, cls in AUTO_SELECT_BACKEND_ORDER.items():
            if k not in FORMAT_DICT[format]:
                continue
            ok, err = cls.validate(bits=bits, group_size=group_size, desc_act=desc_act, sym=sym, pack_dtype=pack_dtype,
                                   dynamic=dynamic, device=device, trainable=False, adapter=adapter)
            if ok:
                valid.append(cls)
                if not multi_select:
                    return cls
        if not valid:
            raise err if err else NotImplementedError(f"no QuantLinear serves format `{format}`")
        return valid
    if backend not in FORMAT_DICT[format]:
        raise ValueError(f"select_quant_linear: backend `{backend.value}` does not serve format `{format}`")
    cls = AUTO_SELECT_BACKEND_ORDER[backend]
    ok, err = cls.validate(bits=bits, group_size=group_size, desc_act=desc_act, sym=sym, pack_dtype=pack_dtype, dynamic=dynamic,
                           device=device, trainable=False, adapter=adapter)
    if not ok:
        raise err
    return [cls] if multi_select else cls


def _features(lin: nn.Module):
    if type(lin).__name__ == "Conv1D":  # transformers.pytorch_utils.Conv1D: weight is [in, out]
        return lin.weight.shape[0], lin.weight.shape[1]
    return lin.in_features, lin.out_features


def create_quant_layer(linear_cls: Type[BaseQuantLinear], bits: int, desc_act: bool, dynamic, group_size: int, module: nn.Module,
                       quant_result: Dict[str, Dict[str, Any]], sym: bool, device, lm_head_name: Optional[str],
                       pack_dtype: torch.dtype, backend: BACKEND, adapter=None) -> Type[BaseQuantLinear]:
    """utils/model.py:282-370: replace every module named in quant_result by an (empty) linear_cls instance"""
    named = dict(module.named_modules())
    # validate EVERY module before swapping any: a NotImplementedError (the caller's cue to try the next candidate class,
    # utils/model.py:234-239) must leave the model untouched
    todo = []
    for name, res in quant_result.items():
        lin = named[name]
        if isinstance(lin, linear_cls):
            continue
        in_f, out_f = _features(lin)
        ok, err = linear_cls.validate(bits=res.get("bits", bits), group_size=group_size, desc_act=desc_act, sym=sym,
                                      in_features=in_f, out_features=out_f, pack_dtype=pack_dtype, device=device)
        if not ok:
            raise err
        todo.append((name, res, lin, in_f, out_f))
    swapped = []
    try:
        for name, res, lin, in_f, out_f in todo:
            nnz = 0 if res.get("ganq_outliers") is None else int(res["ganq_outliers"][1].numel())
            new = linear_cls(bits=res.get("bits", bits), group_size=group_size, desc_act=desc_act, sym=sym, in_features=in_f,
                             out_features=out_f, pack_dtype=pack_dtype, bias=lin.bias is not None, name=name,
                             lm_head_name=lm_head_name, backend=backend, adapter=adapter, outliers=nnz).to(lin.weight.device)
            # the quantized nn.Linear, until pack_module has consumed it.  Not through nn.Module.__setattr__: that would register
            # it as a submodule of the new layer (named_modules / state_dict / parameters would carry the fp weights twice)
            object.__setattr__(new, "_packed_from", lin)
            parent, _, child = name.rpartition(".")
            owner = named[parent] if parent else module
            setattr(owner, child, new)
            swapped.append((owner, child, lin))
    except BaseException:
        for owner, child, lin in swapped:  # a constructor failed midway: put the original modules back
            setattr(owner, child, lin)
        raise
    return linear_cls


def make_quant(module: nn.Module, quant_result, qcfg, backend: BACKEND, lm_head_name: Optional[str] = None, pack: bool = False,
               device=None) -> Optional[Type[BaseQuantLinear]]:
    """utils/model.py:150-246: pick the kernel class for the config and swap the modules in"""
    if not FORMAT_DICT.get(qcfg.format):
        return None  # FORMAT.FAKE: nothing to build
    candidates = select_quant_linear(bits=qcfg.bits, group_size=qcfg.group_size, desc_act=qcfg.desc_act, sym=qcfg.sym,
                                     backend=backend, format=qcfg.format, pack=pack, dynamic=qcfg.dynamic, device=device,
                                     pack_dtype=qcfg.pack_dtype, multi_select=True)
    for cls in candidates:
        try:
            return create_quant_layer(linear_cls=cls, bits=qcfg.bits, desc_act=qcfg.desc_act, dynamic=qcfg.dynamic,
                                      group_size=qcfg.group_size, module=module, quant_result=quant_result, sym=qcfg.sym,
                                      device=device, lm_head_name=lm_head_name, pack_dtype=qcfg.pack_dtype, backend=backend)
        except NotImplementedError:
            if BACKEND(backend) != BACKEND.AUTO:
                raise
    raise ValueError(f"No compatible quant linear was found for this module: {module.__class__.__name__}")


def pack_module(name: str, qmodules: Dict[str, BaseQuantLinear], quant_result: Dict[str, Dict[str, Any]]):
    """utils/model.py:552-570, on the module's own device, with the LUT layer's extra inputs"""
    r = quant_result[name]
    q = qmodules[name]
    q.pack(q._packed_from, r["scale"], r["zero"], r["g_idx"], ganq_indices=r["ganq_q"], ganq_codebook=r["ganq_lut"],
           ganq_outliers=r.get("ganq_outliers"))
    object.__delattr__(q, "_packed_from")


def pack_model(model: nn.Module, quant_result: Dict[str, Dict[str, Any]], qcfg, backend: BACKEND = BACKEND.AUTO,
               lm_head_name: Optional[str] = None) -> Optional[Type[BaseQuantLinear]]:
    """utils/model.py:573-639: nn.Linear -> packed QuantLinear for every module of quant_result; returns the kernel class
    (None for FORMAT.FAKE)"""
    dev = next(model.parameters()).device
    cls = make_quant(model, quant_result=quant_result, qcfg=qcfg, backend=backend, lm_head_name=lm_head_name, pack=True,
                     device=dev)
    if cls is None:
        return None
    qmodules = {n: m for n, m in model.named_modules() if isinstance(m, cls) and n in quant_result}
    assert len(qmodules) == len(quant_result), f"No quantized modules[{cls}] found in the model."
    for name in qmodules:
        pack_module(name, qmodules, quant_result)
    return cls


__all__ = ["BACKEND", "AUTO_SELECT_BACKEND_ORDER", "FORMAT_DICT", "select_quant_linear", "create_quant_layer", "make_quant",
           "pack_module", "pack_model"]

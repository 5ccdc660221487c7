"""quantize_model / save_quantized / load_quantized / gptq_style_ppl.

quantize_model  == BaseGPTQModel.quantize for quant_method=GANQ (models/base.py:317-459): capture the inputs of
                   decoder layer 0 for every calibration batch (module_looper.py:44-127 `cache_inputs`), then run the
                   device-resident looper over the layers.
save/load       the packed format the reference lacks: it can only save FORMAT.FAKE (dequantised fp16) and refuses to
                   load even that (config.py:369,398-401).  Here: one safetensors file with, per quantized Linear,
                   `<name>.qweight` int32 [in*bits/32, out], `<name>.lut` fp16 [out, 2^bits], `<name>.bias`; every other
                   tensor of the model unchanged; `quantize_config.json` next to it with the per-module bit widths.
gptq_style_ppl  the evaluator behind the README numbers (examples/quantization/basic_usage_wikitext2.py:63-93):
                   non-overlapping windows of `seqlen` tokens, mean token NLL, exp.
"""
import json
import os
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn as nn

from ..looper.gptq_processor import GPTQProcessor
from ..looper.module_looper import ModuleLooper
from ..nn_modules.qlinear.ganq_hip import GanqHipQuantLinear
from ..quantization.config import FORMAT, QuantizeConfig
from .definitions import LayerMap, layer_map_for


def _get_module(root: nn.Module, path: str) -> nn.Module:
    mod = root
    for part in path.split("."):
        mod = getattr(mod, part)
    return mod


class _StopForward(Exception):
    pass


@torch.no_grad()
def capture_layer_inputs(model: nn.Module, layers: Sequence[nn.Module], batches: Sequence[torch.Tensor]):
    """run the model up to decoder layer 0 for every batch and keep that layer's positional/keyword inputs"""
    hidden, kwargs_list = [], []

    def hook(_, args, kwargs):
        hidden.append(args[0] if args else kwargs["hidden_states"])
        kw = {k: v for k, v in kwargs.items() if k not in ("hidden_states", "past_key_values", "past_key_value", "use_cache")}
        kwargs_list.append(kw)
        raise _StopForward()

    handle = layers[0].register_forward_pre_hook(hook, with_kwargs=True)
    try:
        for ids in batches:
            try:
                model(ids, use_cache=False)
            except _StopForward:
                pass
    finally:
        handle.remove()
    return hidden, kwargs_list


@torch.no_grad()
def quantize_model(model: nn.Module, calibration: Sequence[torch.Tensor], qcfg: QuantizeConfig,
                   layer_map: Optional[LayerMap] = None, progress=None, processor: Optional[GPTQProcessor] = None,
                   **looper_options) -> GPTQProcessor:
    """Quantize every Linear the layer map names, layer by layer, on the device the model lives on.  `calibration`:
    token-id tensors [b, seq].  Returns the processor (per-module log, results); the model is modified in place
    (GanqHipQuantLinear modules for FORMAT.GANQ_LUT, dequantised nn.Linear weights for FORMAT.FAKE).
    looper_options: `early_exit=` / `cache_outputs=` / `share_group_hessian=` of ModuleLooper (results do not depend on
    them).  processor: a GPTQProcessor (subclass) to drive instead of a fresh one -- the reference's LoopProcessor slot."""
    lm = layer_map or layer_map_for(model)
    layers = _get_module(model, lm.layers_node)
    dev = next(model.parameters()).device
    from .. import distributed as gdist

    dist = gdist.Dist.current()
    # several ranks, dist_mode="rows" (the default): calibration is data-parallel -- this rank embeds and forwards only the
    # batches rank, rank + world, ...
    local = dist.world > 1 and looper_options.get("dist_mode", "rows") == "rows"
    batches = [b.to(dev) for b in (calibration[dist.rank::dist.world] if local else calibration)]
    was_training = model.training
    model.eval()
    hidden, kwargs_list = capture_layer_inputs(model, layers, batches)
    proc = processor if processor is not None else GPTQProcessor(qcfg)
    looper_options.setdefault("share_group_hessian", lm.shared_group_inputs)

    def fwd(layer, x, kw):
        return layer(x, **kw)

    ModuleLooper(proc, layers, lm.layer_modules, layers_prefix=lm.layers_node, **looper_options).loop(
        hidden, kwargs_list, forward=fwd, progress=progress, inputs_are_local=local)
    proc.finalize(model)
    model.quantize_config = qcfg
    model.train(was_training)
    return proc


def save_quantized(model: nn.Module, path: str, qcfg: Optional[QuantizeConfig] = None) -> None:
    from safetensors.torch import save_file

    qcfg = qcfg or getattr(model, "quantize_config", None)
    if qcfg is None:
        raise ValueError("save_quantized: no QuantizeConfig given and the model carries none")
    os.makedirs(path, exist_ok=True)
    modules = {n: {"bits": m.bits, "in_features": m.in_features, "out_features": m.out_features,
                   "bias": m.bias is not None, "outliers": int(getattr(m, "outliers", 0))}
               for n, m in model.named_modules() if isinstance(m, GanqHipQuantLinear)}
    state = {k: v.detach().contiguous().cpu() for k, v in model.state_dict().items()}
    # tied weights (lm_head <-> embeddings) share storage: safetensors wants each tensor once
    seen, out = {}, {}
    for k, v in state.items():
        key = (v.data_ptr(), tuple(v.shape), v.dtype)
        if key in seen and v.numel() > 0:
            continue
        seen[key] = k
        out[k] = v
    save_file(out, os.path.join(path, "model.safetensors"), metadata={"format": "pt", "ganq_format": FORMAT.GANQ_LUT})
    cfg = qcfg.to_dict()
    cfg["modules"] = modules
    with open(os.path.join(path, "quantize_config.json"), "w") as f:
        json.dump(cfg, f, indent=2, default=str)


def load_quantized(model: nn.Module, path: str) -> nn.Module:
    """`model`: a freshly constructed (unquantized, any weights) instance of the same architecture.  Replaces the
    recorded Linear modules by GanqHipQuantLinear, then loads every tensor of the checkpoint."""
    from safetensors.torch import load_file

    with open(os.path.join(path, "quantize_config.json")) as f:
        cfg = json.load(f)
    modules: Dict[str, dict] = cfg.pop("modules")
    qcfg = QuantizeConfig.from_dict(cfg)
    named = dict(model.named_modules())
    for name, info in modules.items():
        lin = named[name]
        q = GanqHipQuantLinear(bits=info["bits"], group_size=qcfg.group_size, sym=qcfg.sym, desc_act=qcfg.desc_act,
                               in_features=info["in_features"], out_features=info["out_features"], bias=info["bias"],
                               pack_dtype=torch.int32, name=name, outliers=info.get("outliers", 0)).to(lin.weight.device)
        parent, _, child = name.rpartition(".")
        setattr(named[parent] if parent else model, child, q)
    state = load_file(os.path.join(path, "model.safetensors"))
    missing, unexpected = model.load_state_dict(state, strict=False)
    missing = [k for k in missing if k not in ("lm_head.weight",)]  # tied to the embeddings
    if missing or unexpected:
        raise RuntimeError(f"load_quantized: missing {missing[:5]} unexpected {list(unexpected)[:5]}")
    if hasattr(model, "tie_weights"):
        model.tie_weights()
    model.quantize_config = qcfg
    return model


@torch.no_grad()
def gptq_style_ppl(model: nn.Module, token_ids: torch.Tensor, seqlen: int = 2048) -> float:
    """token_ids: [1, total] (e.g. the wikitext-2-raw-v1 test split joined with "\\n\\n" and tokenized)."""
    dev = next(model.parameters()).device
    ids = token_ids.to(dev)
    nsamples = ids.numel() // seqlen
    if nsamples == 0:
        raise ValueError("gptq_style_ppl: fewer tokens than one window")
    model.eval()
    nlls = []
    loss_fct = nn.CrossEntropyLoss()
    for i in range(nsamples):
        batch = ids[:, i * seqlen:(i + 1) * seqlen]
        logits = model(batch).logits
        shift_logits = logits[:, :-1, :].contiguous().float()
        shift_labels = batch[:, 1:]
        loss = loss_fct(shift_logits.view(-1, shift_logits.size(-1)), shift_labels.reshape(-1))
        nlls.append(loss.float() * seqlen)
    return float(torch.exp(torch.stack(nlls).sum() / (nsamples * seqlen)))

"""Per-architecture layer maps: where the repeating decoder layers live and which Linear modules of a layer are
quantized in which order (modules of one group see the same inputs).  Same tables as the reference:
  OPT    gptqmodel/models/definitions/opt.py:34-41
  Llama  gptqmodel/models/definitions/llama.py:28-39
"""
from dataclasses import dataclass
from typing import List


@dataclass(frozen=True)
class LayerMap:
    layers_node: str
    layer_modules: List[List[str]]


LAYER_MAPS = {
    "opt": LayerMap("model.decoder.layers", [
        ["self_attn.k_proj", "self_attn.v_proj", "self_attn.q_proj"],
        ["self_attn.out_proj"],
        ["fc1"],
        ["fc2"],
    ]),
    "llama": LayerMap("model.layers", [
        ["self_attn.k_proj", "self_attn.v_proj", "self_attn.q_proj"],
        ["self_attn.o_proj"],
        ["mlp.up_proj", "mlp.gate_proj"],
        ["mlp.down_proj"],
    ]),
}


def layer_map_for(model) -> LayerMap:
    mt = getattr(getattr(model, "config", None), "model_type", None)
    if mt not in LAYER_MAPS:
        raise NotImplementedError(f"no layer map for model_type `{mt}` (have: {sorted(LAYER_MAPS)})")
    return LAYER_MAPS[mt]

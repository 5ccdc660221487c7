"""Per-architecture layer maps: where the repeating decoder layers live and which Linear modules of a layer are
quantized in which order (modules of one group see the same inputs).  Same tables as the reference's
gptqmodel/models/definitions/*.py (opt.py:34-41, llama.py:28-39, ...), for the dense decoder families whose
Hugging Face module names are listed below.  Mixture-of-experts families are not mapped: this transformers
version fuses the experts into 3-D parameters instead of per-expert nn.Linear modules.  (The looper itself handles
non-shared groups -- `shared_group_inputs=False` -- and reports modules no calibration token reached, as the
reference's does.)
"""
from dataclasses import dataclass
from typing import List


@dataclass(frozen=True)
class LayerMap:
    layers_node: str
    layer_modules: List[List[str]]
    shared_group_inputs: bool = True  # the modules of a group receive the same tensor (dense decoders); False for MoE experts


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
    "phi3": LayerMap("model.layers", [
        ["self_attn.qkv_proj"],
        ["self_attn.o_proj"],
        ["mlp.gate_up_proj"],
        ["mlp.down_proj"],
    ]),
    "starcoder2": LayerMap("model.layers", [
        ["self_attn.k_proj", "self_attn.v_proj", "self_attn.q_proj"],
        ["self_attn.o_proj"],
        ["mlp.c_fc"],
        ["mlp.c_proj"],
    ]),
    "gpt_neox": LayerMap("gpt_neox.layers", [
        ["attention.query_key_value"],
        ["attention.dense"],
        ["mlp.dense_h_to_4h"],
        ["mlp.dense_4h_to_h"],
    ]),
    "gptj": LayerMap("transformer.h", [
        ["attn.k_proj", "attn.v_proj", "attn.q_proj"],
        ["attn.out_proj"],
        ["mlp.fc_in"],
        ["mlp.fc_out"],
    ]),
    "falcon": LayerMap("transformer.h", [
        ["self_attention.query_key_value"],
        ["self_attention.dense"],
        ["mlp.dense_h_to_4h"],
        ["mlp.dense_4h_to_h"],
    ]),
    "bloom": LayerMap("transformer.h", [
        ["self_attention.query_key_value"],
        ["self_attention.dense"],
        ["mlp.dense_h_to_4h"],
        ["mlp.dense_4h_to_h"],
    ]),
    "gpt2": LayerMap("transformer.h", [  # transformers' Conv1D modules (weight stored [in, out])
        ["attn.c_attn"],
        ["attn.c_proj"],
        ["mlp.c_fc"],
        ["mlp.c_proj"],
    ]),
}
# decoder layers with Llama's module names
for _alias in ("mistral", "qwen2", "qwen3", "gemma", "gemma2", "granite", "olmo", "olmo2", "cohere", "stablelm", "yi"):
    LAYER_MAPS[_alias] = LAYER_MAPS["llama"]


def layer_map_for(model) -> LayerMap:
    mt = getattr(getattr(model, "config", None), "model_type", None)
    if mt not in LAYER_MAPS:
        raise NotImplementedError(f"no layer map for model_type `{mt}` (have: {sorted(LAYER_MAPS)})")
    return LAYER_MAPS[mt]

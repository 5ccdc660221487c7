"""GPU tests of the model-level path on random-initialised OPT and Llama architectures (no network: tiny configs built
from transformers' model classes): quantize_model end to end, LUT forward vs dequantised forward, packed checkpoint
round trip, perplexity evaluator."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def tiny(kind):
    import transformers

    torch.manual_seed(0)
    if kind == "opt":
        cfg = transformers.OPTConfig(vocab_size=320, hidden_size=64, ffn_dim=128, num_hidden_layers=2, num_attention_heads=4,
                                     max_position_embeddings=128, word_embed_proj_dim=64)
        return transformers.OPTForCausalLM(cfg).half().cuda().eval()
    cfg = transformers.LlamaConfig(vocab_size=320, hidden_size=64, intermediate_size=128, num_hidden_layers=2,
                                   num_attention_heads=4, num_key_value_heads=2, max_position_embeddings=128)
    return transformers.LlamaForCausalLM(cfg).half().cuda().eval()


FAMILIES = {
    # model_type: (config class, model class, config kwargs, linear layers per decoder layer)
    "mistral": ("MistralConfig", "MistralForCausalLM", dict(vocab_size=320, hidden_size=64, intermediate_size=128,
                num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2, max_position_embeddings=128), 7),
    "qwen2": ("Qwen2Config", "Qwen2ForCausalLM", dict(vocab_size=320, hidden_size=64, intermediate_size=128,
              num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2, max_position_embeddings=128), 7),
    "qwen3": ("Qwen3Config", "Qwen3ForCausalLM", dict(vocab_size=320, hidden_size=64, intermediate_size=128,
              num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2, head_dim=16, max_position_embeddings=128), 7),
    "gemma": ("GemmaConfig", "GemmaForCausalLM", dict(vocab_size=320, hidden_size=64, intermediate_size=128,
              num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2, head_dim=16, max_position_embeddings=128), 7),
    "phi3": ("Phi3Config", "Phi3ForCausalLM", dict(vocab_size=320, hidden_size=64, intermediate_size=128,
             num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=4, max_position_embeddings=128, pad_token_id=0), 4),
    "gpt_neox": ("GPTNeoXConfig", "GPTNeoXForCausalLM", dict(vocab_size=320, hidden_size=64, intermediate_size=128,
                 num_hidden_layers=2, num_attention_heads=4, max_position_embeddings=128), 4),
    "gptj": ("GPTJConfig", "GPTJForCausalLM", dict(vocab_size=320, n_embd=64, n_inner=128, n_layer=2, n_head=4,
             n_positions=128, rotary_dim=16), 6),
    "falcon": ("FalconConfig", "FalconForCausalLM", dict(vocab_size=320, hidden_size=64, num_hidden_layers=2,
               num_attention_heads=4, max_position_embeddings=128), 4),
    "bloom": ("BloomConfig", "BloomForCausalLM", dict(vocab_size=320, hidden_size=64, n_layer=2, n_head=4), 4),
    "gpt2": ("GPT2Config", "GPT2LMHeadModel", dict(vocab_size=320, n_embd=64, n_layer=2, n_head=4, n_positions=128), 4),
    "starcoder2": ("Starcoder2Config", "Starcoder2ForCausalLM", dict(vocab_size=320, hidden_size=64, intermediate_size=128,
                   num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2, max_position_embeddings=128), 6),
}


@pytest.mark.parametrize("kind", sorted(FAMILIES))
@torch.no_grad()
def test_quantize_model_other_families(kind):
    # the layer maps beyond OPT / Llama: every mapped Linear of a tiny random model is quantized and the packed model
    # reproduces the dequantised (FORMAT.FAKE) one
    import copy

    import transformers

    from ganq_amd.models import quantize_model
    from ganq_amd.nn_modules.qlinear.ganq_hip import GanqHipQuantLinear
    from ganq_amd.quantization import QuantizeConfig

    cfg_name, cls_name, kwargs, per_layer = FAMILIES[kind]
    torch.manual_seed(0)
    model = getattr(transformers, cls_name)(getattr(transformers, cfg_name)(**kwargs)).half().cuda().eval()
    g = torch.Generator().manual_seed(1)
    calib = [torch.randint(1, 320, (2, 64), generator=g) for _ in range(3)]
    fake = copy.deepcopy(model)
    qcfg = QuantizeConfig(bits=4, act_sort="asc", l_damp_style="ganq", dead="mean", ganq_iterations=2)
    proc = quantize_model(model, calib, qcfg)
    assert len(proc.results()) == 2 * per_layer
    assert sum(isinstance(mod, GanqHipQuantLinear) for mod in model.modules()) == 2 * per_layer
    # the FORMAT.FAKE run uses the plain looper of the reference (whole layer forward in every pass, nothing cached):
    # the shortcuts of the default looper must not change a single weight
    quantize_model(fake, calib, QuantizeConfig(bits=4, act_sort="asc", l_damp_style="ganq", dead="mean", ganq_iterations=2,
                                               format="fake"), early_exit=False, cache_outputs=False)
    fake_mods = dict(fake.named_modules())
    for name, mod in model.named_modules():
        if isinstance(mod, GanqHipQuantLinear):
            w = fake_mods[name].weight.data
            w = w.t() if type(fake_mods[name]).__name__ == "Conv1D" else w
            assert torch.equal(mod.dequantize_weight(), w), name
    x = calib[0][:1].cuda()
    a, b = model(x).logits.float(), fake(x).logits.float()
    assert torch.isfinite(a).all() and torch.allclose(a, b, rtol=3e-2, atol=3e-2)


@pytest.mark.parametrize("kind,outlier_ratio", [("opt", 0.0), ("llama", 0.0), ("llama", 0.01)])
@torch.no_grad()
def test_quantize_model_save_load_ppl(kind, outlier_ratio, tmp_path):
    import copy

    from ganq_amd.models import gptq_style_ppl, load_quantized, quantize_model, save_quantized
    from ganq_amd.nn_modules.qlinear.ganq_hip import GanqHipQuantLinear
    from ganq_amd.quantization import QuantizeConfig

    model = tiny(kind)
    g = torch.Generator().manual_seed(1)
    calib = [torch.randint(0, 320, (2, 96), generator=g) for _ in range(4)]
    test_ids = torch.randint(0, 320, (1, 64 * 6), generator=g)
    ppl_fp = gptq_style_ppl(model, test_ids, seqlen=64)

    fake = copy.deepcopy(model)
    qcfg = QuantizeConfig(bits=4, act_sort="asc", l_damp_style="ganq", dead="mean", ganq_iterations=3,
                          ganq_outlier_ratio=outlier_ratio)
    proc = quantize_model(model, calib, qcfg)
    n_lin = 2 * (6 if kind == "opt" else 7)
    if outlier_ratio:  # paper section 3.3: every layer keeps its row-wise tails as sparse fp16 values
        assert all(mod.outliers > 0 for mod in model.modules() if isinstance(mod, GanqHipQuantLinear))
    assert len(proc.results()) == n_lin and len(proc.log) == n_lin
    assert sum(isinstance(mod, GanqHipQuantLinear) for mod in model.modules()) == n_lin

    # the same run in the reference's FORMAT.FAKE view (dequantised weights in nn.Linear) must give the same logits
    qcfg_fake = QuantizeConfig(bits=4, act_sort="asc", l_damp_style="ganq", dead="mean", ganq_iterations=3, format="fake",
                               ganq_outlier_ratio=outlier_ratio)
    quantize_model(fake, calib, qcfg_fake)
    x = test_ids[:, :64].cuda()
    a, b = model(x).logits.float(), fake(x).logits.float()
    assert torch.allclose(a, b, rtol=2e-2, atol=2e-2)

    ppl_q = gptq_style_ppl(model, test_ids, seqlen=64)
    assert ppl_q == pytest.approx(gptq_style_ppl(fake, test_ids, seqlen=64), rel=2e-2)
    assert abs(ppl_q - ppl_fp) / ppl_fp < 0.2  # random weights: quantization must not wreck the model

    path = os.path.join(tmp_path, "ckpt")
    save_quantized(model, path)
    fresh = load_quantized(tiny(kind), path)
    assert torch.equal(fresh(x).logits, model(x).logits)
    packed = sum(p.numel() * p.element_size() for n, p in fresh.state_dict().items() if ".qweight" in n or ".lut" in n)
    dense = sum(mod.in_features * mod.out_features * 2 for mod in fresh.modules() if isinstance(mod, GanqHipQuantLinear))
    assert packed < 0.5 * dense

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


@pytest.mark.parametrize("kind", ["opt", "llama"])
@torch.no_grad()
def test_quantize_model_save_load_ppl(kind, tmp_path):
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
    qcfg = QuantizeConfig(bits=4, act_sort="asc", l_damp_style="ganq", dead="mean", ganq_iterations=3)
    proc = quantize_model(model, calib, qcfg)
    n_lin = 2 * (6 if kind == "opt" else 7)
    assert len(proc.results()) == n_lin and len(proc.log) == n_lin
    assert sum(isinstance(mod, GanqHipQuantLinear) for mod in model.modules()) == n_lin

    # the same run in the reference's FORMAT.FAKE view (dequantised weights in nn.Linear) must give the same logits
    qcfg_fake = QuantizeConfig(bits=4, act_sort="asc", l_damp_style="ganq", dead="mean", ganq_iterations=3, format="fake")
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

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


@torch.no_grad()
def test_opt125m_architecture_hip_vs_cpu_oracle_at_model_boundary():
    """BASELINE configs[1] as far as the box allows (no checkpoints, no datasets): the opt-125m ARCHITECTURE at its real
    dimensions (hidden 768, ffn 3072, 12 heads, vocabulary 50272; 2 decoder layers, random initialisation), 4-bit GANQ
    with the example's settings (basic_usage_wikitext2.py:126-141), quantized twice through the same looper: once by the
    HIP path, once by the CPU oracle in the quantizer slot (tests/oracle_quantizer.py: the reference's op sequence on
    the host).  The two quantized models must agree in their logits and in the GPTQ-style perplexity to the metric's
    tolerance: +-0.05 at the reference's 28.45, i.e. 0.18 % relative."""
    import copy

    import transformers

    from ganq_amd.models import gptq_style_ppl, quantize_model
    from ganq_amd.quantization import QuantizeConfig
    from oracle_quantizer import OracleProcessor

    torch.manual_seed(0)
    cfg = transformers.OPTConfig(vocab_size=50272, hidden_size=768, ffn_dim=3072, num_hidden_layers=2,
                                 num_attention_heads=12, max_position_embeddings=2048, word_embed_proj_dim=768)
    base = transformers.OPTForCausalLM(cfg).half().cuda().eval()
    g = torch.Generator().manual_seed(1)
    calib = [torch.randint(0, cfg.vocab_size, (2, 256), generator=g) for _ in range(4)]
    test_ids = torch.randint(0, cfg.vocab_size, (1, 4 * 512), generator=g)

    def qcfg():
        return QuantizeConfig(bits=4, quant_method="ganq", format="fake", act_sort="asc", l_damp_style="ganq", dead="mean",
                              damp_percent=0.01, desc_act=True, group_size=128, ganq_iterations=2)

    m_hip, m_cpu = copy.deepcopy(base), copy.deepcopy(base)
    p_hip = quantize_model(m_hip, calib, qcfg())
    p_cpu = quantize_model(m_cpu, calib, qcfg(), processor=OracleProcessor(qcfg()), share_group_hessian=False)
    assert sorted(p_hip.results()) == sorted(p_cpu.results()) and len(p_hip.results()) == 12
    # Per module: how many indices differ.  The first group (layer 0 q/k/v) sees identical inputs in both runs and must
    # agree index for index up to a stray near-tie.  From then on every module is calibrated on the outputs of the modules quantized before it,
    # so the two runs no longer see the same Hessians: codebooks that agree to 1e-4 upstream change activations by 1e-4,
    # which moves near-ties -- differences grow with depth (measured: 0.3 % .. 8 % of the indices in the later modules,
    # with equal losses).  That is a property of sequential calibration, not of either solver; what has to hold at
    # the model boundary is that both quantized models are equally good: losses, logits, perplexity.
    worst = 0.0
    for name in sorted(p_hip.results()):
        a, b = p_hip.results()[name], p_cpu.results()[name]
        frac = float((a["ganq_q"] != b["ganq_q"]).float().mean())
        tdiff = float((a["ganq_lut"] - b["ganq_lut"]).norm() / b["ganq_lut"].norm())
        print(f"[opt-125m architecture, HIP vs CPU oracle] {name}: differing indices {frac:.5f}, codebooks rel. Frobenius {tdiff:.3e}")
        worst = max(worst, frac)
        if "layers.0.self_attn" in name and not name.endswith("out_proj"):
            assert frac * a["ganq_q"].numel() <= 4, name  # measured: 0 or 1 of 589 824 (a near-tie of the S-solve)
    assert worst < 0.15
    la = [float(r["loss"]) for r in p_hip.log]
    lb = [float(r["loss"]) for r in p_cpu.log]
    assert np_close(la, lb, 2e-2)  # the logged loss (ganq.py:637-638) divides by diag(Hinv)^2: a logging quantity
    x = test_ids[:, :512].cuda()
    lo_hip, lo_cpu, lo_base = m_hip(x).logits.float(), m_cpu(x).logits.float(), base(x).logits.float()
    d_q = float((lo_hip - lo_cpu).norm() / lo_cpu.norm())
    d_b = float((lo_cpu - lo_base).norm() / lo_base.norm())
    print(f"[opt-125m architecture] logits: HIP vs oracle {d_q:.3e}; quantized vs fp16 {d_b:.3e}")
    # two equally good roundings of the same weights: closer to each other than either is to the fp16 model (measured 0.3x)
    assert d_q < 0.5 * d_b
    ppl_hip, ppl_cpu = gptq_style_ppl(m_hip, test_ids, seqlen=512), gptq_style_ppl(m_cpu, test_ids, seqlen=512)
    print(f"[opt-125m architecture] GPTQ-style PPL (random weights, random tokens): HIP {ppl_hip:.3f}  oracle {ppl_cpu:.3f}")
    assert abs(ppl_hip - ppl_cpu) / ppl_cpu < 0.05 / 28.45


def np_close(a, b, rtol):
    import numpy as np

    return bool(np.allclose(np.asarray(a), np.asarray(b), rtol=rtol))


def test_eval_ppl_tool_end_to_end_on_local_files(tmp_path):
    """tools/eval_ppl.py on a model directory and datasets that exist LOCALLY (a tiny random OPT with a word-level tokenizer,
    wikitext-style parquet splits, a c4-style json.gz shard): the tool that produces the PPL half of the metric the moment
    real opt-125m weights and wikitext-2 / c4 files are placed on the box"""
    import gzip
    import json
    import subprocess
    import sys

    import pyarrow as pa
    import pyarrow.parquet as pq
    import transformers
    from tokenizers import Tokenizer, models, pre_tokenizers

    words = ["w%d" % i for i in range(300)]
    vocab = {"<unk>": 0, "</s>": 1, **{w: i + 2 for i, w in enumerate(words)}}
    tk = Tokenizer(models.WordLevel(vocab, unk_token="<unk>"))
    tk.pre_tokenizer = pre_tokenizers.Whitespace()
    fast = transformers.PreTrainedTokenizerFast(tokenizer_object=tk, unk_token="<unk>", eos_token="</s>")
    mdir = tmp_path / "tiny-opt"
    fast.save_pretrained(str(mdir))
    torch.manual_seed(0)
    cfg = transformers.OPTConfig(vocab_size=320, hidden_size=64, ffn_dim=128, num_hidden_layers=2, num_attention_heads=4,
                                 max_position_embeddings=128, word_embed_proj_dim=64, eos_token_id=1, bos_token_id=1, pad_token_id=1)
    transformers.OPTForCausalLM(cfg).half().save_pretrained(str(mdir))
    g = torch.Generator().manual_seed(3)

    def doc(nwords):
        return " ".join(words[int(i)] for i in torch.randint(0, 300, (nwords,), generator=g))

    wdir = tmp_path / "wikitext" / "wikitext-2-raw-v1"
    wdir.mkdir(parents=True)
    pq.write_table(pa.table({"text": [doc(80) for _ in range(12)]}), str(wdir / "train-00000-of-00001.parquet"))
    pq.write_table(pa.table({"text": [doc(60) for _ in range(10)]}), str(wdir / "test-00000-of-00001.parquet"))
    cdir = tmp_path / "c4" / "en"
    cdir.mkdir(parents=True)
    with gzip.open(cdir / "c4-train.00000-of-01024.json.gz", "wt") as f:
        for _ in range(40):
            f.write(json.dumps({"text": doc(100)}) + "\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    proc = subprocess.run([sys.executable, os.path.join(root, "tools", "eval_ppl.py"), "--model-path", str(mdir), "--wikitext-path",
                           str(tmp_path / "wikitext"), "--c4-path", str(tmp_path / "c4"), "--nsamples", "6", "--seqlen", "64",
                           "--eval-seqlen", "64", "--iters", "2", "--save", str(tmp_path / "out")],
                          capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, proc.stderr[-2000:]
    out = json.loads(proc.stdout.strip().splitlines()[-1])
    assert out["calib"] == "c4" and len(out["modules"]) == 12
    assert 1.0 < out["ppl_fp16"] < 1e4 and 1.0 < out["ppl_ganq"] < 1e4
    assert abs(out["ppl_delta"]) < 0.2 * out["ppl_fp16"]
    assert os.path.exists(os.path.join(out["saved"], "model.safetensors"))

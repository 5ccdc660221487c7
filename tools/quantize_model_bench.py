#!/usr/bin/env python3
"""Full-model GANQ quantization timing on a random-initialised architecture (BASELINE.json configs[1]/[2] shape:
opt-125m / Llama-3.2-1B, 4-bit, 128 x 2048 synthetic calibration tokens).  No network: weights are random, so this
measures time and self-consistency, not perplexity against the published numbers.

    python tools/quantize_model_bench.py --arch opt-125m --nsamples 128 --seqlen 2048 --iters 10
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

ARCHS = {
    "opt-125m": ("opt", dict(vocab_size=50272, hidden_size=768, ffn_dim=3072, num_hidden_layers=12, num_attention_heads=12,
                             max_position_embeddings=2048, word_embed_proj_dim=768)),
    "opt-350m-like": ("opt", dict(vocab_size=50272, hidden_size=1024, ffn_dim=4096, num_hidden_layers=24,
                                  num_attention_heads=16, max_position_embeddings=2048, word_embed_proj_dim=1024)),
    "llama-3.2-1b": ("llama", dict(vocab_size=128256, hidden_size=2048, intermediate_size=8192, num_hidden_layers=16,
                                   num_attention_heads=32, num_key_value_heads=8, max_position_embeddings=131072,
                                   rope_theta=500000.0, tie_word_embeddings=True)),
}


def run(arch="opt-125m", nsamples=128, seqlen=2048, batch=8, bits=4, iters=10, layers=0, looper_options=None):
    import transformers

    from ganq_amd.models import gptq_style_ppl, quantize_model
    from ganq_amd.quantization import QuantizeConfig

    kind, kw = ARCHS[arch]
    if layers:
        kw = dict(kw, num_hidden_layers=layers)
    torch.manual_seed(0)
    cfg = (transformers.OPTConfig if kind == "opt" else transformers.LlamaConfig)(**kw)
    model = (transformers.OPTForCausalLM if kind == "opt" else transformers.LlamaForCausalLM)(cfg).half().cuda().eval()
    g = torch.Generator().manual_seed(1)
    calib = [torch.randint(0, cfg.vocab_size, (batch, seqlen), generator=g) for _ in range(nsamples // batch)]
    test_ids = torch.randint(0, cfg.vocab_size, (1, seqlen * 4), generator=g)
    ppl_fp = gptq_style_ppl(model, test_ids, seqlen)
    qcfg = QuantizeConfig(bits=bits, act_sort="asc", l_damp_style="ganq", dead="mean", ganq_iterations=iters,
                          damp_percent=0.01, desc_act=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    proc = quantize_model(model, calib, qcfg, **(looper_options or {}))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ppl_q = gptq_style_ppl(model, test_ids, seqlen)
    quant_s = sum(float(r["time"]) for r in proc.log)
    cols = sum(m.in_features for m in model.modules() if type(m).__name__ == "GanqHipQuantLinear")
    shapes = {}
    for m in model.modules():
        if type(m).__name__ == "GanqHipQuantLinear":
            shapes[(m.out_features, m.in_features)] = shapes.get((m.out_features, m.in_features), 0) + 1
    by_name = {}
    for r in proc.log:
        by_name[r["module"]] = by_name.get(r["module"], 0.0) + float(r["time"])
    return {"arch": arch, "layers": cfg.num_hidden_layers, "modules": len(proc.log), "bits": bits,
            "ganq_iterations": iters, "calibration": f"{nsamples}x{seqlen} synthetic tokens",
            "total_s": round(dt, 3), "sum_module_quantize_s": round(quant_s, 3),
            "quantize_s_by_module": {k: round(v, 3) for k, v in by_name.items()}, "weight_columns": cols,
            "columns_per_s_whole_run": round(cols / dt, 1),
            "module_shapes": {f"{m}x{n}": c for (m, n), c in sorted(shapes.items())},
            "ppl_random_init_fp16": round(ppl_fp, 2), "ppl_random_init_ganq": round(ppl_q, 2)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--arch", default="opt-125m", choices=sorted(ARCHS))
    ap.add_argument("--nsamples", type=int, default=128)
    ap.add_argument("--seqlen", type=int, default=2048)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--bits", type=int, default=4)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--layers", type=int, default=0, help="truncate the model to this many layers (0 = all)")
    a = ap.parse_args()
    print(json.dumps(run(a.arch, a.nsamples, a.seqlen, a.batch, a.bits, a.iters, a.layers)))


if __name__ == "__main__":
    main()

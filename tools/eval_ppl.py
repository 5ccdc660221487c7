#!/usr/bin/env python3
"""The second half of the headline metric: Wiki2 perplexity of a 4-bit GANQ model, GPTQ-style (README.md:21-27 of the
reference: opt-125m fp16 27.65 / GANQ 28.45; recipe examples/quantization/basic_usage_wikitext2.py:63-93,114-141).

Everything comes from LOCAL paths (the boxes have no network):

    python tools/eval_ppl.py --model-path /data/opt-125m --wikitext-path /data/wikitext --c4-path /data/c4 \
        [--calib c4|wikitext2] [--nsamples 32] [--seqlen 2048] [--bits 4] [--iters 10] [--format ganq_lut|fake] [--save DIR]

  --model-path      a Hugging Face model directory (config.json, tokenizer files, *.safetensors / *.bin)
  --wikitext-path   wikitext-2-raw-v1 with `train` and `test` splits (layouts: ganq_amd/models/calibration.py)
  --c4-path         c4 en/c4-train.00000-of-01024.json.gz, or a directory holding it (only for --calib c4, the example's choice)

Prints one JSON line: fp16 PPL, quantized PPL, their difference, quantization seconds, per-module log.  With the
reference's settings the GANQ figure to compare with is 28.45 (its CPU path); the acceptance window is +-0.05.
Exits with status 2 and a message when a path is missing -- it never downloads.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def need(path, what):
    if not path or not os.path.exists(path):
        print(f"eval_ppl: {what} not found at `{path}`.  This tool only reads local files (no network): place the "
              f"{what} there or pass another path.", file=sys.stderr)
        raise SystemExit(2)


def evaluate(model_path, wikitext_path, c4_path=None, calib="c4", nsamples=32, seqlen=2048, eval_seqlen=2048, bits=4, iters=10,
             fmt="ganq_lut", outlier_ratio=0.0, skip_fp16=False, save=None, looper_options=None):
    """fp16 PPL, quantization with the example's configuration, quantized PPL -> dict (see module docstring)"""
    import torch
    import transformers

    from ganq_amd.models import (as_batches, get_c4, get_wikitext2, gptq_style_ppl, quantize_model, save_quantized,
                                 wikitext2_test_ids)
    from ganq_amd.quantization import QuantizeConfig

    tok = transformers.AutoTokenizer.from_pretrained(model_path, use_fast=True, local_files_only=True)
    model = transformers.AutoModelForCausalLM.from_pretrained(model_path, torch_dtype=torch.float16,
                                                              local_files_only=True).cuda().eval()
    test_ids = wikitext2_test_ids(tok, wikitext_path)
    out = {"model": model_path, "bits": bits, "ganq_iterations": iters, "calib": calib,
           "nsamples": nsamples, "seqlen": seqlen, "eval_tokens": int(test_ids.numel())}
    if not skip_fp16:
        out["ppl_fp16"] = gptq_style_ppl(model, test_ids, seqlen=eval_seqlen)
    if calib == "c4":
        samples = get_c4(tok, nsamples, seqlen, c4_path)
    else:
        samples = get_wikitext2(tok, nsamples, seqlen, wikitext_path)
    qcfg = QuantizeConfig(bits=bits, quant_method="ganq", format=fmt, ganq_iterations=iters, act_sort="asc",
                          l_damp_style="ganq", dead="mean", ganq_outlier_ratio=outlier_ratio)
    t0 = time.time()
    proc = quantize_model(model, as_batches(samples), qcfg, **(looper_options or {}))
    torch.cuda.synchronize()
    out["quantize_s"] = round(time.time() - t0, 2)
    out["ppl_ganq"] = gptq_style_ppl(model, test_ids, seqlen=eval_seqlen)
    if "ppl_fp16" in out:
        out["ppl_delta"] = out["ppl_ganq"] - out["ppl_fp16"]
    out["reference_readme"] = {"opt-125m fp16": 27.65, "opt-125m GANQ 4-bit (CPU path)": 28.45, "window": 0.05}
    out["modules"] = proc.log
    if save:
        save_quantized(model, save, qcfg)
        out["saved"] = save
    return out


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--model-path", required=True)
    ap.add_argument("--wikitext-path", required=True)
    ap.add_argument("--c4-path", default=None)
    ap.add_argument("--calib", choices=["c4", "wikitext2"], default="c4")
    ap.add_argument("--nsamples", type=int, default=32)
    ap.add_argument("--seqlen", type=int, default=2048)
    ap.add_argument("--eval-seqlen", type=int, default=2048)
    ap.add_argument("--bits", type=int, default=4)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--format", choices=["ganq_lut", "fake"], default="ganq_lut")
    ap.add_argument("--outlier-ratio", type=float, default=0.0)
    ap.add_argument("--skip-fp16", action="store_true")
    ap.add_argument("--save", default=None)
    args = ap.parse_args()

    need(args.model_path, "model directory")
    need(args.wikitext_path, "wikitext-2-raw-v1 data")
    if args.calib == "c4":
        need(args.c4_path, "c4 shard (en/c4-train.00000-of-01024.json.gz)")

    import torch

    if not torch.cuda.is_available():
        print("eval_ppl: needs the MI355X (the HIP path has no CPU fallback)", file=sys.stderr)
        raise SystemExit(2)
    out = evaluate(args.model_path, args.wikitext_path, args.c4_path, args.calib, args.nsamples, args.seqlen, args.eval_seqlen,
                   args.bits, args.iters, args.format, args.outlier_ratio, args.skip_fp16, args.save)
    print(json.dumps(out))


if __name__ == "__main__":
    main()

"""A TRAINED tiny causal LM and what the reference's own GANQ makes of it -- the stand-in for the Wiki2-PPL half of the metric
while no opt-125m checkpoint or dataset is reachable (no network).  Build container only:

    python tests/golden/make_golden_tiny_lm.py train        # trains tests/golden/tiny_lm/model.safetensors (CPU, ~15 min)
    python tests/golden/make_golden_tiny_lm.py reference    # quantizes it with the reference's GANQ, writes fixture.npz

* corpus: text that ships with the interpreter -- the top-level modules of the standard library (source with docstrings) and
  the language reference in `pydoc_data.topics`, 5.2 MB -- as byte-level tokens (vocabulary 256); the first 85 % train /
  calibrate, the next 5 % pick the checkpoint, 128 windows of the last 10 % are the evaluation text.
* model: the OPT architecture (BASELINE.json's opt-125m family; `definitions/opt.py:34-41` layer map), 4 decoder layers,
  hidden 256, ffn 1024, 4 heads, 512 positions: 3.4 M parameters, trained with AdamW from `torch.manual_seed(0)`
  (dropout 0.1, best validation checkpoint of 2400 steps); the
  weights are rounded to fp16 and stored as fp16 (6.7 MB) -- what both sides quantize is exactly representable in fp16.
* reference run: the reference's `GANQ(GPTQ)` object (ganq.py:397-646, gptq.py:42-393; loaded by ref_loader.py, kmeans1d
  replaced by the oracle's exact k-means: T0 is parity-unpinned) sits in the quantizer slot of this
  repository's looper -- the layer / group order and the re-forward through the quantized layer are those of
  module_looper.py:205-417 -- on the CPU in fp32, with the recipe of examples/quantization/basic_usage_wikitext2.py:120-134:
  4-bit, K = 10, act_sort="asc", l_damp_style="ganq", dead="mean", desc_act=True, damp 0.01.
* recorded (fixture.npz): calibration / evaluation token ids, per module the reference's best codebook T [m,16] and its
  indices (4-bit packed, original column order) with the sha256 of the returned weight, avg_loss per module, the GPTQ-style
  PPL (basic_usage_wikitext2.py:63-93) of the fp model and of the reference-quantized model -- and of SEVEN MORE reference
  runs with 1e-6 relative noise on the calibration activations: GANQ's error-feedback solve is a chaotic map of its inputs
  (noise of the size of one platform's rounding re-decides a third of the indices downstream), so on a 3.4 M-parameter model
  the reference's own PPL moves by more than the +-0.05 of the metric; the spread is what a faithful implementation is held to.
`tests/test_tiny_lm.py` (CPU: the oracle in the quantizer slot; -m gpu: the HIP path) quantizes the same model on the same batches with the HIP path and compares.
"""
import hashlib
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

OUT = os.path.join(HERE, "tiny_lm")
SEQ = 512
N_CALIB = 64
K = 10
BITS = 4
TRAIN_STEPS = 2400


def corpus() -> np.ndarray:
    """text that ships with the interpreter: the top-level modules of the standard library (source code with its docstrings,
    4.7 MB) and the language reference of `pydoc_data.topics` (0.46 MB), files in the order of the sha1 of their names"""
    import glob
    import sysconfig

    std = sysconfig.get_paths()["stdlib"]
    files = sorted(glob.glob(os.path.join(std, "*.py"))) + [os.path.join(std, "pydoc_data", "topics.py")]
    files.sort(key=lambda f: hashlib.sha1(os.path.basename(f).encode()).hexdigest())
    data = b"\n\n".join(open(f, "rb").read() for f in files)
    return np.frombuffer(data, dtype=np.uint8).copy()


def model_config(dropout=0.0):
    from transformers import OPTConfig

    return OPTConfig(vocab_size=256, hidden_size=256, ffn_dim=1024, num_hidden_layers=4, num_attention_heads=4,
                     max_position_embeddings=SEQ, word_embed_proj_dim=256, dropout=dropout, attention_dropout=0.0,
                     pad_token_id=1, bos_token_id=2, eos_token_id=2)


def load_model(dtype=torch.float32):
    """the committed tiny model (also used by the GPU test and bench.py: no reference needed)"""
    from safetensors.torch import load_file
    from transformers import OPTConfig, OPTForCausalLM

    with open(os.path.join(OUT, "config.json")) as f:
        cfg = OPTConfig(**json.load(f))
    model = OPTForCausalLM(cfg)
    state = load_file(os.path.join(OUT, "model.safetensors"))
    missing, unexpected = model.load_state_dict({k: v.float() for k, v in state.items()}, strict=False)
    assert not unexpected and all("lm_head" in k for k in missing), (missing, unexpected)
    model.tie_weights()
    return model.to(dtype).eval()


def train():
    from safetensors.torch import save_file
    from transformers import OPTForCausalLM

    os.makedirs(OUT, exist_ok=True)
    data = corpus()
    split, vsplit = int(len(data) * 0.9), int(len(data) * 0.85)   # train | validation (checkpoint selection) | held-out evaluation
    ids = torch.from_numpy(data[:vsplit].astype(np.int64))
    val = torch.from_numpy(data[vsplit:split].astype(np.int64))
    val = val[: 64 * SEQ].reshape(-1, SEQ)
    torch.manual_seed(0)
    torch.set_num_threads(os.cpu_count() or 1)
    cfg = model_config(dropout=0.1)  # dropout while training; the best validation checkpoint is kept
    model = OPTForCausalLM(cfg)
    opt = torch.optim.AdamW(model.parameters(), lr=2e-3, weight_decay=0.1, betas=(0.9, 0.95))
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=2e-3, total_steps=TRAIN_STEPS, pct_start=0.05)
    g = torch.Generator().manual_seed(1)
    t0 = time.time()
    best_val, best_state = float("inf"), None
    for step in range(TRAIN_STEPS):
        model.train()
        off = torch.randint(0, len(ids) - 257, (32,), generator=g)
        x = torch.stack([ids[o:o + 256] for o in off.tolist()])
        loss = model(x, labels=x).loss
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        opt.step()
        sched.step()
        opt.zero_grad(set_to_none=True)
        if step % 100 == 99 or step == TRAIN_STEPS - 1:
            model.eval()
            with torch.no_grad():
                vl = float(model(val, labels=val).loss)
            if vl < best_val:
                best_val, best_state = vl, {k: v.detach().clone() for k, v in model.state_dict().items()}
            print(f"step {step}: loss {float(loss.detach()):.4f}  validation {vl:.4f} (best {best_val:.4f})  ({time.time() - t0:.0f} s)", flush=True)
    model.load_state_dict(best_state)
    cfg = model_config()
    state = {k: v.detach().half().contiguous() for k, v in model.state_dict().items() if k != "lm_head.weight"}
    save_file(state, os.path.join(OUT, "model.safetensors"), metadata={"format": "pt"})
    keep = ("vocab_size", "hidden_size", "ffn_dim", "num_hidden_layers", "num_attention_heads", "max_position_embeddings",
            "word_embed_proj_dim", "dropout", "attention_dropout", "pad_token_id", "bos_token_id", "eos_token_id",
            "do_layer_norm_before", "activation_function")
    with open(os.path.join(OUT, "config.json"), "w") as f:
        json.dump({k: getattr(cfg, k) for k in keep}, f, indent=1)
    print(f"saved {os.path.getsize(os.path.join(OUT, 'model.safetensors')) / 1e6:.2f} MB")


def token_sets():
    """(calibration batches [N_CALIB, SEQ] int64 from the training part, evaluation ids [1, total] from the held-out part)"""
    data = corpus()
    split, vsplit = int(len(data) * 0.9), int(len(data) * 0.85)
    rng = np.random.default_rng(7)
    offs = rng.choice(vsplit - SEQ, size=N_CALIB, replace=False)
    calib = np.stack([data[o:o + SEQ] for o in offs]).astype(np.int64)
    return calib, data[split:split + 128 * SEQ].astype(np.int64)[None, :]  # 128 evaluation windows


def pack4(Q: np.ndarray) -> np.ndarray:
    return (Q[:, 0::2] | (Q[:, 1::2] << 4)).astype(np.uint8)


def reference():
    import ref_loader
    from oracle import c_oracle

    from ganq_amd.looper.gptq_processor import GPTQProcessor
    from ganq_amd.models.quantize import gptq_style_ppl, quantize_model
    from ganq_amd.quantization.config import QuantizeConfig as OurConfig

    def kmeans_cluster(values, k, weights=None):  # stands in for kmeans1d.cluster (ganq.py:29): exact weighted 1-D k-means
        w = np.ascontiguousarray(np.asarray(values, dtype=np.float32).reshape(1, -1))
        cw = None if weights is None else np.asarray(weights, dtype=np.float64)
        return None, c_oracle.kmeans_init(w, cw, k)[0].astype(np.float64).tolist()

    ganq_mod, _gptq_mod, cfg_mod, RefNamedModule = ref_loader.load_reference(kmeans_cluster)
    torch.set_num_threads(os.cpu_count() or 1)
    calib, eval_ids = token_sets()
    ppl_fp = gptq_style_ppl(load_model(torch.float32), torch.from_numpy(eval_ids), seqlen=SEQ)
    print(f"fp32 PPL {ppl_fp:.4f}", flush=True)

    N_RUNS = 8   # run 0: the reference as it is (per-module records); runs 1..7: the same with 1e-6 relative noise on the calibration
                 # activations -- how far the reference's OWN result moves under noise of the size of a platform's rounding
    state = {"rec": {}, "noise": 0}

    class RefTask:
        """the reference's GANQ object behind the attribute names this repository's looper / processor use"""

        def __init__(self, module, qcfg):
            rq = cfg_mod.QuantizeConfig(bits=qcfg.bits, quant_method="ganq", format="fake", act_sort=qcfg.act_sort,
                                        l_damp_style=qcfg.l_damp_style, dead=qcfg.dead, desc_act=qcfg.desc_act,
                                        ganq_iterations=qcfg.ganq_iterations, group_size=qcfg.group_size,
                                        damp_percent=qcfg.damp_percent)
            self.full_name = module.full_name
            self.g = ganq_mod.GANQ(RefNamedModule(module.module, module.name, module.full_name, module.layer_index), rq)
            self.qcfg = qcfg
            self.ganq_indices = self.ganq_codebook = self.ganq_outliers = None
            self.ganq_stats = {}
            seed = int(hashlib.sha1(f"{state['noise']}:{module.full_name}".encode()).hexdigest()[:8], 16)
            self.gen = torch.Generator().manual_seed(seed)

        quantizer = property(lambda self: self.g.quantizer)
        fwd_counter = property(lambda self: self.g.fwd_counter)
        nsamples = property(lambda self: self.g.nsamples)
        columns = property(lambda self: self.g.columns)

        def add_batch(self, inp, out):
            if state["noise"]:
                inp = inp * (1.0 + 1e-6 * torch.randn(inp.shape, generator=self.gen, dtype=inp.dtype))
            self.g.add_batch(inp, out)

        def quantize(self):
            sols, losses = [], []
            real_lstsq, real_loss = torch.linalg.lstsq, ganq_mod.quad_loss_2

            def lstsq_wrap(A, B, *a, **k):
                out = real_lstsq(A, B, *a, **k)
                sols.append(out.solution.clone())
                return out

            def loss_wrap(Wm, Wq, G):
                out = real_loss(Wm, Wq, G)
                losses.append(float(out))
                return out

            torch.linalg.lstsq, ganq_mod.quad_loss_2 = lstsq_wrap, loss_wrap
            try:
                out = self.g.quantize()
            finally:
                torch.linalg.lstsq, ganq_mod.quad_loss_2 = real_lstsq, real_loss
            wq = out[0]
            best = int(np.argmin(np.array(losses)))  # ganq.py:625: strict <, first minimum
            T = sols[best].mT.squeeze(-2).float().numpy()
            Wq = wq.float().numpy()
            # indices in the returned (original) column order: the position of each weight in its row's codebook
            Q = np.argmin(np.abs(Wq[:, :, None] - T[:, None, :]), axis=2).astype(np.uint8)
            assert np.array_equal(np.take_along_axis(T, Q.astype(np.int64), axis=1), Wq), self.full_name
            self.ganq_indices = torch.from_numpy(Q)
            self.ganq_codebook = torch.from_numpy(T)
            if not state["noise"]:
                state["rec"][self.full_name] = dict(T=T, Q=pack4(Q), sha=hashlib.sha256(Wq.tobytes()).hexdigest(),
                                                    avg_loss=float(out[5]), dists=np.array(losses), best_k=best)
                print(f"  {self.full_name}: avg_loss {out[5]:.6g} best_k {best}", flush=True)
            return out

        def free(self):
            self.g.free()

    class RefProcessor(GPTQProcessor):
        def preprocess(self, module, buffered_fwd=False):
            tmp = RefTask(module, self.qcfg)
            tmp.quantizer.configure(perchannel=True)
            self.tasks[module.name] = tmp

    qcfg = OurConfig(bits=BITS, quant_method="ganq", format="fake", act_sort="asc", l_damp_style="ganq", dead="mean",
                     desc_act=True, ganq_iterations=K, group_size=128, damp_percent=0.01)
    batches = [torch.from_numpy(calib[i:i + 1]) for i in range(N_CALIB)]
    ppl_runs = []
    for run in range(N_RUNS):
        state["noise"] = run
        t0 = time.time()
        model = load_model(torch.float32)
        quantize_model(model, batches, qcfg, processor=RefProcessor(qcfg), share_group_hessian=False, concurrent_group=False,
                       dist_mode="none")
        ppl_runs.append(gptq_style_ppl(model, torch.from_numpy(eval_ids), seqlen=SEQ))
        print(f"reference GANQ run {run} ({'as is' if run == 0 else '1e-6 input noise'}): PPL {ppl_runs[-1]:.4f} "
              f"(fp {ppl_fp:.4f})  {time.time() - t0:.0f} s", flush=True)
    rec = state["rec"]
    ppl_ref = ppl_runs[0]
    print(f"reference-GANQ PPL {ppl_ref:.4f}; over {N_RUNS} runs: mean {np.mean(ppl_runs):.4f} std {np.std(ppl_runs, ddof=1):.4f} "
          f"min {min(ppl_runs):.4f} max {max(ppl_runs):.4f}", flush=True)
    names = sorted(rec)
    out = dict(calib=calib.astype(np.uint8), eval_ids=eval_ids.astype(np.uint8), names=np.array(names), ppl_fp=np.float64(ppl_fp),
               ppl_ref=np.float64(ppl_ref), ppl_ref_runs=np.array(ppl_runs, dtype=np.float64), bits=BITS, K=K, seq=SEQ,
               sha_model=hashlib.sha256(open(os.path.join(OUT, "model.safetensors"), "rb").read()).hexdigest())
    for i, n in enumerate(names):
        r = rec[n]
        out[f"T_{i}"], out[f"Q_{i}"], out[f"sha_{i}"] = r["T"], r["Q"], r["sha"]
        out[f"avg_loss_{i}"], out[f"dists_{i}"], out[f"best_k_{i}"] = np.float64(r["avg_loss"]), r["dists"], r["best_k"]
    path = os.path.join(OUT, "fixture.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {os.path.getsize(path) / 1e6:.2f} MB")


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "reference"
    {"train": train, "reference": reference}[what]()

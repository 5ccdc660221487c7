"""Large golden cases: the reference's OWN loop (`GANQ._perform_quantization_loop`, ganq.py:456-646) at n = 2048 and on a
4096 x 4096 layer.  Build container only (`/root/reference` must exist):

    python tests/golden/make_golden_large.py [case ...]

The inputs of the loop (W, Xxt_damped, L, diag(Hinv), T0) come from tests/golden/exact_inputs.py -- every box rebuilds them
bit for bit from the seed, so the fixture stores only their sha256 (an n x n fp32 L cannot be committed at these sizes) and
what the reference produced:

  Q_k      indices of every iteration (captured at torch.argmin, as make_golden.py does): stored as first iteration + XOR
           deltas; the 4096 x 4096 case is hash-only -- sha256 of each Q_k plus one 64-bit digest per row
  T_k      codebooks T_0 .. T_K (captured at torch.linalg.lstsq)
  dists    quad_loss_2 of every iteration
  Wq/Losses  sha256 / sums of the loop's outputs
The reference object is constructed exactly as in make_golden.py; `self.L` / `self.Xxt_damped` (what gptq.py:288-300 leaves
behind) are set from the generated inputs and the override point is called directly, so the prologue's LAPACK calls --
which no other box could reproduce bit for bit -- stay out of the picture.  Stages pinned: a4/a5 S-solve, a6 T-update,
a7 loss / best-of-K / outputs.
"""
import hashlib
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import exact_inputs  # noqa: E402
import ref_loader  # noqa: E402

OUT = os.path.join(HERE, "large")

CASES = [
    dict(name="l128x2048_b4_k3", m=128, n=2048, bits=4, K=3, seed=201, tokens=4096, hash_only=False),
    dict(name="l128x2048_b3_k3", m=128, n=2048, bits=3, K=3, seed=201, tokens=4096, hash_only=False),
    dict(name="l128x2048_b4_k10", m=128, n=2048, bits=4, K=10, seed=202, tokens=4096, hash_only=False),
    dict(name="l256x512_b2_k3", m=256, n=512, bits=2, K=3, seed=204, tokens=2048, hash_only=False),
    dict(name="h4096x4096_b4_k2", m=4096, n=4096, bits=4, K=2, seed=203, tokens=8192, hash_only=True),
    # the module shapes of opt-125m (BASELINE.json configs[0] / [1]: fc1, fc2, attention projections), all K = 10 iterations
    dict(name="h3072x768_b4_k10", m=3072, n=768, bits=4, K=10, seed=206, tokens=2048, hash_only=True),
    dict(name="h768x3072_b4_k10", m=768, n=3072, bits=4, K=10, seed=207, tokens=4096, hash_only=True),
    dict(name="h768x768_b4_k10", m=768, n=768, bits=4, K=10, seed=208, tokens=2048, hash_only=True),
    # round 4: the long chains of the Llama shapes (down_proj: n = 8192 in Llama-3.2-1B, n = 14336 in Llama-3-8B) on a few rows --
    # rows are independent in every stage (ganq.py:525-634), so 32 / 64 rows cost the reference seconds -- and a whole layer
    # at V = 8 (BASELINE.json configs[4]); the residual gemv of ganq.py:564-565 sums up to 14335 products here
    dict(name="l64x8192_b4_k3", m=64, n=8192, bits=4, K=3, seed=211, tokens=8192, hash_only=False),
    dict(name="l32x14336_b4_k2", m=32, n=14336, bits=4, K=2, seed=212, tokens=16384, hash_only=False),
    dict(name="l32x14336_b3_k2", m=32, n=14336, bits=3, K=2, seed=212, tokens=16384, hash_only=False),
    dict(name="h1024x4096_b3_k2", m=1024, n=4096, bits=3, K=2, seed=213, tokens=8192, hash_only=True),
    # torch.argmin's NaN semantics in the S-solve (ganq.py:547): two codebook entries are NaN; only the S-solve of the one
    # iteration is captured (what lstsq / the loss make of NaN inputs is not part of the contract: gptq.py:328-330 raises)
    dict(name="nan48x256_b4_k1", m=48, n=256, bits=4, K=1, seed=205, tokens=1024, hash_only=False, nan_entries=[(3, 6), (10, 0), (10, 9)]),
]


def run_case(c, ganq_mod, cfg_mod, NamedModule):
    m, n, K, V = c["m"], c["n"], c["K"], 2 ** c["bits"]
    t0 = time.time()
    inp = exact_inputs.make(m, n, c["bits"], c["seed"], c["tokens"])
    print(f"{c['name']}: inputs in {time.time() - t0:.1f} s", flush=True)
    lin = torch.nn.Linear(n, m, bias=False).half()
    qcfg = cfg_mod.QuantizeConfig(bits=c["bits"], quant_method="ganq", format="fake", act_sort="asc", l_damp_style="ganq",
                                  dead="mean", desc_act=True, ganq_iterations=K, group_size=128, damp_percent=0.01)
    g = ganq_mod.GANQ(NamedModule(lin, "fc1", "model.layers.0.fc1", 0), qcfg)
    g.quantizer.configure(perchannel=True)
    g.L = torch.from_numpy(inp["L"])
    g.Xxt_damped = torch.from_numpy(inp["H"])
    Hinv = torch.diag(torch.from_numpy(inp["hinv_diag"]))
    for (r, e) in c.get("nan_entries", []):
        inp["T0"][r, e] = np.nan
    T0 = torch.from_numpy(inp["T0"])

    Qs = np.zeros((K, m, n), dtype=np.uint8)
    rec = dict(step=0, lstsq=[], loss=[])
    real_argmin, real_lstsq, real_loss = torch.argmin, torch.linalg.lstsq, ganq_mod.quad_loss_2
    real_init = ganq_mod.GANQ._initialize_codebook_kmeans

    def argmin_wrap(*a, **k):
        out = real_argmin(*a, **k)
        s = rec["step"]
        Qs[s // n, :, n - 1 - s % n] = out.numpy().astype(np.uint8)
        rec["step"] = s + 1
        return out

    def lstsq_wrap(A, B, *a, **k):
        out = real_lstsq(A, B, *a, **k)
        rec["lstsq"].append(out.solution.clone())
        return out

    def loss_wrap(Wm, Wq, G):
        out = real_loss(Wm, Wq, G)
        rec["loss"].append(float(out))
        print(f"  iteration {len(rec['loss'])}: dist {float(out):.8g}  ({time.time() - t0:.0f} s)", flush=True)
        return out

    ganq_mod.GANQ._initialize_codebook_kmeans = lambda self, *a, **k: T0.clone()
    torch.argmin, torch.linalg.lstsq, ganq_mod.quad_loss_2 = argmin_wrap, lstsq_wrap, loss_wrap
    loop_error = None
    try:
        Wq, Losses, _, _ = g._perform_quantization_loop(torch.from_numpy(inp["W"]).clone(), Hinv, 128)
    except Exception as e:  # NaN case: LAPACK may refuse the NaN system after the S-solve was captured
        if not c.get("nan_entries"):
            raise
        loop_error = e
    finally:
        ganq_mod.GANQ._initialize_codebook_kmeans = real_init
        torch.argmin, torch.linalg.lstsq, ganq_mod.quad_loss_2 = real_argmin, real_lstsq, real_loss
    if c.get("nan_entries"):
        assert rec["step"] >= n, loop_error
        out = dict(m=m, n=n, bits=c["bits"], K=K, seed=c["seed"], tokens=c["tokens"], hash_only=False, nan_case=True,
                   T0=inp["T0"], nan_entries=np.array(c["nan_entries"]), Q_first=Qs[0], sha_Q=np.array([exact_inputs.sha(Qs[0])]))
        out.update({"sha_" + k: v for k, v in exact_inputs.hashes(inp).items()})
        return out
    assert rec["step"] == K * n and len(rec["lstsq"]) == K and len(rec["loss"]) == K
    Ts = np.stack([inp["T0"]] + [s.mT.squeeze(-2).numpy() for s in rec["lstsq"]]).astype(np.float32)
    dists = np.array(rec["loss"], dtype=np.float64)
    best_k = int(np.argmin(dists))
    # the loop's output is T_best.gather(Q_last) (the aliasing quirk, ganq.py:487,550,625-626)
    assert np.array_equal(np.take_along_axis(Ts[best_k + 1], Qs[K - 1].astype(np.int64), axis=1), Wq.numpy())
    out = dict(m=m, n=n, bits=c["bits"], K=K, seed=c["seed"], tokens=c["tokens"], hash_only=c["hash_only"],
               T=Ts, dists=dists, T0=inp["T0"],
               sha_Wq=exact_inputs.sha(Wq.numpy()), sha_Losses=exact_inputs.sha(Losses.numpy()),
               losses_sum=np.float64(Losses.double().sum().item()),
               sha_Q=np.array([exact_inputs.sha(Qs[k]) for k in range(K)]))
    out.update({"sha_" + k: v for k, v in exact_inputs.hashes(inp).items()})
    if c["hash_only"]:
        out["Q_row_digest"] = np.stack([exact_inputs.row_digest(Qs[k]) for k in range(K)])
    else:
        out.update(exact_inputs.pack_q_trace(Qs))
    return out


def main():
    ganq_mod, _gptq_mod, cfg_mod, NamedModule = ref_loader.load_reference(None)
    torch.set_num_threads(os.cpu_count() or 1)
    os.makedirs(OUT, exist_ok=True)
    want = set(sys.argv[1:])
    for c in CASES:
        if want and c["name"] not in want:
            continue
        out = run_case(c, ganq_mod, cfg_mod, NamedModule)
        path = os.path.join(OUT, c["name"] + ".npz")
        np.savez_compressed(path, **out)
        print(f"{c['name']}: dists={out.get('dists')} -> {os.path.getsize(path) / 1e6:.2f} MB", flush=True)


if __name__ == "__main__":
    main()

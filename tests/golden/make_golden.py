"""Generate the committed golden vectors by running the upstream reference itself.

Run in the build container only (``/root/reference`` must exist):

    python tests/golden/make_golden.py

For each case the reference's own ``GANQ(GPTQ)`` object (gptqmodel/quantization/ganq.py +
gptq.py, loaded by ref_loader.py) is fed seeded inputs on CPU and instrumented from the
outside (wrappers around ``torch.argmin``, ``torch.linalg.lstsq`` and ``quad_loss_2``) so that
every stage of ``_perform_quantization_loop`` is captured without touching reference code:

  G1 prologue   (W, X batches, config)            -> perm, W_perm, L, Xxt_damped, diag(Hinv)
  G2 S-solve    (W_perm, L, T_k)                  -> Q_k            for every iteration k
  G3 T-update   (Q_k, Xxt_damped, W_perm)         -> A_k, B_k, T_{k+1}
  G4 loss       (W_perm, Xxt_damped, T_{k+1}, Q_k)-> dist_k
  G5 loop out   best-of-K                         -> Wq_loop, Losses
  G6 quantize() 7-tuple                           -> Wq (module dtype, un-permuted), scale, zero,
                                                     g_idx, avg_loss, damp_percent
  G7 forward    (x, Wq.half(), bias)              -> FakeQuantLinear.forward == F.linear

The initial codebook T0 is produced by the oracle's k-means (the reference's kmeans1d
dependency is not installed) and stored as an input.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import ref_loader  # noqa: E402
from oracle import c_oracle  # noqa: E402

CASES = [
    # name, m, n, bits, K, nbatch, bsz, seq, act_sort, l_damp_style, dead, desc_act, corr, dead_cols, bias
    dict(name="t24x48_b2", m=24, n=48, bits=2, K=3, nb=3, bsz=2, seq=40, act_sort="asc", l_damp="ganq", dead="mean",
         desc_act=True, corr=0.0, dead_cols=0, seed=11),
    dict(name="a32x64_b4", m=32, n=64, bits=4, K=3, nb=4, bsz=2, seq=64, act_sort="asc", l_damp="ganq", dead="mean",
         desc_act=True, corr=0.0, dead_cols=2, seed=12),
    dict(name="b32x64_b3", m=32, n=64, bits=3, K=3, nb=4, bsz=2, seq=64, act_sort="desc", l_damp="gptq",
         dead="zero", desc_act=True, corr=0.3, dead_cols=1, seed=13),
    dict(name="c64x256_b4", m=64, n=256, bits=4, K=3, nb=4, bsz=4, seq=96, act_sort="asc", l_damp="ganq",
         dead="mean", desc_act=True, corr=0.5, dead_cols=0, seed=14),
    dict(name="d48x384_b4", m=48, n=384, bits=4, K=2, nb=3, bsz=2, seq=128, act_sort="none", l_damp="ganq",
         dead="mean", desc_act=False, corr=0.0, dead_cols=0, seed=15),
    dict(name="e16x512_b3", m=16, n=512, bits=3, K=1, nb=2, bsz=2, seq=160, act_sort="asc", l_damp="ganq",
         dead="mean", desc_act=True, corr=0.1, dead_cols=0, seed=16),
]


def make_inputs(c):
    g = torch.Generator().manual_seed(c["seed"])
    m, n = c["m"], c["n"]
    W = (0.02 * torch.randn(m, n, generator=g)).half()
    bias = (0.01 * torch.randn(m, generator=g)).half()
    scale = 0.1 + torch.rand(n, generator=g)
    mix = torch.randn(n, n, generator=g) / (n ** 0.5)
    xs = []
    for _ in range(c["nb"]):
        z = torch.randn(c["bsz"], c["seq"], n, generator=g)
        x = ((1.0 - c["corr"]) * z + c["corr"] * (z @ mix)) * scale
        if c["dead_cols"]:
            x[..., : c["dead_cols"]] = 0.0
        xs.append(x.half())
    return W, bias, xs


def run_case(c, ganq_mod, gptq_mod, cfg_mod, NamedModule):
    W, bias, xs = make_inputs(c)
    m, n, V = c["m"], c["n"], 2 ** c["bits"]
    lin = torch.nn.Linear(n, m, bias=True).half()
    with torch.no_grad():
        lin.weight.copy_(W)
        lin.bias.copy_(bias)
    qcfg = cfg_mod.QuantizeConfig(bits=c["bits"], quant_method="ganq", format="fake", act_sort=c["act_sort"],
                                  l_damp_style=c["l_damp"], dead=c["dead"], desc_act=c["desc_act"],
                                  ganq_iterations=c["K"], group_size=128, damp_percent=0.01)
    g = ganq_mod.GANQ(NamedModule(lin, "fc1", "model.layers.0.fc1", 0), qcfg)
    g.quantizer.configure(perchannel=True)
    for x in xs:
        g.add_batch(x, None)
    H_raw = g.H.clone()
    nsamples = g.nsamples

    rec = dict(argmin=[], lstsq=[], loss=[], T0=None, W_perm=None, Hinv_diag=None)
    real_argmin, real_lstsq, real_loss = torch.argmin, torch.linalg.lstsq, ganq_mod.quad_loss_2
    real_init = ganq_mod.GANQ._initialize_codebook_kmeans

    def init_wrap(self, Wp, Hinv, num_bits, device):
        rec["W_perm"] = Wp.clone()
        rec["Hinv_diag"] = torch.diagonal(Hinv).clone()
        T0 = real_init(self, Wp, Hinv, num_bits, device)
        rec["T0"] = T0.clone()
        return T0

    def argmin_wrap(*a, **k):
        out = real_argmin(*a, **k)
        rec["argmin"].append(out.clone())
        return out

    def lstsq_wrap(A, B, *a, **k):
        out = real_lstsq(A, B, *a, **k)
        rec["lstsq"].append((A.clone(), B.clone(), out.solution.clone()))
        return out

    def loss_wrap(Wm, Wq, G):
        out = real_loss(Wm, Wq, G)
        rec["loss"].append((Wq.clone(), float(out)))
        return out

    ganq_mod.GANQ._initialize_codebook_kmeans = init_wrap
    torch.argmin = argmin_wrap
    torch.linalg.lstsq = lstsq_wrap
    ganq_mod.quad_loss_2 = loss_wrap
    loop_out = {}
    real_loop = ganq_mod.GANQ._perform_quantization_loop

    def loop_wrap(self, Wp, Hinv, blocksize, perm=None, invperm=None):
        loop_out["perm"] = None if perm is None else perm.clone()
        out = real_loop(self, Wp, Hinv, blocksize, perm, invperm)
        loop_out["Wq_loop"], loop_out["Losses"] = out[0].clone(), out[1].clone()
        loop_out["L"] = self.L.clone()
        loop_out["Xxt_damped"] = self.Xxt_damped.clone()
        return out

    ganq_mod.GANQ._perform_quantization_loop = loop_wrap
    try:
        wq, scale, zero, g_idx, duration, avg_loss, damp_percent = g.quantize()
    finally:
        ganq_mod.GANQ._initialize_codebook_kmeans = real_init
        ganq_mod.GANQ._perform_quantization_loop = real_loop
        torch.argmin = real_argmin
        torch.linalg.lstsq = real_lstsq
        ganq_mod.quad_loss_2 = real_loss

    K = c["K"]
    assert len(rec["argmin"]) == K * n and len(rec["lstsq"]) == K and len(rec["loss"]) == K
    Qs = np.zeros((K, m, n), dtype=np.uint8)
    for k in range(K):
        for step in range(n):
            j = n - 1 - step
            Qs[k, :, j] = rec["argmin"][k * n + step].numpy().astype(np.uint8)
    Ts = np.stack([rec["T0"].numpy()] + [s[2].mT.squeeze(-2).numpy() for s in rec["lstsq"]]).astype(np.float32)
    As = np.stack([s[0].numpy() for s in rec["lstsq"]]).astype(np.float32)
    Bs = np.stack([s[1].squeeze(-1).numpy() for s in rec["lstsq"]]).astype(np.float32)
    dists = np.array([s[1] for s in rec["loss"]], dtype=np.float64)
    # consistency: Wq recorded by the loss wrapper is T_{k+1}.gather(Q_k)
    for k in range(K):
        wq_k = np.take_along_axis(Ts[k + 1], Qs[k].astype(np.int64), axis=1)
        assert np.array_equal(wq_k, rec["loss"][k][0].numpy()), "captured Q/T inconsistent"

    # G7 forward oracle == FakeQuantLinear.forward (fake.py:88-89).  CPU fp16 F.linear.
    gx = torch.Generator().manual_seed(c["seed"] + 1000)
    x_fwd = torch.randn(5, n, generator=gx).half()
    y_fwd = torch.nn.functional.linear(x_fwd, wq, lin.bias.data)

    out = dict(
        m=m, n=n, bits=c["bits"], K=K, act_sort=c["act_sort"], l_damp_style=c["l_damp"], dead=c["dead"],
        desc_act=c["desc_act"], damp_percent_in=0.01, group_size=128, nsamples=nsamples,
        W=W.numpy(), bias=bias.numpy(), X=np.stack([x.numpy() for x in xs]),
        H_raw=H_raw.numpy(),
        perm=(np.arange(n) if loop_out["perm"] is None else loop_out["perm"].numpy()).astype(np.int64),
        W_perm=rec["W_perm"].numpy(), L=loop_out["L"].numpy(), Xxt_damped=loop_out["Xxt_damped"].numpy(),
        Hinv_diag=rec["Hinv_diag"].numpy(),
        T=Ts, Q=Qs, A=As, B=Bs, dists=dists,
        Wq_loop=loop_out["Wq_loop"].numpy(), Losses=loop_out["Losses"].numpy(),
        Wq=wq.numpy(), scale=scale.numpy(), zero=zero.numpy(), g_idx=g_idx.numpy(),
        avg_loss=np.float64(avg_loss), damp_percent=np.float64(damp_percent),
        x_fwd=x_fwd.numpy(), y_fwd=y_fwd.numpy(),
    )
    return out


def main():
    def km(values, k, weights=None):
        v = np.asarray(values, dtype=np.float32).reshape(1, -1)
        T0 = c_oracle.kmeans_init(v, None if weights is None else np.asarray(weights, dtype=np.float64), k)
        return None, [float(t) for t in T0[0]]

    mods = ref_loader.load_reference(km)
    torch.set_num_threads(4)
    for c in CASES:
        out = run_case(c, *mods)
        path = os.path.join(HERE, f"{c['name']}.npz")
        np.savez_compressed(path, **out)
        best_k = int(np.argmin(out["dists"]))
        print(f"{c['name']}: dists={out['dists']} best_k={best_k} avg_loss={out['avg_loss']:.6g} "
              f"-> {os.path.getsize(path) / 1e6:.2f} MB")


if __name__ == "__main__":
    main()

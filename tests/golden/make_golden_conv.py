"""Golden vectors for an nn.Conv2d module: the reference flattens its weight to [out_channels, in_channels * kh * kw]
(gptq.py:80-81) and accumulates the Hessian over the unfolded patches (gptq.py:111-121).  Build container only:

    python tests/golden/make_golden_conv.py

The reference's own `GANQ(GPTQ)` object on a Conv2d, instrumented from the outside exactly like make_golden.py; T0 from the
oracle's k-means (kmeans1d is not installed: parity unpinned for T0).  Stored: W [O,C,kh,kw], bias, the calibration images,
H_raw, the permuted / flattened weight and diag(Hinv) the loop received, L, Xxt_damped, T_k, Q_k, dists, the 7-tuple's Wq (4-D),
avg_loss.
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

CASES = [dict(name="conv16x8x3x3_b4", out_ch=16, in_ch=8, k=3, pad=1, stride=1, hw=6, bits=4, K=3, nb=3, bsz=2, seed=21)]


def run_case(c, ganq_mod, gptq_mod, cfg_mod, NamedModule):
    g = torch.Generator().manual_seed(c["seed"])
    conv = torch.nn.Conv2d(c["in_ch"], c["out_ch"], c["k"], padding=c["pad"], stride=c["stride"], bias=True).half()
    with torch.no_grad():
        conv.weight.copy_((0.05 * torch.randn(conv.weight.shape, generator=g)).half())
        conv.bias.copy_((0.01 * torch.randn(c["out_ch"], generator=g)).half())
    xs = [(torch.randn(c["bsz"], c["in_ch"], c["hw"], c["hw"], generator=g) * (0.2 + torch.rand(1, c["in_ch"], 1, 1, generator=g))).half()
          for _ in range(c["nb"])]
    qcfg = cfg_mod.QuantizeConfig(bits=c["bits"], quant_method="ganq", format="fake", act_sort="asc", l_damp_style="ganq",
                                  dead="mean", desc_act=True, ganq_iterations=c["K"], group_size=128, damp_percent=0.01)
    q = ganq_mod.GANQ(NamedModule(conv, "conv", "model.layers.0.conv", 0), qcfg)
    q.quantizer.configure(perchannel=True)
    for x in xs:
        q.add_batch(x, None)
    H_raw, nsamples = q.H.clone(), q.nsamples
    m, n, K = q.rows, q.columns, c["K"]

    rec = dict(argmin=[], lstsq=[], loss=[])
    real_argmin, real_lstsq, real_loss = torch.argmin, torch.linalg.lstsq, ganq_mod.quad_loss_2
    real_init, real_loop = ganq_mod.GANQ._initialize_codebook_kmeans, ganq_mod.GANQ._perform_quantization_loop

    def init_wrap(self, Wp, Hinv, num_bits, device):
        rec["W_perm"], rec["Hinv_diag"] = Wp.clone(), torch.diagonal(Hinv).clone()
        rec["T0"] = real_init(self, Wp, Hinv, num_bits, device).clone()
        return rec["T0"].clone()

    def argmin_wrap(*a, **k):
        out = real_argmin(*a, **k)
        rec["argmin"].append(out.clone())
        return out

    def lstsq_wrap(A, B, *a, **k):
        out = real_lstsq(A, B, *a, **k)
        rec["lstsq"].append(out.solution.clone())
        return out

    def loss_wrap(Wm, Wq, G):
        out = real_loss(Wm, Wq, G)
        rec["loss"].append(float(out))
        return out

    def loop_wrap(self, Wp, Hinv, blocksize, perm=None, invperm=None):
        rec["perm"] = None if perm is None else perm.clone()
        out = real_loop(self, Wp, Hinv, blocksize, perm, invperm)
        rec["L"], rec["Xxt_damped"] = self.L.clone(), self.Xxt_damped.clone()
        return out

    ganq_mod.GANQ._initialize_codebook_kmeans, ganq_mod.GANQ._perform_quantization_loop = init_wrap, loop_wrap
    torch.argmin, torch.linalg.lstsq, ganq_mod.quad_loss_2 = argmin_wrap, lstsq_wrap, loss_wrap
    try:
        wq, scale, zero, g_idx, duration, avg_loss, damp_percent = q.quantize()
    finally:
        ganq_mod.GANQ._initialize_codebook_kmeans, ganq_mod.GANQ._perform_quantization_loop = real_init, real_loop
        torch.argmin, torch.linalg.lstsq, ganq_mod.quad_loss_2 = real_argmin, real_lstsq, real_loss
    assert len(rec["argmin"]) == K * n and wq.shape == conv.weight.shape
    Qs = np.zeros((K, m, n), dtype=np.uint8)
    for k in range(K):
        for step in range(n):
            Qs[k, :, n - 1 - step] = rec["argmin"][k * n + step].numpy().astype(np.uint8)
    Ts = np.stack([rec["T0"].numpy()] + [s.mT.squeeze(-2).numpy() for s in rec["lstsq"]]).astype(np.float32)
    return dict(out_ch=c["out_ch"], in_ch=c["in_ch"], k=c["k"], pad=c["pad"], stride=c["stride"], bits=c["bits"], K=K,
                W=conv.weight.data.numpy(), bias=conv.bias.data.numpy(), X=np.stack([x.numpy() for x in xs]), H_raw=H_raw.numpy(),
                nsamples=nsamples, perm=rec["perm"].numpy().astype(np.int64), W_perm=rec["W_perm"].numpy(),
                Hinv_diag=rec["Hinv_diag"].numpy(), L=rec["L"].numpy(), Xxt_damped=rec["Xxt_damped"].numpy(), T=Ts, Q=Qs,
                dists=np.array(rec["loss"], dtype=np.float64), Wq=wq.numpy(), g_idx=g_idx.numpy(), avg_loss=np.float64(avg_loss))


def main():
    def km(values, k, weights=None):
        v = np.asarray(values, dtype=np.float32).reshape(1, -1)
        T0 = c_oracle.kmeans_init(v, None if weights is None else np.asarray(weights, dtype=np.float64), k)
        return None, [float(t) for t in T0[0]]

    mods = ref_loader.load_reference(km)
    torch.set_num_threads(4)
    for c in CASES:
        out = run_case(c, *mods)
        path = os.path.join(HERE, "conv", c["name"] + ".npz")
        np.savez_compressed(path, **out)
        print(f"{c['name']}: [{out['W_perm'].shape}] dists={out['dists']} avg_loss={float(out['avg_loss']):.6g} -> {os.path.getsize(path) / 1e3:.0f} KB")


if __name__ == "__main__":
    main()

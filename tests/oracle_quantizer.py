"""TEST INFRASTRUCTURE: the reference's quantizer object restated on the CPU, for model-boundary parity tests.

`OracleGANQ` has the surface the looper / processor use of `ganq_amd.quantization.GANQ` (add_batch, quantize() -> 7-tuple,
free, fwd_counter, nsamples, columns, ganq_indices, ganq_codebook) but computes everything on the host with the CPU
oracle and the reference's own op sequence:
  Hessian      gptq.py:96-131     torch CPU (fp32 matmul of the scaled activations, as the reference does)
  prologue     gptq.py:259-319    torch CPU / LAPACK: dead columns, act_sort, ganq-style L, damping, cholesky ->
                                  cholesky_inverse -> cholesky(upper)
  codebook     ganq.py:423-438    oracle k-means with weights diag(Hinv)^-4
  loop         ganq.py:516-634    oracle/ganq_oracle.c (ganq_oracle_run_layer)
  outputs      ganq.py:633-646, gptq.py:324-375
`OracleProcessor` is the product's GPTQProcessor with this quantizer in the GANQ slot.  Nothing under ganq_amd/ imports
this file.
"""
import copy
import time

import numpy as np
import torch

from ganq_amd.looper.gptq_processor import GPTQProcessor
from ganq_amd.quantization.quantizer import Quantizer
from oracle import c_oracle


class OracleSolver:
    """CPU stand-in for ganq_amd.distributed.HipSolver (same two methods): the collective logic of
    run_layer_row_sharded is exercised under gloo with the oracle serving the per-slice loop"""

    def run_layer_rows(self, W, H, L, T0, K, alias_q, rcond):
        tr = c_oracle.run_layer_trace(W.numpy(), H.numpy(), L.numpy(), T0.numpy(), K, rcond)
        return dict(T_all=torch.from_numpy(tr["T_all"]), loss_rows_all=torch.from_numpy(tr["loss_rows_all"]),
                    Q_last=torch.from_numpy(tr["Q_all"][K - 1].copy()), Q_all=None if alias_q else torch.from_numpy(tr["Q_all"]))

    def select_best(self, loss_rows_all):
        d = loss_rows_all.sum(dim=1)  # the HIP library sums in its own fixed order; any fixed order serves the CPU test
        best, bk = float("inf"), -1
        for k in range(d.shape[0]):  # ganq.py:625: strict <, first minimum
            if float(d[k]) < best:
                best, bk = float(d[k]), k
        return d, torch.tensor(bk, dtype=torch.int32)


class OracleGANQ:
    def __init__(self, module, qcfg):
        self.module = module.module
        self.qcfg = qcfg
        self.device = self.module.weight.device
        self.W = self.module.weight.data.detach().float().cpu().clone()
        self.rows, self.columns = self.W.shape
        self.nsamples = 0
        self.fwd_counter = 0
        self.H = torch.zeros((self.columns, self.columns), dtype=torch.float32)
        self.quantizer = Quantizer(qcfg=qcfg, name=module.name)
        self.iterations = qcfg.ganq_iterations
        self.ganq_indices = self.ganq_codebook = self.ganq_outliers = None
        self.row_dist = None  # set by the looper: rows over the ranks (ganq_amd.distributed.run_layer_row_sharded)
        self.ganq_stats = {}

    @property
    def hessian(self):
        return self.H

    # gptq.py:88-131
    def add_batch(self, inp, out):
        self.fwd_counter += 1
        x = inp.detach().to("cpu")
        if x.dim() == 2:
            x = x.unsqueeze(0)
        batch = x.shape[0]
        x = x.reshape(-1, x.shape[-1]).t()
        self.H *= self.nsamples / (self.nsamples + batch)
        self.nsamples += batch
        x = (2.0 / self.nsamples) ** 0.5 * x.float()
        self.H += x.matmul(x.t())

    def _prologue_key(self):
        return id(self)  # never shares a prologue

    def follow(self, leader):
        raise RuntimeError("OracleGANQ: run the looper with share_group_hessian=False")

    @torch.no_grad()
    def quantize(self, blocksize=128):
        start = time.time()
        c = self.qcfg
        W, H = self.W, self.H
        self.quantizer.find_params(W, weight=True)
        dead = torch.diag(H) == 0
        H[dead, dead] = 1
        if c.dead == "zero":
            W[:, dead] = 0
        else:
            W[:, dead] = torch.mean(W[:, ~dead], dim=1, keepdim=True)
        perm = invperm = None
        if c.act_sort != "none":
            perm = torch.argsort(torch.diag(H), descending=c.act_sort == "desc")
            W = W[:, perm].contiguous()
            H = H[perm][:, perm].contiguous()
            invperm = torch.argsort(perm)
        L = None
        if c.l_damp_style == "ganq":
            offset = (torch.sum(torch.abs(H), dim=1) - 2 * torch.diag(H)).clamp(min=1e-8)
            L = torch.linalg.cholesky(H + torch.diag(offset))
        damp_percent = c.damp_percent
        damp = damp_percent * torch.mean(torch.diag(H))
        diag = torch.arange(self.columns)
        H[diag, diag] += damp
        Xxt_damped = H.clone()
        Lg = torch.linalg.cholesky(H)
        if c.l_damp_style == "gptq":
            L = Lg.clone()
        Hinv = torch.linalg.cholesky(torch.cholesky_inverse(Lg), upper=True)
        hd = torch.diagonal(Hinv)
        V = 2 ** c.bits
        weights = (hd ** (-4)).double().numpy()
        Wn = W.numpy()
        alias = bool(getattr(c, "ganq_reference_q_alias", True))
        if self.row_dist is not None and self.row_dist.world > 1:
            from ganq_amd import distributed as gdist

            Tt, Qt, _, _ = gdist.run_layer_row_sharded(
                W, Xxt_damped, L, None, self.iterations, alias_q=alias, dist=self.row_dist, solver=OracleSolver(),
                t0_fn=lambda Wr: torch.from_numpy(c_oracle.kmeans_init(Wr.numpy(), weights, V)))
            T, Q = Tt.numpy(), Qt.numpy()
        else:
            T0 = c_oracle.kmeans_init(Wn, weights, V)
            T, Q, dists, best_k = c_oracle.run_layer(Wn, Xxt_damped.numpy(), L.numpy(), T0, self.iterations, alias_q=alias)
        Wq, Losses = c_oracle.dequant_losses(Wn, T, Q, hd.numpy())
        avg_loss = float(Losses.astype(np.float64).sum()) / self.nsamples
        group_size = c.group_size if c.group_size != -1 else self.columns
        g_idx = (torch.arange(self.columns) // group_size).to(torch.int32)
        Wq_t, Q_t = torch.from_numpy(Wq), torch.from_numpy(Q)
        if c.desc_act and invperm is not None:
            Wq_t, Q_t, g_idx = Wq_t[:, invperm], Q_t[:, invperm].contiguous(), g_idx[invperm]
        self.ganq_indices = Q_t.to(self.device)
        self.ganq_codebook = torch.from_numpy(T).to(self.device)
        wq = Wq_t.reshape(self.module.weight.shape).type_as(self.module.weight.data).to(self.device)
        self.quantizer.find_params(W, weight=True)
        return (wq, self.quantizer.scale, self.quantizer.zero, g_idx.to(self.device), time.time() - start, avg_loss,
                damp_percent)

    def free(self):
        self.W = self.H = None


class OracleProcessor(GPTQProcessor):
    """GPTQProcessor with the CPU oracle in the quantizer slot (gptq_processor.py:86-87)"""

    def preprocess(self, module, buffered_fwd=False):
        if self.qcfg.dynamic_get(layer_name=module.full_name) is False:
            return
        tmp = OracleGANQ(module, copy.deepcopy(self.qcfg))
        tmp.quantizer.configure(perchannel=True)
        self.tasks[module.name] = tmp


def reference_prologue_ops(self, W, H):
    """TEST INFRASTRUCTURE (moved out of the product in round 4): the reference's own prologue op sequence (gptq.py:267-316) on
    torch.linalg, with the signature of `ganq_amd.quantization.GPTQ._prologue_hip` -- the checker the HIP prologue is compared
    with.  `self` is a product GPTQ/GANQ object; Hinv is returned as the full upper factor, as the reference does."""
    dead = torch.diag(H) == 0
    H[dead, dead] = 1
    if self.qcfg.dead == "zero":
        W[:, dead] = 0
    elif self.qcfg.dead == "mean":
        W[:, dead] = torch.mean(W[:, ~dead], dim=1, keepdim=True)
    else:
        assert False, f"Unknown dead mode: {self.qcfg.dead}"

    perm = None
    invperm = None
    if self.qcfg.act_sort != "none":
        assert self.qcfg.act_sort in ["asc", "desc"]
        perm = torch.argsort(torch.diag(H), descending=self.qcfg.act_sort == "desc")
        W = W[:, perm].contiguous()
        H = H[perm][:, perm].contiguous()
        invperm = torch.argsort(perm)

    self.Xxt = H.clone()  # undamped
    if self.qcfg.l_damp_style == "ganq":
        offset = (torch.sum(torch.abs(H), dim=1) - 2 * torch.diag(H)).clamp(min=1e-8)
        self.L = torch.linalg.cholesky(H + torch.diag(offset))

    damp_percent = self.qcfg.damp_percent
    Hinv = None
    while 1 > damp_percent > 0:
        try:
            damp = damp_percent * torch.mean(torch.diag(H))
            diag = torch.arange(self.columns, device=self.device)
            H[diag, diag] += damp
            self.Xxt_damped = H.clone()
            L = torch.linalg.cholesky(H)
            if self.qcfg.l_damp_style == "gptq":
                self.L = L.clone()
            Hinv = torch.linalg.cholesky(torch.cholesky_inverse(L), upper=True)
            break
        except torch._C._LinAlgError as e:
            if self.qcfg.damp_auto_increment != 0:
                damp_percent += self.qcfg.damp_auto_increment
            else:
                raise e
    return W, dead, perm, invperm, Hinv, damp_percent


def use_reference_prologue(q):
    """make the product quantizer object `q` run the reference's prologue op sequence (above) instead of the HIP prologue --
    A/B runs in the tests only; the product has no such switch"""
    import types

    q._prologue_hip = types.MethodType(reference_prologue_ops, q)
    return q

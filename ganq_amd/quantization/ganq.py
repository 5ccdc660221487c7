"""GANQ quantizer whose `_perform_quantization_loop` runs on the MI355X through libganq_hip.so.

Drop-in for `gptqmodel.quantization.ganq.GANQ` (ganq.py:397-646): same constructor, same override point and
return contract `(Wq, Losses, scale: list, zero: list)`, same config fields (`bits`, `ganq_iterations`, ...).
Additionally keeps what the reference throws away (ganq.py:633-634,646): after `quantize()`,
`self.ganq_indices` (uint8 [m,n], original column order) and `self.ganq_codebook` (fp32 [m,V]) hold the
assignment and the per-row codebook that `GanqHipQuantLinear` packs.

`qcfg.ganq_outlier_ratio > 0` adds the outlier split of the paper (section 3.3, Appendix A Algorithm 2; not in the
reference code): the row-wise tails of W are taken out on the device (`_lib.outlier_split`), GANQ runs on the rest, the
returned weight is `T.gather(1, Q) + W_sparse`, and `self.ganq_outliers` = (rowptr, cols, vals) is what the layer keeps.
"""
import time

import torch

from .. import _lib
from .gptq import GPTQ


def _hinv_diag(Hinv):
    """diag(Hinv) whether the prologue handed over the upper factor (reference op sequence) or only its diagonal."""
    return Hinv if Hinv.dim() == 1 else torch.diagonal(Hinv)


class GANQ(GPTQ):
    """Quantize following "GANQ: GPU-Adaptive Layer-Wise LUT-Based Non-Uniform Quantization" (arXiv 2501.12956)."""

    def __init__(self, module, qcfg=None):
        super().__init__(module, qcfg)
        self.iterations = getattr(self.qcfg, "ganq_iterations", 5)
        self.ganq_indices = None
        self.ganq_codebook = None
        self.ganq_outliers = None  # CSR (rowptr int32 [m+1], cols int32 [nnz], vals fp32 [nnz]) when the split is on
        self.ganq_stats = {}
        # set by the looper when torch.distributed runs this module's rows over several GPUs (SURVEY 8(e) axis 2): a
        # ganq_amd.distributed.Dist with world > 1.  Every rank must then call quantize() on the same statistics.
        self.row_dist = None
        # cleared by the looper for the modules of a group it quantizes side by side on one GPU (streams): a helper workgroup
        # of the S-solve may then not be resident with its tile (GANQ_FLAG_NO_HELPERS; same bits either way)
        self.solve_helpers = True

    def _needs_only_hinv_diag(self) -> bool:
        return True

    def _initialize_codebook_kmeans(self, W, Hinv, num_bits, device):
        """ganq.py:423-438: weighted 1-D k-means per row, column weight diag(Hinv)^-4 (computed in fp32 as the
        reference does before handing it to kmeans1d, which works in double)."""
        exp = 4
        col_weight = (_hinv_diag(Hinv) ** (-exp)).double()
        return _lib.kmeans_init(W, col_weight, 2 ** num_bits)

    @torch.no_grad()
    def _perform_quantization_loop(self, W, Hinv, blocksize, perm=None, invperm=None):
        """Algorithm 1 of the paper as implemented by ganq.py:456-646, on the device:
        T0 = k-means;  K x (S-solve, T-update, loss);  best-of-K;  Wq, Losses."""
        num_bits = self.qcfg.bits
        V = 2 ** num_bits
        if num_bits not in (2, 3, 4):
            raise NotImplementedError(f"GANQ HIP path implements bits in (2, 3, 4); got bits={num_bits}")

        scale, zero = [], []
        if self.qcfg.group_size != -1:  # "Not supported, here for compatibility" (ganq.py:489-495)
            self.quantizer.find_params(W, weight=True)
            scale.append(self.quantizer.scale)
            zero.append(self.quantizer.zero)

        t0 = time.perf_counter()
        ratio = float(getattr(self.qcfg, "ganq_outlier_ratio", 0.0) or 0.0)
        sparse = None
        if ratio > 0.0:  # W becomes W_dense in place (it is this call's private fp32 copy, gptq.py:77-86)
            rowptr, cols, vals, _ = _lib.outlier_split(W, ratio)
            sparse = (rowptr, cols, vals)
        alias = bool(getattr(self.qcfg, "ganq_reference_q_alias", True))
        rd = self.row_dist
        if rd is not None and (rd.world > 1 or getattr(self, "force_row_path", False)):  # (forced at world 1: bench phase timing)
            # rows of W are independent in the codebook initialisation, the S-solve and the T-update; only best-of-K
            # looks at all of them: each rank clusters and iterates its own rows, one exchange of K x m row losses
            # decides, one all-gather each brings the chosen codebook rows and the indices to every rank
            from .. import distributed as gdist

            stats = {"timing": bool(getattr(self, "time_collectives", False))}
            T, Q, dists, best_k = gdist.run_layer_row_sharded(
                W, self.Xxt_damped, self.L, None, self.iterations, alias_q=alias, dist=rd, stats=stats,
                turn=getattr(self, "_collective_turn", None), solver=gdist.HipSolver(helpers=self.solve_helpers), V=V,
                t0_fn=lambda W_rows: self._initialize_codebook_kmeans(W_rows, Hinv, num_bits, W.device))
            self.ganq_stats.update({k: v for k, v in stats.items() if k != "timing"})
        else:
            T0 = self._initialize_codebook_kmeans(W, Hinv, num_bits, W.device)
            assert T0.shape == (W.shape[0], V)
            T, Q, dists, best_k = _lib.run_layer(W, self.Xxt_damped, self.L, T0, self.iterations, alias_q=alias,
                                                 helpers=self.solve_helpers)
        Wq, Losses = _lib.dequant_losses(W, T, Q, _hinv_diag(Hinv).contiguous())
        if sparse is not None:  # effective weight = dequantised dense part + the exact outliers
            rowptr, cols, vals = sparse
            rows = torch.repeat_interleave(torch.arange(W.shape[0], device=W.device), (rowptr[1:] - rowptr[:-1]).long())
            Wq.index_put_((rows, cols.long()), vals, accumulate=True)
            self.ganq_outliers = sparse
        self.ganq_indices = Q          # permuted column order until quantize() un-permutes it
        self.ganq_codebook = T
        self.ganq_stats.update({"dists": dists, "best_k": best_k, "enqueue_s": time.perf_counter() - t0})

        if not scale:  # "Unused, compatibility with interface" (ganq.py:640-644)
            self.quantizer.find_params(W, weight=True)
            scale.append(self.quantizer.scale)
            zero.append(self.quantizer.zero)
        return Wq, Losses, scale, zero

    def _unpermute_state(self, invperm):
        # gptq.py:341-343 un-permutes Wq only when desc_act is set; the indices follow the same rule so that
        # T.gather(1, Q) always equals the returned weight
        if invperm is not None and self.ganq_indices is not None:
            self.ganq_indices = self.ganq_indices[:, invperm].contiguous()
            if self.ganq_outliers is not None:  # permuted column c is original column perm[c]; keep rows ascending
                rowptr, cols, vals = self.ganq_outliers
                perm = torch.argsort(invperm)
                n = invperm.numel()
                rows = torch.repeat_interleave(torch.arange(rowptr.numel() - 1, device=cols.device),
                                               (rowptr[1:] - rowptr[:-1]).long())
                key, order = torch.sort(rows * n + perm[cols.long()])
                self.ganq_outliers = (rowptr, (key % n).to(torch.int32), vals[order])

    def make_quantized_weight(self, Q, T):
        return T.gather(1, Q.long())

    def free(self):
        super().free()


__all__ = ["GANQ"]

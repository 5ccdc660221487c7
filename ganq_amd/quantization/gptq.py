"""GPTQ base class for the GANQ path: weight clone, Hessian accumulation, prologue (dead columns, act_sort,
ganq-/gptq-style Cholesky, damping, inverse-Cholesky) and epilogue of `quantize()`.

Host-side mirror of gptqmodel/quantization/gptq.py:42-393 with the same public surface
(`GPTQ(module, qcfg)`, `add_batch`, `quantize() -> 7-tuple`, `hf_quantize`, `free`, override point
`_perform_quantization_loop`).  What differs from the reference:
  * the Hessian update of `process_batch` (gptq.py:122-131) runs in the HIP kernel ganq_hessian_accum;
  * everything lives on the GPU; there is no CPU fallback (the module must be on a cuda device);
  * GPTQ's own uniform-grid column loop (gptq.py:164-236) is NOT part of this path: only the GANQ subclass
    implements `_perform_quantization_loop`.
The prologue is three passes over the matrices (csrc/prologue.hip) and two in-place factorisations by csrc/cholesky.hip;
there is no library-backed alternative in the product (the reference's own op sequence on torch.linalg lives in
tests/oracle_quantizer.py as a checker).
"""
import math
import time
from typing import Optional

import torch
import torch.nn as nn

from .. import _lib
from ..looper.named_module import NamedModule
from .config import QuantizeConfig
from .quantizer import HF_OPTIMUM, Quantizer


_SIDE_STREAMS = {}
# Hessian staging buffers alive per device (see GPTQ.__init__): each is stage_tokens x in_features activations (134 MB at
# n = 4096, 470 MB at 14336).  A group of E mixture-of-experts modules would hold E of them through its forward passes; beyond
# this many live buffers on a device a task accumulates batch by batch instead (same H up to fp32 summation order).
_STAGE_LIVE = {}   # device index -> weakref.WeakSet of the live buffers' holders: a task that is dropped without free() /
                   # end_of_calibration() (an exception in a forward pass, an abandoned looper) gives its slot back when it dies
STAGE_MAX_LIVE = 8


class _StageBuffer:
    """holder of one staging tensor (torch tensors are not hashable members of a WeakSet by identity semantics we want)"""

    __slots__ = ("tensor", "__weakref__")

    def __init__(self, tensor):
        self.tensor = tensor


def _side_stream(device) -> "torch.cuda.Stream":
    """one extra stream per device for the part of the prologue that runs beside the main stream"""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if idx not in _SIDE_STREAMS:
        _SIDE_STREAMS[idx] = torch.cuda.Stream(device=idx)
    return _SIDE_STREAMS[idx]


def _is_conv1d(module) -> bool:
    return type(module).__name__ == "Conv1D"  # transformers.pytorch_utils.Conv1D, without importing transformers


class GPTQ:
    def __init__(self, module: nn.Module, qcfg: Optional[QuantizeConfig] = None):
        if isinstance(module, NamedModule):
            self.module = module.module
            name = module.name
        else:
            name = HF_OPTIMUM
            self.module = module
        if not isinstance(self.module, (nn.Linear, nn.Conv2d)) and not _is_conv1d(self.module):
            raise NotImplementedError(f"GANQ HIP path supports nn.Linear / Conv1D / nn.Conv2d modules, got {type(self.module)}")
        self.qcfg = qcfg if qcfg else QuantizeConfig()
        self.device = self.module.weight.device
        if self.device.type != "cuda":
            raise _lib.GanqHipError(f"module `{name}` lives on {self.device}: the HIP quantizer needs it on the GPU "
                                    f"(there is no CPU fallback)")
        self.module_copy = self._clone_module()
        self.rows, self.columns = self.module_copy.shape[0], self.module_copy.shape[1]
        self.nsamples = 0
        self.quantizer = self.create_quantizer(name=name)
        self.fwd_inputs_buffered = False
        self.fwd_inputs_buffered_data = []
        self.fwd_counter = 0
        # Hessian staging (not in the reference): fp16 / bf16 calibration batches are copied into one device buffer and
        # go to the HIP kernel several at a time.  H <- H N/(N+b) + (2/(N+b)) X^T X telescopes over a group of batches to
        # H N0/(N0+B) + (2/(N0+B)) sum_i X_i^T X_i, so a group is ONE kernel call with B = its sequences: H (64 MB at
        # n = 4096) is read-modify-written once per group instead of once per sequence, and the kernel's main loop is
        # eight times longer per tile.  0 turns the staging off.
        self.hessian_stage_tokens = int(getattr(self.qcfg, "ganq_hessian_stage_tokens", 16384) or 0)
        self._stage = None        # [capacity, columns] in the activations' dtype -- or, round 4, TRANSPOSED [columns, capacity]
        self._stage_t = False     # (layers the transposed Hessian kernel serves: the staging copy transposes, _lib.hessian_stage_t)
        self._stage_holder = None
        self._stage_rows = 0
        self._stage_seqs = 0

    def create_quantizer(self, name: str) -> Quantizer:
        return Quantizer(qcfg=self.qcfg, name=name)

    def shape(self):
        return self.module.weight.shape if hasattr(self, "module") else (0, 0)

    def _clone_module(self):
        clone = self.module.weight.data.clone()
        if isinstance(self.module, nn.Conv2d):
            clone = clone.flatten(1)  # gptq.py:80-81: [out_channels, in_channels * kh * kw]
        if _is_conv1d(self.module):
            clone = clone.t()
        return clone.float().contiguous()

    # ---- gptq.py:88-131 --------------------------------------------------------------------------------
    def add_batch(self, inp, out):
        self.fwd_counter += 1
        if self.fwd_inputs_buffered:
            self.fwd_inputs_buffered_data.append(inp.to(device="cpu"))
        else:
            self.process_batch(inp)

    def process_batch(self, inp):
        inp = inp.to(device=self.device)
        if len(inp.shape) == 2:
            inp = inp.unsqueeze(0)
        batch = inp.shape[0]  # sequences, not tokens (gptq.py:104)
        if isinstance(self.module, nn.Conv2d):
            # gptq.py:111-121: the patches every output position sees (nn.Unfold), one "token" per (image, position):
            # [B, C kh kw, L] -> [B L, C kh kw] in the layout the Hessian kernel streams (torch: tensor plumbing)
            inp = torch.nn.functional.unfold(inp, self.module.kernel_size, dilation=self.module.dilation,
                                             padding=self.module.padding, stride=self.module.stride)
            inp = inp.permute(0, 2, 1).reshape(-1, inp.shape[1]).contiguous()
        if len(inp.shape) == 3:
            inp = inp.reshape((-1, inp.shape[-1]))
        if not hasattr(self, "H"):
            self.H = torch.zeros((self.columns, self.columns), device=self.device)
        if inp.dtype in (torch.float16, torch.bfloat16):
            rows = inp.shape[0]
            if self.hessian_stage_tokens > 0 and rows < self.hessian_stage_tokens:
                if self._stage is not None and (self._stage.dtype != inp.dtype or self._stage_rows + rows > self.hessian_stage_tokens):
                    self._flush_stage()
                if (self._stage is None or self._stage.dtype != inp.dtype) and not self._stage_acquire(inp.dtype):
                    _lib.hessian_accum(self.H, inp, self.nsamples, batch)  # the device already holds STAGE_MAX_LIVE buffers
                    self.nsamples += batch
                    return
                if self._stage_t and (rows % 8 or self._stage_rows % 8):
                    # (the transposing copy moves whole groups of 8 tokens; a ragged batch ends the group and goes alone)
                    self._flush_stage()
                    _lib.hessian_accum(self.H, inp, self.nsamples, batch)
                    self.nsamples += batch
                    return
                if self._stage_t:
                    _lib.hessian_stage_t(self._stage, inp, self._stage_rows)
                else:
                    self._stage[self._stage_rows:self._stage_rows + rows].copy_(inp)
                self._stage_rows += rows
                self._stage_seqs += batch
                self.nsamples += batch
                return
            self._flush_stage()
            _lib.hessian_accum(self.H, inp, self.nsamples, batch)
        else:
            self._flush_stage()
            # fp32 activations: exact fp32 products on the fp32 matrix cores
            x = inp.float().contiguous()
            total = self.nsamples + batch
            upd = _lib.matmul_f32(x.t().contiguous(), x)
            self.H.mul_(self.nsamples / total).add_(upd, alpha=2.0 / total)
        self.nsamples += batch

    def _stage_acquire(self, dtype) -> bool:
        """allocate this task's staging buffer unless the device already holds STAGE_MAX_LIVE of them"""
        import weakref

        self._stage_release()
        key = self.device.index if self.device.index is not None else torch.cuda.current_device()
        live = _STAGE_LIVE.setdefault(key, weakref.WeakSet())
        if len(live) >= STAGE_MAX_LIVE:
            return False
        self._stage_t = _lib.hessian_t_supported(self.columns, self.hessian_stage_tokens, self.device)
        shape = (self.columns, self.hessian_stage_tokens) if self._stage_t else (self.hessian_stage_tokens, self.columns)
        self._stage_holder = _StageBuffer(torch.empty(shape, dtype=dtype, device=self.device))
        self._stage = self._stage_holder.tensor
        live.add(self._stage_holder)
        return True

    def _stage_release(self):
        if self._stage is not None:
            key = self.device.index if self.device.index is not None else torch.cuda.current_device()
            live = _STAGE_LIVE.get(key)
            if live is not None and self._stage_holder is not None:
                live.discard(self._stage_holder)
            self._stage = None
            self._stage_holder = None

    def end_of_calibration(self):
        """The looper calls this when the forward passes of the module's group are over: the staged batches go to the
        Hessian kernel and the staging buffer is freed HERE, not at quantize() -- between the two the other modules of the
        group are quantized, and a mixture-of-experts group would otherwise hold one buffer per expert all that time."""
        if hasattr(self, "H"):
            self._flush_stage()
        self._stage_release()

    def _flush_stage(self):
        """send the staged calibration batches to the Hessian kernel as one group (see __init__)"""
        if self._stage_rows:
            before = self.nsamples - self._stage_seqs  # self.nsamples already counts the staged sequences
            if self._stage_t:
                rows = -(-self._stage_rows // 32) * 32  # the kernel walks slices of 32 tokens: a ragged last one is zero-filled
                if rows != self._stage_rows:
                    self._stage[:, self._stage_rows:rows].zero_()
                _lib.hessian_accum_t(self.H, self._stage, rows, before, self._stage_seqs)
            else:
                _lib.hessian_accum(self.H, self._stage[:self._stage_rows], before, self._stage_seqs)
            self._stage_rows = 0
            self._stage_seqs = 0

    @property
    def hessian(self):
        """the accumulated Hessian with every staged batch applied"""
        self._flush_stage()
        return self.H

    # ---- HF/optimum entry (gptq.py:133-162) --------------------------------------------------------------
    def fasterquant(self, blocksize=128, percdamp=0.01, damp_auto_increment=0.0015, group_size=-1, actorder=False,
                    static_groups=False):
        return self.hf_quantize(blocksize, percdamp, damp_auto_increment, group_size, actorder, static_groups)

    def hf_quantize(self, blocksize=128, percdamp=0.01, damp_auto_increment=0.0015, group_size=-1, actorder=False,
                    static_groups=False):
        self.qcfg.group_size = group_size
        self.qcfg.damp_percent = percdamp
        self.qcfg.damp_auto_increment = damp_auto_increment
        self.qcfg.desc_act = actorder
        self.qcfg.static_groups = static_groups
        (Q, scale, zero, g_idx, duration, avg_loss, damp_percent) = self.quantize(blocksize=blocksize)
        self.module.weight.data = Q
        return scale, zero, g_idx, duration, avg_loss, damp_percent

    def _perform_quantization_loop(self, W, Hinv, blocksize, perm=None, invperm=None):
        raise NotImplementedError("the uniform-grid GPTQ column loop (gptq.py:164-236) is outside this path; "
                                  "use ganq_amd.quantization.GANQ")

    # ---- gptq.py:238-375 ---------------------------------------------------------------------------------
    # ---- not in the reference: modules of one looper group (q/k/v, gate/up) see the same inputs, hence the same
    #      Hessian, permutation, factor and damping; the reference recomputes all of it per module --------------------
    def follow(self, leader: "GPTQ"):
        """Take the Hessian statistics and the prologue from `leader` (same in_features, same calibration inputs,
        quantized before this module) instead of accumulating and factoring an identical copy."""
        if leader.columns != self.columns:
            raise ValueError("follow(): the modules of a group must have the same in_features")
        self._group_leader = leader
        self._leader_prologue = None  # filled in by the leader's quantize()
        leader._followers = getattr(leader, "_followers", []) + [self]

    def _prologue_key(self):
        c = self.qcfg
        return (self.columns, c.act_sort, c.l_damp_style, c.damp_percent, c.damp_auto_increment, c.dead)

    def _needs_only_hinv_diag(self) -> bool:
        """True when the quantization loop reads nothing of Hinv but its diagonal (GANQ does; see ganq.py)."""
        return False

    @staticmethod
    def _hip_cholesky(H):
        from .. import _lib
        return _lib.cholesky(H)

    def _prologue_hip(self, W, H):
        """gptq.py:267-308 on the device in three passes over the matrices (csrc/prologue.hip) plus n-vector steps: one read
        of H for its diagonal and absolute row sums; dead columns, act_sort permutation, ganq-style offset and damping decided
        on n-vectors; one gather that writes `Xxt_damped`, the ganq-style factorisation input and the index-reversed damped
        matrix (whose factor yields diag(Hinv): U = P L'^-1 P with P H P = L' L'^T, so diag(U)[i] = 1 / L'[n-1-i][n-1-i])
        directly; one gather of W.  The two factorisations run in place on two streams.  The reference's undamped copy `Xxt`
        (gptq.py:288) is not kept: nothing on this path reads it."""
        c = self.qcfg
        n = self.columns
        dev = H.device
        diag, rowabs = _lib.prologue_rowstats(H)
        dead = diag == 0                                            # gptq.py:267
        diag_fixed = torch.where(dead, torch.ones_like(diag), diag)  # gptq.py:268: H[dead, dead] = 1
        rowabs = rowabs + dead.to(rowabs.dtype)                     # ... which adds |1| to those rows' sums
        if c.dead not in ("zero", "mean"):
            raise AssertionError(f"Unknown dead mode: {c.dead}")
        perm = invperm = None
        if c.act_sort != "none":
            assert c.act_sort in ["asc", "desc"]
            perm = torch.argsort(diag_fixed, descending=c.act_sort == "desc")  # gptq.py:282
            invperm = torch.argsort(perm)
        W = _lib.prologue_weights(W, perm, dead, mean_fill=c.dead == "mean")
        ganq_style = c.l_damp_style == "ganq"
        offset = None
        if ganq_style:                                              # gptq.py:289-291
            offset = (rowabs - 2 * diag_fixed).clamp(min=1e-8)
            offset = offset if perm is None else offset[perm]
        mean_diag = torch.mean(diag_fixed)                          # gptq.py:296: damp = damp_percent * mean(diag(H))
        damp_percent = c.damp_percent
        total_damp = torch.zeros((), dtype=torch.float32, device=dev)
        cur, side = torch.cuda.current_stream(dev), _side_stream(dev)
        pending = None  # the ganq-style factor, running on the side stream
        Hinv = None
        first = True
        while 1 > damp_percent > 0:
            # every retry adds damp_percent * mean(diag) of the ALREADY damped matrix on top (gptq.py:294-316)
            total_damp = total_damp + damp_percent * (mean_diag + total_damp)
            add = total_damp.expand(n).contiguous()
            outs = [(add, False), (add, True)]                      # Xxt_damped; its index-reversed copy
            if ganq_style and first:
                outs.append((offset, False))                        # H + diag(offset), undamped (gptq.py:291)
            mats = _lib.prologue_gather(H, perm, diag_fixed, outs)
            self.Xxt_damped, Hflip = mats[0], mats[1]
            if ganq_style and first:
                A1 = mats[2]
                side.wait_stream(cur)
                A1.record_stream(side)
                with torch.cuda.stream(side):
                    L1, info1 = _lib.cholesky_inplace(A1, check=False)
                L1.record_stream(cur)
                pending = (L1, info1)
            first = False
            try:
                if not ganq_style:
                    self.L = _lib.cholesky(self.Xxt_damped)          # gptq-style: the factor of the damped matrix (a copy)
                Lr = _lib.cholesky_inplace(Hflip)
                Hinv = torch.flip(1.0 / torch.diagonal(Lr), dims=(0,)).contiguous()  # 1-D: the diagonal only
                break
            except torch._C._LinAlgError as e:
                if c.damp_auto_increment != 0:
                    damp_percent += c.damp_auto_increment
                else:
                    raise e
        if pending is not None:
            L1, info1 = pending
            cur.wait_stream(side)
            if int(info1):
                raise torch.linalg.LinAlgError("ganq_cholesky: H + diag(offset) is not positive-definite "
                                               f"(leading minor of order {int(info1)})")
            self.L = L1
        self.Xxt = None
        return W, dead, perm, invperm, Hinv, damp_percent

    @torch.inference_mode()
    def quantize(self, blocksize=128):
        start = time.time()
        for inp in self.fwd_inputs_buffered_data:
            self.process_batch(inp)
        self.fwd_inputs_buffered_data = []
        self.end_of_calibration()

        if self.module_copy is None:
            W = self._clone_module()
        else:
            W = self.module_copy
            self.module_copy = None
        if not hasattr(self, "H") and getattr(self, "_leader_prologue", None) is None:
            raise RuntimeError("quantize() called before any add_batch(): no calibration activations were seen")

        self.quantizer.find_params(W, weight=True)

        cached = getattr(self, "_leader_prologue", None)
        if cached is not None and cached["key"] == self._prologue_key():
            dead, perm, invperm = cached["dead"], cached["perm"], cached["invperm"]
            W = _lib.prologue_weights(W, perm, dead, mean_fill=self.qcfg.dead == "mean")  # as the leader's own weights
            self.Xxt, self.L, self.Xxt_damped = cached["Xxt"], cached["L"], cached["Xxt_damped"]
            Hinv, damp_percent = cached["Hinv"], cached["damp_percent"]
            self.nsamples = cached["nsamples"]
        else:
            if not hasattr(self, "H"):
                raise RuntimeError("quantize(): this module follows a leader that has not been quantized (yet)")
            H = self.H
            del self.H
            if not self._needs_only_hinv_diag():
                raise NotImplementedError("the HIP prologue yields diag(Hinv) only (all the GANQ loop reads of it, ganq.py:427-429,"
                                          "637-638); a quantizer that needs the full inverse factor is outside this path")
            W, dead, perm, invperm, Hinv, damp_percent = self._prologue_hip(W, H)
            if not (0 < damp_percent < 1):
                raise ValueError(f"Quantization: `damp_percent` must between 0 and 1. current is {damp_percent}")
            followers = getattr(self, "_followers", [])
            if followers:
                shared = {"key": self._prologue_key(), "dead": dead, "perm": perm, "invperm": invperm, "Xxt": self.Xxt,
                          "L": self.L, "Xxt_damped": self.Xxt_damped, "Hinv": Hinv, "damp_percent": damp_percent,
                          "nsamples": self.nsamples}
                for f in followers:
                    f._leader_prologue = shared
                # the followers can start from here (ModuleLooper runs them on side streams beside this module's own loop)
                hook = getattr(self, "_on_prologue_shared", None)
                if hook is not None:
                    hook()

        Q, Losses, scale, zero = self._perform_quantization_loop(W, Hinv, blocksize, perm, invperm)

        # the reference synchronises the device here (gptq.py:324); this module's work is all on the current stream, and a
        # device-wide wait would make a follower on a side stream wait for its leader's whole loop
        torch.cuda.current_stream(self.device).synchronize()
        avg_loss = torch.sum(Losses).item() / self.nsamples
        if math.isnan(avg_loss):
            raise ValueError("Quantization: Failed due to `NaN` loss")

        group_size = self.qcfg.group_size if self.qcfg.group_size != -1 else self.columns
        if self.qcfg.static_groups and self.qcfg.desc_act:
            g_idx = (perm // group_size).to(dtype=torch.int32)
        else:
            g_idx = (torch.arange(self.columns, device=Q.device) // group_size).to(dtype=torch.int32)
        if self.qcfg.desc_act:
            # with act_sort == "none" the reference would index with invperm = None; the permutation is the identity
            if invperm is not None:
                Q = Q[:, invperm]
                g_idx = g_idx[invperm]
        self._unpermute_state(invperm if self.qcfg.desc_act else None)

        if _is_conv1d(self.module):
            Q = Q.t()
        if Q.shape != self.module.weight.shape:
            Q = Q.reshape(self.module.weight.shape).type_as(self.module.weight.data)
        else:
            Q = Q.type_as(self.module.weight.data)
        Q = Q.to(device=self.device)

        if scale == []:
            scale.append(self.quantizer.scale)
            zero.append(self.quantizer.zero)
        scale = torch.cat(scale, dim=1)
        zero = torch.cat(zero, dim=1)
        duration = time.time() - start
        return Q, scale, zero, g_idx, duration, avg_loss, damp_percent

    def _unpermute_state(self, invperm):
        pass

    def free(self):
        self._stage_release()
        if hasattr(self, "H"):
            del self.H
        for name in ("quantizer", "module_copy", "module", "L", "Xxt", "Xxt_damped"):
            if hasattr(self, name):
                delattr(self, name)


__all__ = ["GPTQ"]

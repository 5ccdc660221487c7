"""GanqHipQuantLinear -- LUT-dequant QuantLinear for GANQ on MI355X.

Supersedes the reference's FakeQuantLinear for this path (gptqmodel/nn_modules/qlinear/fake.py:15-89, which
stores the dequantised fp16 weight and runs F.linear): here the layer stores what GANQ actually produced,
  qweight  int32 [in_features * bits / 32, out_features]   bits-wide indices, GPTQ int32 packing along
                                                           in_features (qlinear/__init__.py:508-538)
  lut      fp16  [out_features, 2^bits]                    per-output-channel codebook
  bias     fp16  [out_features] (optional)
and `forward` == `F.linear(x, lut.gather(1, Q), bias)` == FakeQuantLinear.forward on T.gather(1,Q).half().
With the outlier split (`QuantizeConfig.ganq_outlier_ratio`, paper section 3.3) it also stores the exact outliers,
  outlier_rowptr int32 [out_features + 1], outlier_cols int32 [nnz], outlier_vals fp16 [nnz]   (CSR by output feature)
and `forward` == `F.linear(x, lut.gather(1, Q) + W_sparse, bias)`.

`pack()` keeps the reference signature `(linear, scales, zeros, g_idx)` (utils/model.py:552-570): because that
call only carries the dequantised nn.Linear, the indices/codebook are either passed by keyword
(`ganq_indices=`, `ganq_codebook=`, what ganq_amd's processor does) or recovered from the weight itself -- a
GANQ weight row has at most 2^bits distinct values, so sorted-unique values are the codebook.
"""
import torch
import torch.nn as nn

from ... import _lib
from . import BaseQuantLinear

BACKEND_GANQ_HIP = "ganq_hip"
GEMV_MAX_ROWS = 64


class GanqHipQuantLinear(BaseQuantLinear):
    SUPPORTS_BITS = [2, 3, 4]
    SUPPORTS_GROUP_SIZE = [-1, 16, 32, 64, 128]  # accepted and ignored, as GANQ ignores it (ganq.py:489-495)
    SUPPORTS_DESC_ACT = [True, False]
    SUPPORTS_SYM = [True, False]
    SUPPORTS_SHARDS = True
    SUPPORTS_TRAINING = False
    SUPPORTS_AUTO_PADDING = False
    SUPPORTS_IN_FEATURES_DIVISIBLE_BY = [32]
    SUPPORTS_OUT_FEATURES_DIVISIBLE_BY = [1]
    SUPPORTS_DEVICES = ["cuda"]  # ROCm presents as "cuda" in torch (models/_const.py:34,46)
    SUPPORTS_PLATFORM = ["linux"]
    SUPPORTS_PACK_DTYPES = [torch.int32]
    SUPPORTS_ADAPTERS = []
    SUPPORTS_DTYPES = [torch.float16, torch.bfloat16]

    QUANT_TYPE = "ganq_hip"

    def __init__(self, bits: int, group_size: int, sym: bool, desc_act: bool, in_features: int, out_features: int,
                 bias: bool = False, pack_dtype: torch.dtype = torch.int32, adapter=None, outliers: int = 0, **kwargs):
        super().__init__(bits=bits, group_size=group_size, sym=sym, desc_act=desc_act, in_features=in_features,
                         out_features=out_features, bias=bias, pack_dtype=pack_dtype,
                         backend=kwargs.pop("backend", BACKEND_GANQ_HIP), adapter=adapter, **kwargs)
        self.register_buffer("qweight", torch.zeros((in_features * bits // 32, out_features), dtype=torch.int32))
        self.register_buffer("lut", torch.zeros((out_features, 2 ** bits), dtype=torch.float16))
        if bias:
            self.register_buffer("bias", torch.zeros(out_features, dtype=torch.float16))
        else:
            self.bias = None
        self.outliers = int(outliers)  # number of stored outliers (0: plain GANQ layer, no extra buffers)
        if self.outliers:
            self.register_buffer("outlier_rowptr", torch.zeros(out_features + 1, dtype=torch.int32))
            self.register_buffer("outlier_cols", torch.zeros(self.outliers, dtype=torch.int32))
            self.register_buffer("outlier_vals", torch.zeros(self.outliers, dtype=torch.float16))

    def post_init(self):
        pass

    @staticmethod
    def codebook_from_weight(W: torch.Tensor, bits: int):
        """Recover (Q uint8 [m,n], T [m,V]) from a GANQ-quantised weight: per row, the sorted distinct values.
        Raises if a row has more than 2^bits distinct values (then it is not a GANQ weight)."""
        V = 2 ** bits
        m, n = W.shape
        vals, order = torch.sort(W.float(), dim=1)
        new = torch.ones_like(vals, dtype=torch.bool)
        new[:, 1:] = vals[:, 1:] != vals[:, :-1]
        rank = torch.cumsum(new.to(torch.int64), dim=1) - 1
        if int(rank.max()) >= V:
            raise ValueError(f"weight has more than {V} distinct values in a row: not a {bits}-bit GANQ weight")
        T = torch.zeros((m, V), dtype=torch.float32, device=W.device)
        T.scatter_(1, rank, vals)
        Q = torch.empty((m, n), dtype=torch.uint8, device=W.device)
        Q.scatter_(1, order, rank.to(torch.uint8))
        return Q, T

    def pack(self, linear: nn.Module, scales: torch.Tensor = None, zeros: torch.Tensor = None,
             g_idx: torch.Tensor = None, ganq_indices: torch.Tensor = None, ganq_codebook: torch.Tensor = None,
             ganq_outliers=None):
        W = linear.weight.data
        if type(linear).__name__ == "Conv1D":
            W = W.t()
        dev = self.qweight.device  # the packing kernel runs where the layer lives
        if dev.type != "cuda":
            dev = W.device if W.is_cuda else torch.device("cuda", torch.cuda.current_device())
        if self.outliers and ganq_outliers is None:
            raise ValueError("a layer with outliers cannot be recovered from its weight: pass ganq_outliers=")
        if ganq_indices is None or ganq_codebook is None:
            ganq_indices, ganq_codebook = self.codebook_from_weight(W.to(dev), self.bits)
        Q = ganq_indices.to(device=dev, dtype=torch.uint8).contiguous()
        if Q.shape != (self.out_features, self.in_features):
            raise ValueError(f"indices shape {tuple(Q.shape)} != ({self.out_features}, {self.in_features})")
        self.qweight = _lib.pack_indices(Q, self.bits).to(self.qweight.device)
        self.lut = ganq_codebook.to(dtype=torch.float16).to(self.lut.device).contiguous()
        if self.bias is not None:
            self.bias[:] = linear.bias.to(self.bias.device, dtype=self.bias.dtype)
        else:
            assert linear.bias is None
        if self.outliers:
            rowptr, cols, vals = ganq_outliers
            if cols.numel() != self.outliers or rowptr.numel() != self.out_features + 1:
                raise ValueError("ganq_outliers do not match the layer's outlier count / out_features")
            self.outlier_rowptr = rowptr.to(device=self.qweight.device, dtype=torch.int32).contiguous()
            self.outlier_cols = cols.to(device=self.qweight.device, dtype=torch.int32).contiguous()
            self.outlier_vals = vals.to(device=self.qweight.device, dtype=torch.float16).contiguous()

    def _sparse_dense(self, dtype) -> torch.Tensor:
        """W_sparse as a dense [out_features, in_features] matrix (prefill path / dequantize_weight)"""
        Ws = torch.zeros((self.out_features, self.in_features), dtype=dtype, device=self.outlier_cols.device)
        rows = torch.repeat_interleave(torch.arange(self.out_features, device=Ws.device),
                                       (self.outlier_rowptr[1:] - self.outlier_rowptr[:-1]).long())
        Ws[rows, self.outlier_cols.long()] = self.outlier_vals.to(dtype)
        return Ws

    def dequantize_weight(self) -> torch.Tensor:
        """[out_features, in_features] in the lut dtype == the FakeQuantLinear weight"""
        Wq = _lib.lut_dequant(self.qweight, self.lut, self.in_features, self.bits)
        return Wq + self._sparse_dense(Wq.dtype) if self.outliers else Wq

    def forward(self, x: torch.Tensor):
        if not x.is_cuda:
            raise _lib.GanqHipError("GanqHipQuantLinear runs on the GPU only (no CPU fallback)")
        lut, bias = self.lut, self.bias
        if lut.dtype != x.dtype:
            lut = lut.to(x.dtype)
            bias = None if bias is None else bias.to(x.dtype)
        out_shape = x.shape[:-1] + (self.out_features,)
        x2 = x.reshape(-1, self.in_features)
        # one C-ABI entry for every batch size: the decode kernels up to 64 rows, the fused LUT-dequant GEMM above
        # (csrc/lut_gemm.hip) -- the dequantised weight is never materialised and no library GEMM is called
        if self.outliers:  # x @ W_sparse^T in fp32, added inside the LUT kernel before its one rounding
            vals = self.outlier_vals if self.outlier_vals.dtype == x.dtype else self.outlier_vals.to(x.dtype)
            y = _lib.lut_linear_outliers(x2, self.qweight, lut, bias, self.bits, self.outlier_rowptr, self.outlier_cols, vals)
        else:
            y = _lib.lut_linear(x2, self.qweight, lut, bias, self.bits)
        return y.reshape(out_shape)


__all__ = ["GanqHipQuantLinear", "BACKEND_GANQ_HIP"]

"""Uniform scale/zero finder kept for interface compatibility only: GANQ calls it to fill the (scale, zero)
slots of its return tuple ("Unused, compatibility", ganq.py:489-495,640-644).  Restates the per-channel
weight path of gptqmodel/quantization/quantizer.py:40-168 (perchannel=True, weight=True, mse=0)."""
import torch
import torch.nn as nn

from .config import QuantizeConfig

HF_OPTIMUM = "hf_optimum"


class Quantizer(nn.Module):
    def __init__(self, qcfg: QuantizeConfig, shape=1, name: str = None):
        super().__init__()
        self.qcfg = qcfg
        self.register_buffer("maxq", torch.tensor(0))
        self.register_buffer("scale", torch.zeros(shape))
        self.register_buffer("zero", torch.zeros(shape))
        self.name = name
        self.perchannel = False

    def configure(self, perchannel=False, grid=100, maxshrink=0.8, trits=False, bits: int = 4, sym: bool = False):
        if self.name == HF_OPTIMUM:  # quantizer.py:64-66: a bare module (no NamedModule) takes bits/sym from here
            self.qcfg.bits = bits
            self.qcfg.sym = sym
        self.maxq = torch.tensor(2 ** self.qcfg.bits - 1)
        self.perchannel = perchannel
        self.grid = grid
        self.maxshrink = maxshrink
        if trits:
            self.maxq = torch.tensor(-1)

    def find_params(self, x, weight=False):
        if not weight:
            raise NotImplementedError("only the weight path is on the GANQ hot path")
        dev = x.device
        self.maxq = self.maxq.to(dev)
        shape = x.shape
        x = x.flatten(1) if self.perchannel else x.flatten().unsqueeze(0)
        tmp = torch.zeros(x.shape[0], device=dev)
        xmin = torch.minimum(x.min(1)[0], tmp)
        xmax = torch.maximum(x.max(1)[0], tmp)
        if self.qcfg.sym:
            xmax = torch.maximum(torch.abs(xmin), xmax)
            neg = xmin < 0
            if torch.any(neg):
                xmin[neg] = -xmax[neg]
        flat = (xmin == 0) & (xmax == 0)
        xmin[flat] = -1
        xmax[flat] = +1
        if self.maxq < 0:
            self.scale, self.zero = xmax, xmin
        else:
            self.scale = (xmax - xmin) / self.maxq
            if self.qcfg.sym:
                self.zero = torch.full_like(self.scale, (self.maxq + 1) / 2)
            else:
                self.zero = torch.round(-xmin / self.scale)
        if not self.perchannel:
            self.scale = self.scale.repeat(shape[0])
            self.zero = self.zero.repeat(shape[0])
        new_shape = [-1] + [1] * (len(shape) - 1)
        self.scale = self.scale.reshape(new_shape)
        self.zero = self.zero.reshape(new_shape)

"""ganq_amd -- MI355X (gfx950) implementation of GANQ's per-layer alternating optimisation and
LUT-dequant linear forward, behind the GPTQModel quantizer / QuantLinear plugin interface of
smpanaro/ganq.  The compute path is libganq_hip.so (hand-written HIP, C-ABI in
include/ganq_hip.h); PyTorch is used for device memory, streams and torch.distributed only."""

__version__ = "0.1.0"

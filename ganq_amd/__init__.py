"""ganq_amd -- MI355X (gfx950) implementation of GANQ's per-layer alternating optimisation and
LUT-dequant linear forward, behind the GPTQModel quantizer / QuantLinear plugin interface of
smpanaro/ganq.  The compute path is libganq_hip.so (hand-written HIP, C-ABI in
include/ganq_hip.h); PyTorch is used for device memory, streams and torch.distributed only."""

import os as _os

# The host driver of the MI355X pool only supports dmabuf IPC: RCCL and cross-process tensor sharing need this set BEFORE
# the HIP/HSA runtime initialises (i.e. before the first torch.cuda call of the process), so it is done at package import.
# A launcher that touches the GPU before importing ganq_amd must export it itself (bench.py does).
_os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

__version__ = "0.2.0"

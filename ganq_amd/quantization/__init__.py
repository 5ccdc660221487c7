from .config import FORMAT, QUANT_METHOD, QuantizeConfig
from .ganq import GANQ
from .gptq import GPTQ
from .quantizer import Quantizer

__all__ = ["FORMAT", "QUANT_METHOD", "QuantizeConfig", "GANQ", "GPTQ", "Quantizer"]

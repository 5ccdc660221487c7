"""BaseQuantLinear contract of the reference (gptqmodel/nn_modules/qlinear/__init__.py:33-408) reduced to what
a backend must honour: constructor keywords as `create_quant_layer` passes them (utils/model.py:334-348), the
SUPPORTS_* capability attributes with `validate()` returning NotImplementedError for unsupported settings
(so the caller can fall through to the next backend, utils/model.py:234-239), `pack`, `post_init`, `forward`."""
import sys
from typing import List, Optional, Tuple

import torch
import torch.nn as nn


class BaseQuantLinear(nn.Module):
    SUPPORTS_BITS: List[int] = None
    SUPPORTS_GROUP_SIZE: List[int] = None
    SUPPORTS_DESC_ACT: List[bool] = None
    SUPPORTS_SYM: List[bool] = None
    SUPPORTS_SHARDS: bool = None
    SUPPORTS_TRAINING: bool = None
    SUPPORTS_AUTO_PADDING: bool = None
    SUPPORTS_IN_FEATURES_DIVISIBLE_BY: List[int] = None
    SUPPORTS_OUT_FEATURES_DIVISIBLE_BY: List[int] = None
    SUPPORTS_PACK_DTYPES: List[torch.dtype] = None
    SUPPORTS_ADAPTERS: list = None
    SUPPORTS_DEVICES: List[str] = None
    SUPPORTS_PLATFORM: List[str] = None
    SUPPORTS_DTYPES: List[torch.dtype] = None

    def __init__(self, bits: int, group_size: int, desc_act: bool, sym: bool, in_features: int, out_features: int,
                 bias: bool, pack_dtype: torch.dtype, backend=None, adapter=None, name: str = None, **kwargs):
        super().__init__()
        if name is None:
            name = f"{self.__class__.__module__}.{self.__class__.__qualname__}"
        self.name = name
        self.in_features = in_features
        self.out_features = out_features
        self.group_size = group_size if group_size != -1 else in_features
        self.bits = bits
        self.desc_act = desc_act
        self.sym = sym
        self.pack_dtype = pack_dtype
        self.backend = backend
        self.adapter = adapter
        self.maxq = 2 ** self.bits - 1
        _, err = self._validate(bits=bits, group_size=group_size, desc_act=desc_act, sym=sym, in_features=in_features,
                                out_features=out_features, pack_dtype=pack_dtype, adapter=adapter)
        if err:
            raise err

    @classmethod
    def verify_supports_params(cls):
        missing = [n for n, v in BaseQuantLinear.__dict__.items()
                   if n.startswith("SUPPORTS") and v is None and getattr(cls, n) is None]
        if missing:
            raise ValueError(f"{cls.__name__} these SUPPORTS variables are not overridden: {', '.join(sorted(missing))}")

    @classmethod
    def validate(cls, bits: int, group_size: int, desc_act: bool, sym: bool, in_features: int = None,
                 out_features: int = None, pack_dtype: torch.dtype = None, dynamic: Optional[dict] = None,
                 device=None, trainable: bool = False, adapter=None) -> Tuple[bool, Optional[Exception]]:
        return cls._validate(bits=bits, group_size=group_size, desc_act=desc_act, sym=sym, in_features=in_features,
                             out_features=out_features, pack_dtype=pack_dtype, dynamic=dynamic, device=device,
                             trainable=trainable, adapter=adapter)

    @classmethod
    def _validate(cls, bits=4, group_size=128, desc_act=False, sym=False, pack_dtype=None, dynamic=None,
                  in_features=None, out_features=None, device=None, trainable=None, adapter=None):
        cls.verify_supports_params()
        if adapter is not None and adapter.__class__ not in cls.SUPPORTS_ADAPTERS:
            return False, NotImplementedError(f"{cls} does not support adapter: {adapter}")
        if pack_dtype not in cls.SUPPORTS_PACK_DTYPES:
            return False, NotImplementedError(f"{cls} does not support `pack_dtype`: {pack_dtype}")
        if "all" not in cls.SUPPORTS_PLATFORM and sys.platform not in cls.SUPPORTS_PLATFORM:
            return False, NotImplementedError(f"{cls} does not support platform: {sys.platform}")
        if device is not None and "all" not in cls.SUPPORTS_DEVICES:
            dev_type = torch.device(device).type if not isinstance(device, str) or ":" in device else device
            if dev_type not in cls.SUPPORTS_DEVICES:
                return False, NotImplementedError(f"{cls} does not support device: {device}")
        if trainable and not cls.SUPPORTS_TRAINING:
            return False, NotImplementedError(f"{cls} does not support training.")
        if bits not in cls.SUPPORTS_BITS:
            return False, NotImplementedError(f"{cls} only supports `{cls.SUPPORTS_BITS}` bits: actual bits = `{bits}`")
        if group_size not in cls.SUPPORTS_GROUP_SIZE and group_size != in_features:
            return False, NotImplementedError(f"{cls} only supports `{cls.SUPPORTS_GROUP_SIZE}` group_size: actual "
                                              f"group_size = `{group_size}`")
        if sym not in cls.SUPPORTS_SYM:
            return False, NotImplementedError(f"{cls} only supports `{cls.SUPPORTS_SYM}` sym: actual sym = `{sym}`")
        if desc_act not in cls.SUPPORTS_DESC_ACT:
            return False, NotImplementedError(f"{cls} only supports `{cls.SUPPORTS_DESC_ACT}` desc_act: actual "
                                              f"desc_act = `{desc_act}`")
        if dynamic is not None:
            for pattern, d in dynamic.items():
                if d.get("bits", bits) not in cls.SUPPORTS_BITS:
                    return False, NotImplementedError(f"{cls} only supports `{cls.SUPPORTS_BITS}` bits: dynamic bits = "
                                                      f"`{d.get('bits')}` for layer `{pattern}`")
        if in_features is not None and not all(in_features % d == 0 for d in cls.SUPPORTS_IN_FEATURES_DIVISIBLE_BY):
            return False, NotImplementedError(f"{cls}: `in_features` must be divisible by "
                                              f"{cls.SUPPORTS_IN_FEATURES_DIVISIBLE_BY}.")
        if out_features is not None and not all(out_features % d == 0 for d in cls.SUPPORTS_OUT_FEATURES_DIVISIBLE_BY):
            return False, NotImplementedError(f"{cls}: `out_features` must be divisible by "
                                              f"{cls.SUPPORTS_OUT_FEATURES_DIVISIBLE_BY}.")
        return True, None

    def post_init(self):
        pass

    def pack(self, linear: nn.Module, scales: torch.Tensor, zeros: torch.Tensor, g_idx: torch.Tensor = None):
        raise NotImplementedError


__all__ = ["BaseQuantLinear"]

from .bit_type import BIT_TYPE_DICT, BIT_TYPE_LIST, BitType  # noqa: F401
from .layers import QAct, QConv2d, QIntLayerNorm, QIntSoftmax, QLinear  # noqa: F401
from .observer import build_observer  # noqa: F401
from .quantizer import build_quantizer  # noqa: F401

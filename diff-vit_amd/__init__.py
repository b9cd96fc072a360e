"""p2vit-mi355x: MI355X-native PoT-PTQ quantized ViT forward (drop-in for LeSN-Lab/diff-ViT's
models/ptq + models/vit_fquant quantized inference path).  See DESIGN.md."""
from . import synth  # noqa: F401
from . import engine  # noqa: F401
from .plan import FrozenPlan  # noqa: F401

"""p2vit-mi355x: MI355X-native PoT-PTQ quantized ViT forward -- a drop-in for the quantized inference path of
LeSN-Lab/diff-ViT (models/ptq + models/vit_fquant + config).  ``from diff_vit_amd import *`` yields what the
reference's ``from models import *`` + ``from config import Config`` yield for this path.  See DESIGN.md."""
import os as _os

# One hardware queue per HIP stream: the slices of FrozenPlan.forward_streams run on three streams, and the runtime's default of four
# hardware queues is used up as soon as RCCL (torch.distributed "nccl") is initialised in the process - two slice streams then share a
# queue and every rank's step takes 3.6 instead of 2.64 ms (tools/gather_cost.py).  The HIP runtime reads the variable when it initialises
# (the first HIP call of the process); an explicit setting of the user wins.
_hwq_user = 'GPU_MAX_HW_QUEUES' in _os.environ
_os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')


def _hip_already_up():
    import sys
    t = sys.modules.get('torch')
    try:
        return bool(t is not None and t.cuda.is_initialized())
    except Exception:
        return False


# what is in effect, for bench.py's JSON line and for callers that want to check: `in_effect` is False when this import came too late
HW_QUEUES = {'value': _os.environ['GPU_MAX_HW_QUEUES'], 'set_by': 'environment' if _hwq_user else 'diff_vit_amd default',
             'in_effect': bool(_hwq_user or not _hip_already_up())}
if not HW_QUEUES['in_effect']:
    import warnings as _warnings
    _warnings.warn('diff_vit_amd: the HIP runtime was initialised before this import and GPU_MAX_HW_QUEUES was not set: the default of 8 '
                   'hardware queues is NOT in effect; with RCCL initialised the three-stream forward then loses ~28 % '
                   '(profiles/r03_rccl_queues.txt). Export GPU_MAX_HW_QUEUES=8 before the process makes its first HIP call.', RuntimeWarning)

from . import synth, engine, calib_io, checkpoint, dp, harness, search, ops  # noqa: F401
from .config import Config  # noqa: F401
from .plan import FrozenPlan  # noqa: F401
from .ptq import BIT_TYPE_DICT, QAct, QConv2d, QIntLayerNorm, QIntSoftmax, QLinear  # noqa: F401
from .vit import (VisionTransformer, deit_base_patch16_224, deit_small_patch16_224, deit_tiny_patch16_224,  # noqa: F401
                  vit_base_patch16_224, vit_large_patch16_224)
from .swin import (SwinTransformer, swin_base_patch4_window7_224, swin_small_patch4_window7_224,  # noqa: F401
                   swin_tiny_patch4_window7_224)
from .swin_plan import SwinPlan  # noqa: F401

__all__ = ['BIT_TYPE_DICT', 'QAct', 'QConv2d', 'QIntLayerNorm', 'QIntSoftmax', 'QLinear', 'Config', 'VisionTransformer',
           'deit_tiny_patch16_224', 'deit_small_patch16_224', 'deit_base_patch16_224', 'vit_base_patch16_224',
           'vit_large_patch16_224', 'FrozenPlan', 'SwinTransformer', 'swin_tiny_patch4_window7_224',
           'swin_small_patch4_window7_224', 'swin_base_patch4_window7_224', 'SwinPlan']

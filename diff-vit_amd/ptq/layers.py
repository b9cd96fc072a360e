"""The five quantized operator modules -- the drop-in boundary of the hot path.

Same names, constructor signatures, flags (``quant / calibrate / last_calibrate``), attributes
(``bit_type observer quantizer module_type``) and forward signatures as the reference's models/ptq/layers.py
(QConv2d :12-88, QLinear :104-178, QAct :181-220, QIntLayerNorm :226-291, QIntSoftmax :295-395).

What differs is *where the arithmetic runs*:
  - calibration / float mode: torch float ops on whatever device the tensors live on (no hard-coded .cuda());
  - a whole model in quant state never executes these modules one by one: ``VisionTransformer.forward`` hands the
    batch to the fused HIP engine (see ../vit.py, ../plan.py);
  - a module used stand-alone in quant state fake-quantises through the HIP ``p2v_fake_quant_f32`` kernel on GPU
    tensors (UniformQuantizer.forward) and multiplies with torch's library GEMM.
"""
import torch
import torch.nn as nn
from torch.nn import functional as F

from .bit_type import BIT_TYPE_DICT, BIT_TYPE_LIST
from .observer import build_observer, lp_loss
from .quantizer import build_quantizer


def _build(self, bit_type, calibration_mode, observer_str, quantizer_str, module_type, quant, calibrate, last_calibrate):
    self.quant = quant
    self.calibrate = calibrate
    self.last_calibrate = last_calibrate
    self.bit_type = bit_type
    self.calibration_mode = calibration_mode
    self.observer_str = observer_str
    self.quantizer_str = quantizer_str
    self.module_type = module_type
    self.observer = build_observer(observer_str, module_type, bit_type, calibration_mode)
    self.quantizer = build_quantizer(quantizer_str, bit_type, self.observer, module_type)


def _calibrate_weight(mod, weight, x, others, attn=False, attn_para=None):
    """the four-bit-type calibration loop shared by QLinear / QConv2d (layers.py:57-71,148-170); returns the
    per-bit-type weight MSEs (``distance``)."""
    distance = []
    for bit_type in BIT_TYPE_LIST:
        if bit_type == BIT_TYPE_DICT['uint8']:
            continue
        mod.quantizer.bit_type = bit_type
        mod.observer.bit_type = bit_type
        mod.observer.calibration_mode = 'layer_wise' if bit_type == BIT_TYPE_DICT['int8'] else 'channel_wise'
        mod.quantizer.observer.update(weight)
        if mod.module_type == 'linear_weight' or mod.last_calibrate:
            mod.quantizer.update_quantization_params(x, others=others, attn=attn, attn_para=attn_para)
        if mod.module_type == 'linear_weight':
            wq = mod.quantizer.dequantize(mod.quantizer.quant(weight))
            distance.append(lp_loss(weight, wq, p=2.0, reduction='all'))
    return distance


def _select_bits(mod, bit_config):
    if bit_config:
        bt = BIT_TYPE_DICT['int' + str(bit_config)]          # KeyError for unsupported widths, like the reference
        mod.quantizer.bit_type = bt
        mod.observer.bit_type = bt


class QConv2d(nn.Conv2d):

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 quant=False, calibrate=False, last_calibrate=False, bit_type=BIT_TYPE_DICT['int8'],
                 calibration_mode='layer_wise', observer_str='minmax', quantizer_str='uniform'):
        super().__init__(in_channels=in_channels, out_channels=out_channels, kernel_size=kernel_size, stride=stride,
                         padding=padding, dilation=dilation, groups=groups, bias=bias)
        _build(self, bit_type, calibration_mode, observer_str, quantizer_str, 'conv_weight', quant, calibrate, last_calibrate)

    def forward(self, x, bit_config):
        if self.calibrate:
            _calibrate_weight(self, self.weight, x, [self.bias, self.stride, self.padding, self.dilation, self.groups])
        if not self.quant or bit_config == -1:
            return F.conv2d(x, self.weight, self.bias, self.stride, self.padding, self.dilation, self.groups)
        _select_bits(self, bit_config)
        weight = self.quantizer(self.weight)
        return F.conv2d(x, weight, self.bias, self.stride, self.padding, self.dilation, self.groups)


class QLinear(nn.Linear):

    def __init__(self, in_features, out_features, bias=True, quant=False, calibrate=False, last_calibrate=False,
                 bit_type=BIT_TYPE_DICT['int8'], calibration_mode='layer_wise', observer_str='minmax',
                 quantizer_str='uniform'):
        super().__init__(in_features, out_features, bias)
        _build(self, bit_type, calibration_mode, observer_str, quantizer_str, 'linear_weight', quant, calibrate, last_calibrate)

    def forward(self, x, global_distance=[], bit_config=None, weight_smoothed=None, attn=False, attn_para=None):
        if weight_smoothed is None:
            weight_smoothed = self.weight
        if not self.quant or bit_config == -1:
            y = F.linear(x, weight_smoothed, self.bias)
        if self.calibrate:
            global_distance.append(_calibrate_weight(self, weight_smoothed, x, [self.bias], attn, attn_para))
        if not self.quant or bit_config == -1:
            return y
        _select_bits(self, bit_config)
        weight = self.quantizer(weight_smoothed)
        return F.linear(x, weight, self.bias)


class QAct(nn.Module):

    def __init__(self, quant=False, calibrate=False, last_calibrate=False, bit_type=BIT_TYPE_DICT['int8'],
                 calibration_mode='layer_wise', observer_str='minmax', quantizer_str='uniform'):
        super().__init__()
        _build(self, bit_type, calibration_mode, observer_str, quantizer_str, 'activation', quant, calibrate, last_calibrate)

    def forward(self, x, asymmetric=False, attn=False, attn_para=None):
        if self.calibrate:
            if asymmetric:
                self.quantizer.bit_type = BIT_TYPE_DICT['uint8']
                self.observer.bit_type = BIT_TYPE_DICT['uint8']
                self.observer.symmetric = False
            self.quantizer.observer.update(x)
            if self.last_calibrate:
                self.quantizer.update_quantization_params(x, attn=attn, attn_para=attn_para)
        if not self.quant:
            return x
        return self.quantizer(x)


class QIntLayerNorm(nn.LayerNorm):

    def __init__(self, normalized_shape, eps=1e-5, elementwise_affine=True):
        super().__init__(normalized_shape, eps, elementwise_affine)
        assert isinstance(normalized_shape, int)
        self.mode = 'ln'

    def get_MN(self, x):
        """dyadic multiplier of a positive fp32 value: N = clamp(7 - floor(log2 x), 0, 31), M = floor(x 2^N) <= 255
        (layers.py:234-238); the exponent is taken exactly (frexp) instead of through log2f."""
        bit = 7
        _, e = torch.frexp(x)
        fl = torch.where(x == 0, torch.full_like(x, float('-inf')), (e - 1).to(x.dtype))     # floor(log2(0)) = -inf -> N = 31
        N = torch.clamp(bit - fl, 0, 31)
        M = torch.clamp(torch.floor(torch.ldexp(x, N.to(torch.int32))), 0, 2**(bit + 1) - 1)
        return M, N

    def forward(self, x, in_quantizer=None, out_quantizer=None, out_quantizer_scale=None, in_scale_expand=1):
        if self.mode == 'ln':
            return F.layer_norm(x, self.normalized_shape, self.weight, self.bias, self.eps)
        if self.mode != 'int':
            raise NotImplementedError
        in_scale = in_quantizer.scale
        if in_scale_expand != 1:
            in_scale = in_scale.unsqueeze(-1).expand(-1, in_scale_expand).T.reshape(-1)
        out_scale_global = out_quantizer.scale
        assert in_scale is not None and out_scale_global is not None
        channel_nums = x.shape[-1]
        in_scale = in_scale.reshape(1, 1, -1).to(x.device)
        out_scale = out_scale_global * out_quantizer_scale if out_quantizer_scale is not None else out_scale_global
        out_scale = out_scale.reshape(1, 1, -1).to(x.device)
        x_q = (x / in_scale).round()
        in_scale1 = in_scale.min()
        x_q = x_q * (in_scale / in_scale1).round()
        xd = x_q.double()                                   # integer sums are exact; one rounding each, then fp32
        s1 = xd.sum(dim=-1).float()
        s2 = (xd * xd).sum(dim=-1).float()
        mean_x_q = (s1 / channel_nums) * in_scale1
        std_x_q = (in_scale1 / channel_nums) * torch.sqrt((channel_nums * s2 - s1 * s1).double()).float()
        A = (in_scale1 / std_x_q).unsqueeze(-1) * self.weight.reshape(1, 1, -1) / out_scale
        A_sign = A.sign()
        M, N = self.get_MN(A.abs())
        pN = torch.ldexp(torch.ones_like(N), N.to(torch.int32))
        B = ((self.bias.reshape(1, 1, -1) - (mean_x_q / std_x_q).unsqueeze(-1) * self.weight.reshape(1, 1, -1)) / out_scale * pN).round()
        x_q = ((A_sign * M * x_q + B) / pN).round()
        return x_q * out_scale


class QIntSoftmax(nn.Module):

    def __init__(self, log_i_softmax=False, quant=False, calibrate=False, last_calibrate=False,
                 bit_type=BIT_TYPE_DICT['int8'], calibration_mode='layer_wise', observer_str='minmax',
                 quantizer_str='uniform'):
        super().__init__()
        self.log_i_softmax = log_i_softmax
        _build(self, bit_type, calibration_mode, observer_str, quantizer_str, 'activation', quant, calibrate, last_calibrate)

    @staticmethod
    def log_round(x):
        """nearest power-of-two exponent of x >= 1 in the linear domain (layers.py:323-329), exact exponent."""
        m, e = torch.frexp(x)
        return (e - 1).to(x.dtype) + (m >= 0.75).to(x.dtype)

    @staticmethod
    def int_softmax(x, scaling_factor):
        """I-BERT integer exp (layers.py:331-365): returns (exp_int, exp_int_sum)."""
        x_int = x / scaling_factor
        x_int = x_int - x_int.max(dim=-1, keepdim=True)[0]
        x0_int = torch.floor(-0.6931 / scaling_factor)
        x_int = torch.max(x_int, 32 * x0_int)
        q = torch.floor(x_int / x0_int)
        r = x_int - x0_int * q
        b_int = torch.floor((0.96963238 / 0.35815147) / scaling_factor)
        c_int = torch.floor((1. / 0.35815147) / scaling_factor**2)
        z = r * (r + b_int) + c_int
        exp_int = torch.clamp(torch.floor(z * 2**(32 - q)), min=0)
        on_grid = bool((x_int == x_int.round()).all())
        if on_grid:      # quantized scores: the sum of integers up to 2^50 is taken exactly, then rounded once
            exp_sum = exp_int.double().sum(dim=-1, keepdim=True)
            exp_sum = exp_sum.to(torch.int64).to(torch.float32) if float(exp_sum.max()) < 2.0**53 else exp_int.sum(dim=-1, keepdim=True)
        else:            # calibration pass on float scores: plain fp32, as the reference runs it
            exp_sum = exp_int.sum(dim=-1, keepdim=True)
        return exp_int, exp_sum

    def forward(self, x, scale):
        if self.log_i_softmax and scale is not None:
            scale = scale.to(x.device)
            exp_int, exp_int_sum = self.int_softmax(x, scale)
            softmax_out = torch.round(exp_int_sum / exp_int)
            rounds = self.log_round(softmax_out)
            mask = rounds >= 2**self.bit_type.bits
            qlog = torch.clamp(rounds, 0, 2**self.bit_type.bits - 1)
            deq_softmax = 2**(-qlog)
            deq_softmax[mask] = 0
            return deq_softmax
        return x.softmax(dim=-1)

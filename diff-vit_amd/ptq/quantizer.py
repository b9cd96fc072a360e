"""Quantizers (mirror of the reference's models/ptq/quantizer/{base,uniform,log2,build}.py).

``UniformQuantizer.forward`` = dequantize(quant(x)) (base.py:42-45).  On a GPU tensor the round/clamp/rescale
chain is ONE HIP kernel through the C ABI (``p2v_fake_quant_f32``); on CPU tensors (calibration on the host)
the same arithmetic runs as torch ops.  Both are bit-identical (tests/test_engine_gpu.py::test_fake_quant)."""
import torch
import torch.nn as nn


class BaseQuantizer(nn.Module):

    def __init__(self, bit_type, observer, module_type):
        super().__init__()
        self.bit_type = bit_type
        self.observer = observer
        self.module_type = module_type

    def get_reshape_range(self, inputs):
        if self.module_type == 'conv_weight':
            return (-1, 1, 1, 1)
        if self.module_type == 'linear_weight':
            return (-1, 1)
        if self.module_type == 'activation':
            if len(inputs.shape) == 2:
                return (1, -1)
            if len(inputs.shape) == 3:
                return (1, 1, -1)
            if len(inputs.shape) == 4:
                return (1, -1, 1, 1)
            raise NotImplementedError
        raise NotImplementedError

    def update_quantization_params(self, *args, **kwargs):
        pass

    def quant(self, inputs, scale=None, zero_point=None):
        raise NotImplementedError

    def dequantize(self, inputs, scale=None, zero_point=None):
        raise NotImplementedError

    def forward(self, inputs):
        return self.dequantize(self.quant(inputs))


class UniformQuantizer(BaseQuantizer):

    def __init__(self, bit_type, observer, module_type):
        super().__init__(bit_type, observer, module_type)
        self.scale = None
        self.zero_point = None
        self.dic_scale = {}
        self.dic_zero_point = {}

    def update_quantization_params(self, *args, **kwargs):
        scale, zero_point = self.observer.get_quantization_params(*args, **kwargs)
        if self.module_type == 'activation':
            self.scale, self.zero_point = scale, zero_point
        else:
            self.dic_scale[self.bit_type.name] = scale
            self.dic_zero_point[self.bit_type.name] = zero_point

    def _params(self, scale, zero_point):
        if scale is None:
            scale = self.scale if self.module_type == 'activation' else self.dic_scale[self.bit_type.name]
        if zero_point is None:
            zero_point = self.zero_point if self.module_type == 'activation' else self.dic_zero_point[self.bit_type.name]
        return scale, zero_point

    def quant(self, inputs, scale=None, zero_point=None):
        scale, zero_point = self._params(scale, zero_point)
        shape = self.get_reshape_range(inputs)
        scale = scale.reshape(shape).to(inputs.device)
        zero_point = zero_point.reshape(shape).to(inputs.device)
        outputs = inputs / scale + zero_point
        return outputs.round().clamp(self.bit_type.lower_bound, self.bit_type.upper_bound)

    def dequantize(self, inputs, scale=None, zero_point=None):
        scale, zero_point = self._params(scale, zero_point)
        shape = self.get_reshape_range(inputs)
        return (inputs - zero_point.reshape(shape).to(inputs.device)) * scale.reshape(shape).to(inputs.device)

    def forward(self, inputs):
        scale, zero_point = self._params(None, None)
        if inputs.is_cuda and inputs.dtype == torch.float32 and not bool((zero_point != 0).any()):
            from .. import engine as E       # HIP path: one fused kernel
            x = inputs.contiguous()
            s = scale.detach().reshape(-1).float().to(x.device).contiguous()
            shape = self.get_reshape_range(x)
            cdim = [i for i, d in enumerate(shape) if d == -1][0]
            inner = 1
            for d in x.shape[cdim + 1:]:
                inner *= int(d)
            if s.numel() not in (1, x.shape[cdim]):
                raise RuntimeError('scale has %d entries for %d channels' % (s.numel(), x.shape[cdim]))
            out = torch.empty_like(x)
            E.check(E.lib().p2v_fake_quant_f32(E.ptr(x), x.numel(), E.ptr(s), s.numel(), inner, self.bit_type.lower_bound,
                                               self.bit_type.upper_bound, E.ptr(out), None, E.stream_ptr()))
            return out
        return self.dequantize(self.quant(inputs))


class Log2Quantizer(BaseQuantizer):
    """constructed for the softmax but never called in the live path (models/ptq/layers.py:394 is commented)."""

    def __init__(self, bit_type, observer, module_type):
        super().__init__(bit_type, observer, module_type)
        self.softmax_mask = None

    def quant(self, inputs):
        rounds = torch.round(-1 * inputs.log2())
        self.softmax_mask = rounds >= 2**self.bit_type.bits
        return torch.clamp(rounds, 0, 2**self.bit_type.bits - 1)

    def dequantize(self, inputs):
        outputs = 2**(-1 * inputs)
        outputs[self.softmax_mask] = 0
        return outputs


str2quantizer = {'uniform': UniformQuantizer, 'log2': Log2Quantizer}


def build_quantizer(quantizer_str, bit_type, observer, module_type):
    """quantizer/build.py:8"""
    return str2quantizer[quantizer_str](bit_type, observer, module_type)

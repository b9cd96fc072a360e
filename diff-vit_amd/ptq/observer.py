"""Calibration observers (float, one-off): statistics -> (scale, zero_point).

Mirror of the reference's models/ptq/observer/{base,minmax,ptf,ema,omse,percentile,build}.py.  Same classes,
constructor arguments, attributes (``max_val min_val symmetric calibration_mode bit_type eps``) and results;
no ``.cuda()`` calls (the reference hard-codes them, minmax.py:67-73,182-209), and the per-channel Python loops
of the PoT search are batched: one GEMM per candidate exponent instead of one GEMV per channel and candidate
(minmax.py:198-240; 26-54 s per calibration batch in the reference).
"""
import torch
from torch.nn import functional as F

_LOG2 = None


def _log2():
    global _LOG2
    if _LOG2 is None:
        _LOG2 = torch.log(torch.tensor([2.0]))
    return _LOG2


def lp_loss(pred, tgt, p=2.0, reduction='none'):
    """observer/utils.py:2-9"""
    if reduction == 'none':
        return (pred - tgt).abs().pow(p).sum(1).mean()
    return (pred - tgt).abs().pow(p).mean()


def round_ln(x, kind=None):
    """floor / ceil / nearest power-of-two exponent of x (minmax.py:65-73): fp32 log(x)/log(2)."""
    l2 = _log2().to(x.device)
    if kind == 'ceil':
        return torch.ceil(torch.div(torch.log(x), l2))
    y = torch.floor(torch.div(torch.log(x), l2))
    if kind == 'floor':
        return y
    return torch.gt((x - 2**y), (2**(y + 1) - x)) + y


class BaseObserver:

    def __init__(self, module_type, bit_type, calibration_mode):
        self.module_type = module_type
        self.bit_type = bit_type
        self.calibration_mode = calibration_mode
        self.max_val = None
        self.min_val = None
        self.eps = torch.finfo(torch.float32).eps

    def reshape_tensor(self, v):
        if not isinstance(v, torch.Tensor):
            v = torch.tensor(v)
        v = v.detach()
        if self.module_type in ['conv_weight', 'linear_weight']:
            v = v.reshape(v.shape[0], -1)
        elif self.module_type == 'activation':
            if len(v.shape) == 4:
                v = v.permute(0, 2, 3, 1)
            v = v.reshape(-1, v.shape[-1])
            v = v.transpose(0, 1)
        else:
            raise NotImplementedError
        return v

    def update(self, v):
        raise NotImplementedError

    def get_quantization_params(self, *args, **kwargs):
        raise NotImplementedError

    def _minmax_update(self, v):
        self.v = v
        v = self.reshape_tensor(v)
        cur_max = v.max(axis=1).values
        self.max_val = cur_max if self.max_val is None else torch.max(cur_max, self.max_val)
        cur_min = v.min(axis=1).values
        self.min_val = cur_min if self.min_val is None else torch.min(cur_min, self.min_val)
        if self.calibration_mode == 'layer_wise':
            self.max_val = self.max_val.max()
            self.min_val = self.min_val.min()


class MinmaxObserver(BaseObserver):
    """min/max range -> power-of-two scale 2^alpha, alpha searched in {floor-1 .. floor+2} by the MSE of the
    layer OUTPUT (weights) or of the tensor itself (activations).  minmax.py:9-272."""

    def __init__(self, module_type, bit_type, calibration_mode):
        super().__init__(module_type, bit_type, calibration_mode)
        self.symmetric = self.bit_type.signed

    def update(self, v):
        self._minmax_update(v)

    def _layer_out(self, w, bias):
        o = self.others
        if self.module_type == 'conv_weight':
            return F.conv2d(self.input, w, bias, o[1], o[2], o[3], o[4])
        return F.linear(self.input, w, bias)

    def _search(self, scale, zero_point):
        qmin, qmax = self.bit_type.lower_bound, self.bit_type.upper_bound
        alpha_floor = round_ln(scale, 'floor')
        zp = 0 if zero_point is None else zero_point
        if self.module_type == 'activation':
            # layer-wise: score = MSE(x, fake_quant(x)) (get_out returns the tensors themselves, minmax.py:139-152)
            x = self.input
            score = []
            for k in range(4):
                a = alpha_floor[0] - 1 + k
                xq = ((x / 2**a + zp).round().clamp(qmin, qmax) - zp) * 2**a
                score.append(lp_loss(x, xq, p=2.0, reduction='all'))
            alpha = alpha_floor.clone()
            alpha[0] = alpha_floor[0] - 1 + score.index(min(score))
            return alpha
        w = self.v
        bias = self.others[0] if self.others else None
        if self.calibration_mode == 'layer_wise':
            ref = self._layer_out(w, bias)
            score = []
            for k in range(4):
                a = alpha_floor[0] - 1 + k
                wq = ((w / 2**a + zp).round().clamp(qmin, qmax) - zp) * 2**a
                score.append(lp_loss(ref, self._layer_out(wq, bias), p=2.0, reduction='all'))
            alpha = alpha_floor.clone()
            alpha[0] = alpha_floor[0] - 1 + score.index(min(score))
            return alpha
        # channel-wise: all output channels at once; per-channel MSE = mean over every non-channel axis
        ref = self._layer_out(w, bias)
        cdim = 1 if self.module_type == 'conv_weight' else ref.dim() - 1
        red = [d for d in range(ref.dim()) if d != cdim]
        shape = [-1] + [1] * (w.dim() - 1)
        scores = []
        acc = torch.float64 if ref.is_cuda else ref.dtype      # GPU runs: fp64 scores (the fp32 reduction order differs from the CPU's)
        for k in range(4):
            a = (alpha_floor - 1 + k).reshape(shape)
            wq = ((w / 2**a + zp).round().clamp(qmin, qmax) - zp) * 2**a
            scores.append((ref - self._layer_out(wq, bias)).to(acc).abs().pow(2.0).mean(dim=red))
        best = torch.stack(scores, 0).argmin(dim=0)            # first minimum on ties, like list.index(min(...))
        return alpha_floor - 1 + best.to(alpha_floor.dtype)

    def get_quantization_params(self, x, others=None, attn=False, attn_para=None, *args, **kwargs):
        max_val, min_val = self.max_val, self.min_val
        self.input, self.others, self.attn, self.attn_para = x, others, attn, attn_para
        qmax, qmin = self.bit_type.upper_bound, self.bit_type.lower_bound
        if self.symmetric:
            zero_point = torch.zeros_like(max_val, dtype=torch.int64)
            max_val = torch.max(-min_val, max_val)
            scale = max_val / (float(qmax - qmin) / 2)
            scale = 2**self._search(scale, None)
            scale.clamp_(self.eps)
        else:
            scale = (max_val - min_val) / float(qmax - qmin)
            zero_point = qmin - torch.round(min_val / scale)
            zero_point.clamp_(qmin, qmax)
            scale = 2**self._search(scale, zero_point)
            scale.clamp_(self.eps)
        return scale, zero_point


class PtfObserver(BaseObserver):
    """Power-of-Two Factor: one float base scale, per-channel factor in {1,2,4,8} by per-channel MSE.  ptf.py:8-134."""

    def update(self, v):
        self._minmax_update(v)

    def get_quantization_params(self, inputs, *args, **kwargs):
        max_val, min_val = self.max_val, self.min_val
        qmax, qmin = self.bit_type.upper_bound, self.bit_type.lower_bound
        max_val_t = torch.max(-min_val.min(), max_val.max())
        scale8 = 2 * max_val_t / float(qmax - qmin)
        scale8.clamp_(self.eps)
        scale4 = scale8 / 2
        scale2 = scale4 / 2
        scale1 = scale2 / 2
        zero_point = torch.zeros_like(max_val.max(), dtype=torch.int64)
        red = list(range(inputs.dim() - 1))
        scores = [(inputs - ((inputs / s).round().clamp(qmin, qmax)) * s).abs().pow(2.0).mean(dim=red)
                  for s in (scale1, scale2, scale4, scale8)]
        self.scale_mask = 2.0**torch.stack(scores, 0).argmin(dim=0).to(torch.float32)
        return scale1 * self.scale_mask, zero_point


class _FloatScaleObserver(BaseObserver):
    """FQ-ViT's observers with float (non power-of-two) scales (observer/{ema,omse,percentile}.py): the shared min/max ->
    (scale, zero_point) step.  They calibrate the module surface like the reference; the integer engine refuses non-PoT
    activation scales at freeze time (the kernels multiply by exact 1/s)."""

    def __init__(self, module_type, bit_type, calibration_mode):
        super().__init__(module_type, bit_type, calibration_mode)
        self.symmetric = self.bit_type.signed

    def update(self, v):
        self._minmax_update(v)

    def get_quantization_params(self, *args, **kwargs):
        qmax, qmin = self.bit_type.upper_bound, self.bit_type.lower_bound
        max_val, min_val = self.max_val, self.min_val
        if self.symmetric:
            m = torch.max(-min_val, max_val)
            scale = (m / (float(qmax - qmin) / 2)).clamp(self.eps)
            return scale, torch.zeros_like(m, dtype=torch.int64)
        scale = ((max_val - min_val) / float(qmax - qmin)).clamp(self.eps)
        return scale, (qmin - torch.round(min_val / scale)).clamp(qmin, qmax)


class EmaObserver(_FloatScaleObserver):

    def __init__(self, module_type, bit_type, calibration_mode, ema_sigma=0.01):
        super().__init__(module_type, bit_type, calibration_mode)
        self.ema_sigma = ema_sigma

    def update(self, v):
        v = self.reshape_tensor(v)
        cur_max, cur_min = v.max(axis=1).values, v.min(axis=1).values
        self.max_val = cur_max if self.max_val is None else self.max_val + self.ema_sigma * (cur_max - self.max_val)
        self.min_val = cur_min if self.min_val is None else self.min_val + self.ema_sigma * (cur_min - self.min_val)
        if self.calibration_mode == 'layer_wise':
            self.max_val, self.min_val = self.max_val.max(), self.min_val.min()


class OmseObserver(_FloatScaleObserver):
    """min/max range shrunk by the candidate (of 90, 1 % steps) that minimises the L2 quantisation error of the tensor
    (omse.py:31-56, after LAPQ).  The result is always the asymmetric (scale, zero_point) pair, as in the reference.
    Delta: the reference's signature takes ``inputs`` only, so its own QAct/QLinear calls (which pass others/attn keywords,
    layers.py:68,160,216) raise TypeError; the keywords are accepted and ignored here."""

    def get_quantization_params(self, inputs, *args, **kwargs):
        qmax, qmin = self.bit_type.upper_bound, self.bit_type.lower_bound
        hi0, lo0 = self.max_val, self.min_val
        best, scale, zero_point = 1e+10, None, None
        for step in range(90):
            shrink = 1.0 - step * 0.01
            hi, lo = hi0 * shrink, lo0 * shrink
            s = ((hi - lo) / float(qmax - qmin)).clamp(self.eps)
            zp = (qmin - torch.round(lo / s)).clamp(qmin, qmax)
            err = lp_loss(inputs, ((inputs / s + zp).round().clamp(qmin, qmax) - zp) * s, p=2.0, reduction='all')
            if err < best:
                best, scale, zero_point = err, s, zp
                self.max_val, self.min_val = hi, lo
        return scale, zero_point


class PercentileObserver(_FloatScaleObserver):
    """0.99999 / 0.00001 quantiles of the whole tensor, exponentially averaged over calibration batches (sigma 0.01);
    layer-wise only (percentile.py:23-52)."""

    def __init__(self, module_type, bit_type, calibration_mode, percentile_sigma=0.01, percentile_alpha=0.99999):
        super().__init__(module_type, bit_type, calibration_mode)
        self.percentile_sigma = 0.01          # the reference ignores both constructor arguments (percentile.py:19-20)
        self.percentile_alpha = 0.99999

    def update(self, v):
        assert self.calibration_mode == 'layer_wise'
        flat = self.reshape_tensor(v).reshape(-1)
        try:
            cur_max = torch.quantile(flat, self.percentile_alpha)
            cur_min = torch.quantile(flat, 1.0 - self.percentile_alpha)
        except RuntimeError:                  # torch.quantile refuses inputs above 16 M elements
            import numpy as np
            host = flat.cpu().numpy()
            cur_max = torch.tensor(np.percentile(host, self.percentile_alpha * 100), device=v.device, dtype=torch.float32)
            cur_min = torch.tensor(np.percentile(host, (1 - self.percentile_alpha) * 100), device=v.device, dtype=torch.float32)
        self.max_val = cur_max if self.max_val is None else self.max_val + self.percentile_sigma * (cur_max - self.max_val)
        self.min_val = cur_min if self.min_val is None else self.min_val + self.percentile_sigma * (cur_min - self.min_val)


str2observer = {'minmax': MinmaxObserver, 'ema': EmaObserver, 'omse': OmseObserver, 'percentile': PercentileObserver,
                'ptf': PtfObserver}


def build_observer(observer_str, module_type, bit_type, calibration_mode):
    """observer/build.py:17"""
    return str2observer[observer_str](module_type, bit_type, calibration_mode)

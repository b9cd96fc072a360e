"""CPU oracle for the PoT-PTQ quantized ViT forward path  --  TEST INFRASTRUCTURE, NOT PRODUCT.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
file.  The product package (``diff-vit_amd/``) never does; its quantized forward fails loudly when the
HIP library is missing.

What it is: a functional torch-CPU restatement of the reference's algorithm (LeSN-Lab/diff-ViT,
P2-ViT fork of FQ-ViT) for DeiT/ViT:

  * calibration (float pass + observers):   ``calibrate()``      -> models/ptq/observer/minmax.py:16-272,
    ptf.py:14-134, models/vit_fquant.py:196-280 (SmoothQuant-PoT), models/layers_quant.py:222-303
  * quantized inference forward:            ``quant_forward()``  -> models/vit_fquant.py:281-344,414-484,
    700-799, models/layers_quant.py:304-351,416-492, models/ptq/layers.py:55-88,133-178,207-220,
    255-289,323-376, models/ptq/quantizer/uniform.py:50-127

Canonical ("integer-exact") semantics.  The reference simulates integers in fp32, so a handful of its
results depend on the fp32 summation order / libm of the machine it runs on (measured here: MKL
K-blocking changes ``F.linear``+bias rounding for K=1536; torch-CPU GELU uses an A&S polynomial erf;
``sum`` of LIS exponentials up to 2^50 is order dependent).  The oracle pins those points to the
machine-independent reading, which the HIP kernels reproduce bit for bit:

  - every contraction / reduction over integer-valued data is EXACT (int64 / exact fp32), followed by
    the single fp32 rounding the reference applies next;
  - every elementwise fp32 operation is performed in the reference's order with IEEE fp32 ops;
  - ``floor(log2(x))`` is the exact binary exponent (frexp);  ``2**n`` is exact (ldexp);  ``sqrt`` is the
    correctly rounded IEEE value (torch.sqrt on CPU is MKL-VML and differs between Xeon and EPYC hosts);
  - GELU is the correctly-rounded fp32 value of 0.5*y*erfc(-y/sqrt(2)) (evaluated in fp64).

Pinning: ``oracle/gen_golden.py`` imports the real reference in the build container and stores its
inputs/outputs in ``tests/golden``; ``tests/test_oracle_golden.py`` checks this file against them and
reports the (tiny) mismatch the canonical reading introduces.

The decomposition is executable: ``OracleViT(..., imitate={'linear', 'gelu', 'lis', 'ln'})`` swaps each canonical piece back
for the torch-CPU operation the reference calls at that place (``F.linear``/``F.conv2d`` with the bias inside, ``F.gelu``, the
all-fp32 ``int_softmax`` with its fp32 ``sum``, the fp32 LayerNorm statistics with ``torch.sqrt``/``log2``/``pow``).  With all
four the oracle is the reference as run in this container, bit for bit, also at DeiT-S size
(``tests/test_oracle_golden.py::test_deit_small_decomposition``); with one of them left canonical the first differing tap is the
one DESIGN.md section 2 names.
"""
import math

import torch
import torch.nn.functional as F

# models/ptq/bit_type.py:42-57  (lower, upper) per bit type
BITS = {'uint3': (0, 7), 'uint4': (0, 15), 'int4': (-8, 7), 'int8': (-128, 127), 'uint8': (0, 255)}
# calibration loops over BIT_TYPE_LIST minus uint8 (models/ptq/layers.py:151-153)
CALIB_BIT_ORDER = ('uint3', 'uint4', 'int4', 'int8')
ALPHA_ATTN = (0.35,)   # models/vit_fquant.py:32
ALPHA_MLP = (0.5,)     # models/layers_quant.py:14
BIT_POOL = (4, 8)      # models/vit_fquant.py:33
EPS = torch.finfo(torch.float32).eps   # observer/base.py:14

_LOG2 = torch.log(torch.tensor([2.0]))


# --------------------------------------------------------------------------------------------------
# shared small pieces
# --------------------------------------------------------------------------------------------------
def fake_quant(x, scale, lo, hi, shape=None):
    """UniformQuantizer.forward = dequantize(quant(x)), zero_point == 0 (uniform.py:50-127)."""
    s = scale.reshape(shape) if shape is not None else scale
    return torch.clamp(torch.round(x / s), lo, hi) * s


def act_shape(x):
    """BaseQuantizer.get_reshape_range for activations (quantizer/base.py:14-31)."""
    return {2: (1, -1), 3: (1, 1, -1), 4: (1, -1, 1, 1)}[x.dim()]


def pot_floor(x):
    """round_ln(x, 'floor') (minmax.py:65-73): floor(log(x)/log(2)) with torch fp32 ops."""
    return torch.floor(torch.div(torch.log(x), _LOG2))


def pot_round(x):
    """round_ln(x, 'round'): nearest power of two in the linear domain, ties go down."""
    y = pot_floor(x)
    return torch.gt((x - 2**y), (2**(y + 1) - x)) + y


def lp2(a, b):
    """lp_loss(..., p=2, reduction='all') (observer/utils.py:2-9)."""
    return (a - b).abs().pow(2.0).mean()


def _obs_rows(v, module_type):
    """BaseObserver.reshape_tensor (observer/base.py:17-29): one row per channel."""
    v = v.detach()
    if module_type in ('conv_weight', 'linear_weight'):
        return v.reshape(v.shape[0], -1)
    if v.dim() == 4:
        v = v.permute(0, 2, 3, 1)
    return v.reshape(-1, v.shape[-1]).transpose(0, 1)


class _MinMax:
    """MinmaxObserver state: running per-channel max/min, collapsed when layer-wise (minmax.py:16-39)."""

    def __init__(self, module_type):
        self.module_type = module_type
        self.max_val = None
        self.min_val = None

    def update(self, v, mode):
        self.v = v
        r = _obs_rows(v, self.module_type)
        cur_max = r.max(axis=1).values
        cur_min = r.min(axis=1).values
        self.max_val = cur_max if self.max_val is None else torch.max(cur_max, self.max_val)
        self.min_val = cur_min if self.min_val is None else torch.min(cur_min, self.min_val)
        if mode == 'layer_wise':
            self.max_val = self.max_val.max()
            self.min_val = self.min_val.min()

    def params(self, x, bit, mode, others=None):
        """get_quantization_params, symmetric branch (minmax.py:42-272): PoT scale 2^alpha with alpha in
        {floor-1 .. floor+2} minimising the MSE of the layer OUTPUT (weights) or of the tensor itself
        (activations)."""
        qmin, qmax = BITS[bit]
        max_val = torch.max(-self.min_val, self.max_val)
        scale = max_val / (float(qmax - qmin) / 2)
        alpha_floor = pot_floor(scale)
        alpha = pot_round(scale)
        mt = self.module_type
        dim = 1 if mode == 'layer_wise' else scale.shape[0]
        for j in range(dim):
            if dim == 1:
                w = x if mt == 'activation' else self.v
                bias = others[0] if others else None
            else:
                w = self.v[j, ...].unsqueeze(0)
                bias = others[0][j].unsqueeze(0) if others else None

            def out_of(wq):
                if mt == 'activation':
                    return wq
                if mt == 'conv_weight':
                    return F.conv2d(x, wq, bias, others[1], others[2], others[3], others[4])
                return F.linear(x, wq, bias)

            ref = x if mt == 'activation' else out_of(w)
            score = []
            for k in range(4):
                a = alpha_floor[j] - 1 + k
                wq = torch.clamp(torch.round(w / 2**a), qmin, qmax) * 2**a
                score.append(lp2(ref, out_of(wq)))
            alpha[j] = alpha_floor[j] - 1 + score.index(min(score))
        scale = 2**alpha
        scale.clamp_(EPS)
        return scale


def ptf_params(x, bit='int8'):
    """PtfObserver (ptf.py:14-134): per-channel power-of-two factor in {1,2,4,8} on a float base scale."""
    qmin, qmax = BITS[bit]
    r = _obs_rows(x, 'activation')
    max_val, min_val = r.max(axis=1).values, r.min(axis=1).values
    max_t = torch.max(-min_val.min(), max_val.max())
    scale8 = 2 * max_t / float(qmax - qmin)
    scale8.clamp_(EPS)
    scale4 = scale8 / 2
    scale2 = scale4 / 2
    scale1 = scale2 / 2
    mask = torch.ones_like(max_val)
    for j in range(x.shape[2]):
        d = x[..., j].unsqueeze(-1)
        score = [lp2(d, torch.clamp(torch.round(d / s), qmin, qmax) * s) for s in (scale1, scale2, scale4, scale8)]
        mask[j] *= 2**score.index(min(score))
    return scale1 * mask


def lis_float(x, sf):
    """QIntSoftmax.forward exactly as the reference runs it DURING CALIBRATION, i.e. on un-quantized
    scores (models/ptq/layers.py:331-376).  Plain fp32 torch ops, reference order."""
    x_int = x / sf
    x_int = x_int - x_int.max(dim=-1, keepdim=True)[0]
    x0_int = torch.floor(-0.6931 / sf)
    x_int = torch.max(x_int, 32 * x0_int)
    q = torch.floor(x_int / x0_int)
    r = x_int - x0_int * q
    b_int = torch.floor((0.96963238 / 0.35815147) / sf)
    c_int = torch.floor((1. / 0.35815147) / sf**2)
    z = r * (r + b_int) + c_int
    exp_int = torch.clamp(torch.floor(z * 2**(32 - q)), min=0)
    s = exp_int.sum(dim=-1, keepdim=True)
    ratio = torch.round(s / exp_int)
    big = ratio.log2().floor()
    big = big + ((ratio - 2**big) >= 2**(big - 1))
    out = 2**(-torch.clamp(big, 0, 15))
    out[big >= 16] = 0
    return out


# --------------------------------------------------------------------------------------------------
# canonical integer pieces of the quantized forward
# --------------------------------------------------------------------------------------------------
def _exp2(n):
    return torch.ldexp(torch.ones_like(n, dtype=torch.float32), n.to(torch.int32))


def _floor_log2(x):
    """exact binary exponent of |x|; -inf for x == 0 like floor(log2(0)) in the reference's get_MN (layers.py:236: a zero
    multiplier - gamma == 0 - ends with N = 31, M = 0)."""
    _, e = torch.frexp(x)
    out = (e - 1).to(torch.float32)
    return torch.where(x == 0, torch.full_like(out, float('-inf')), out)


def lis_consts(sf):
    """x0_int, b_int, c_int of the I-BERT polynomial (layers.py:334-351), fp32 like the reference."""
    sf = sf.reshape(()).float()
    x0 = torch.floor(-0.6931 / sf)
    b = torch.floor((0.96963238 / 0.35815147) / sf)
    c = torch.floor((1. / 0.35815147) / sf**2)
    return int(x0), int(b), int(c)


def lis_int(codes, sf):
    """Log-Int-Softmax on integer score codes -> exponents k (16 means 'zero').  layers.py:323-376.
    exp_int = z * 2^(32-q) is summed exactly in int64, then rounded once to fp32."""
    x0, b, c = lis_consts(sf)
    xi = codes.to(torch.int64)
    xi = xi - xi.max(dim=-1, keepdim=True)[0]
    xi = torch.clamp(xi, min=32 * x0)
    q = torch.div(xi, x0, rounding_mode='floor')
    r = xi - x0 * q
    z = r * (r + b) + c
    e = z << (32 - q)
    e = torch.clamp(e, min=0)
    s = e.sum(dim=-1, keepdim=True)
    ratio = torch.round(s.float() / e.float())
    m, ex = torch.frexp(ratio)
    k = (ex - 1) + (m >= 0.75)
    return torch.clamp(k, 0, 16)


def lis_probs(k):
    p = torch.ldexp(torch.ones_like(k, dtype=torch.float32), -k.to(torch.int32))
    return torch.where(k >= 16, torch.zeros_like(p), p)


def int_layernorm(x, in_scale, gamma, beta, out_scale):
    """QIntLayerNorm.forward, mode 'int' (layers.py:255-289); returns the UNCLAMPED integer output.
    sum(x_q) and sum(x_q^2) are exact; everything after is fp32 in the reference's order."""
    C = x.shape[-1]
    in_scale = in_scale.reshape(1, 1, -1)
    out_scale = out_scale.reshape(1, 1, -1)
    g = gamma.reshape(1, 1, -1)
    bt = beta.reshape(1, 1, -1)
    x_q = (x / in_scale).round()
    s1 = in_scale.min()
    x_q = x_q * (in_scale / s1).round()
    xd = x_q.double()
    S1 = xd.sum(dim=-1).float()
    S2 = (xd * xd).sum(dim=-1).float()
    mean = (S1 / C) * s1
    # torch.sqrt on CPU goes through MKL VML and is NOT correctly rounded (0x4e3d094f -> ...1b on Xeon, ...1c on
    # EPYC): take the IEEE value (fp64 sqrt rounded once), which is what v_sqrt_f32 / CUDA sqrtf return
    std = (s1 / C) * torch.sqrt((C * S2 - S1 * S1).double()).float()
    A = (s1 / std).unsqueeze(-1) * g / out_scale
    sign = A.sign()
    absA = A.abs()
    N = torch.clamp(7 - _floor_log2(absA), 0, 31)
    pN = _exp2(N)
    M = torch.clamp(torch.floor(absA * pN), 0, 255)
    B = ((bt - (mean / std).unsqueeze(-1) * g) / out_scale * pN).round()
    return ((sign * M * x_q + B) / pN).round()


def int_layernorm_torchcpu(x, in_scale, gamma, beta, out_scale):
    """QIntLayerNorm.forward mode 'int' with the reference's own fp32 torch ops (layers.py:234-238, 255-289): fp32 ``sum`` /
    ``mean`` of integers above 2^24, ``torch.sqrt``, ``torch.log2``, ``torch.pow`` -- platform-dependent roundings included.
    Used only by ``imitate={'ln'}``."""
    C = x.shape[-1]
    in_scale = in_scale.reshape(1, 1, -1)
    out_scale = out_scale.reshape(1, 1, -1)
    x_q = (x / in_scale).round()
    s1 = in_scale.min()
    x_q = x_q * (in_scale / s1).round()
    mean = x_q.mean(dim=-1) * s1
    std = (s1 / C) * torch.sqrt(C * (x_q**2).sum(dim=-1) - x_q.sum(dim=-1)**2)
    A = (s1 / std).unsqueeze(-1) * gamma.reshape(1, 1, -1) / out_scale
    N = torch.clamp(7 - torch.floor(torch.log2(A.abs())), 0, 31)
    M = torch.clamp(torch.floor(A.abs() * torch.pow(2, N)), 0, 255)
    B = ((beta.reshape(1, 1, -1) - (mean / std).unsqueeze(-1) * gamma.reshape(1, 1, -1)) / out_scale * torch.pow(2, N)).round()
    return ((A.sign() * M * x_q + B) / torch.pow(2, N)).round()


def gelu_rn(y):
    """correctly-rounded fp32 GELU (reference: float nn.GELU, layers_quant.py:147,331)."""
    yd = y.double()
    return (0.5 * yd * torch.erfc(-yd * math.sqrt(0.5))).float()


def weight_codes(w, cs, scale, bit):
    """integer weight codes of QLinear/QConv2d for one bit width (layers.py:173-178, uniform.py:82-88).
    ``cs`` is the SmoothQuant channel scale folded into the weight (vit_fquant.py:285)."""
    lo, hi = BITS['int%d' % bit]
    if cs is not None:
        w = w * cs.reshape((1, -1))
    w2 = w.reshape(w.shape[0], -1)
    return torch.clamp(torch.round(w2 / scale.reshape(-1, 1)), lo, hi)


def qgemm(x_codes, s_x, w_codes, s_w, bias):
    """F.linear on fake-quantised operands: exact integer accumulation, scale, then ONE rounding for the
    fp32 bias add."""
    # integer-valued sum of products accumulated in fp64 (exact below 2^53 whatever the order), then ENFORCED to fit fp32's 24 bits:
    # the reference's fp32 F.linear is order-independent only under that bound, and so is the cast below
    acc = x_codes.double() @ w_codes.double().t()
    amax = float(acc.abs().max()) if acc.numel() else 0.0
    assert amax < 2.0 ** 24, 'qgemm: |acc| = %g does not fit 24 bits: the reference result would depend on the summation order' % amax
    acc = acc.float()
    y = acc * (s_x * s_w).reshape(1, -1)
    return y + bias if bias is not None else y


# --------------------------------------------------------------------------------------------------
class OracleViT:
    """Functional ViT/DeiT: weights are a timm-style state_dict, calibration state is a flat dict."""

    IMITATE = ('linear', 'gelu', 'lis', 'ln')

    def __init__(self, arch, state_dict, in_chans=3, ln_eps=1e-6, imitate=()):
        self.a = dict(arch)
        self.W = {k: v.detach().float() for k, v in state_dict.items()}
        self.in_chans = in_chans
        self.ln_eps = ln_eps
        self.calib = None
        self.global_distance = []
        # which canonical pieces of quant_forward are swapped back for the torch-CPU op the reference calls (module docstring)
        self.imitate = frozenset(imitate)
        assert self.imitate <= set(self.IMITATE), self.imitate

    # ----- the four places where the reference's result depends on the machine ---------------------------
    def _linear(self, x_codes, s_x, w_codes, s_w, bias):
        if 'linear' in self.imitate:       # F.linear on the fake-quantised fp32 operands, bias inside (layers.py:178)
            return F.linear(x_codes * s_x, w_codes * s_w.reshape(-1, 1), bias)
        return qgemm(x_codes, s_x, w_codes, s_w, bias)

    def _gelu(self, y):
        return F.gelu(y) if 'gelu' in self.imitate else gelu_rn(y)      # nn.GELU (layers_quant.py:147,331)

    def _lis_k(self, codes, sf):
        if 'lis' in self.imitate:          # the all-fp32 int_softmax incl. its fp32 sum (layers.py:331-376)
            p = lis_float(codes * sf, sf)
            return torch.where(p > 0, -torch.log2(p.clamp(min=1e-30)), torch.full_like(p, 16.0)).round().to(torch.int64)
        return lis_int(codes, sf)

    def _ln(self, x, in_scale, gamma, beta, out_scale):
        return (int_layernorm_torchcpu if 'ln' in self.imitate else int_layernorm)(x, in_scale, gamma, beta, out_scale)

    def float_features(self, x):
        """the float model's input to the head (final norm, cls row): used by oracle/gen_golden.py to plant class margins."""
        return self._float_trunk(x)

    # ----- plain float forward (bit-for-bit what the reference computes before any calibration) ------
    def float_forward(self, x):
        return F.linear(self._float_trunk(x), self.W['head.weight'], self.W['head.bias'])

    def _float_trunk(self, x):
        W, a = self.W, self.a
        D, H = a['embed_dim'], a['num_heads']
        x = F.conv2d(x, W['patch_embed.proj.weight'], W['patch_embed.proj.bias'], a['patch_size'])
        x = x.flatten(2).transpose(1, 2)
        x = torch.cat((W['cls_token'].expand(x.shape[0], -1, -1), x), dim=1) + W['pos_embed']
        for i in range(a['depth']):
            p = 'blocks.%d.' % i
            h = F.layer_norm(x, (D,), W[p + 'norm1.weight'], W[p + 'norm1.bias'], self.ln_eps)
            h = self._attn_float(h, p, H)
            x = x + h
            h = F.layer_norm(x, (D,), W[p + 'norm2.weight'], W[p + 'norm2.bias'], self.ln_eps)
            h = F.linear(F.gelu(F.linear(h, W[p + 'mlp.fc1.weight'], W[p + 'mlp.fc1.bias'])),
                         W[p + 'mlp.fc2.weight'], W[p + 'mlp.fc2.bias'])
            x = x + h
        return F.layer_norm(x, (D,), W['norm.weight'], W['norm.bias'], self.ln_eps)[:, 0]

    def _attn_float(self, h, p, H):
        W = self.W
        B, N, C = h.shape
        # the reference evaluates the SmoothQuant search even un-quantised and returns the *smoothed*
        # product (vit_fquant.py:241-247,280); x/cs and W*cs are exact power-of-two scalings
        qkv = F.linear(h, W[p + 'attn.qkv.weight'], W[p + 'attn.qkv.bias'])
        qkv = qkv.reshape(B, N, 3, H, C // H).permute(2, 0, 3, 1, 4)
        attn = (qkv[0] @ qkv[1].transpose(-2, -1)) * ((C // H) ** -0.5)
        attn = attn.softmax(dim=-1)
        o = (attn @ qkv[2]).transpose(1, 2).reshape(B, N, C)
        return F.linear(o, W[p + 'attn.proj.weight'], W[p + 'attn.proj.bias'])

    # ----- calibration ------------------------------------------------------------------------------
    def _calib_act(self, name, x):
        """QAct in calibrate+last_calibrate state, minmax observer, layer-wise int8 (layers.py:207-218)."""
        ob = _MinMax('activation')
        ob.update(x, 'layer_wise')
        self.calib[name] = ob.params(x, 'int8', 'layer_wise')

    def _calib_weight(self, name, w, x, module_type, others):
        """QLinear/QConv2d calibrate loop over the four bit types (layers.py:57-71,148-170)."""
        ob = _MinMax(module_type)
        dic, dist = {}, []
        for bit in CALIB_BIT_ORDER:
            mode = 'layer_wise' if bit == 'int8' else 'channel_wise'
            ob.update(w, mode)
            dic[bit] = ob.params(x, bit, mode, others)
            if module_type == 'linear_weight':
                lo, hi = BITS[bit]
                dist.append(lp2(w, fake_quant(w, dic[bit], lo, hi, (-1, 1))))
        if module_type == 'linear_weight':
            self.global_distance.append(dist)
        self.calib[name] = dic

    def _smooth_search(self, prefix, lin, x, alphas):
        """SmoothQuant-PoT search (vit_fquant.py:199-280 / layers_quant.py:222-303)."""
        W = self.W
        w, b = W[lin + '.weight'], W[lin + '.bias']
        gmax = torch.abs(x).max(axis=1).values.max(axis=0).values
        wmax = torch.abs(w).max(axis=0).values
        pool, act_s, w_s, loss = [], [], [], [[], []]
        for alpha in alphas:
            cs = 2**pot_round(gmax**alpha / (wmax**(1 - alpha)))
            pool.append(cs)
            xs = x / cs.reshape((1, 1, -1))
            ws = w * cs.reshape((1, -1))
            gt = F.linear(xs, ws, b)
            self._calib_act('_tmp', xs)
            act_s.append(self.calib.pop('_tmp'))
            self._calib_weight('_tmpw', ws, xs, 'linear_weight', [b])
            w_s.append(self.calib.pop('_tmpw'))
            xq = fake_quant(xs, act_s[-1], -128, 127)
            for j, bit in enumerate(BIT_POOL):
                nm = 'int%d' % bit
                lo, hi = BITS[nm]
                out = F.linear(xq, fake_quant(ws, w_s[-1][nm], lo, hi, (-1, 1)), b)
                loss[j].append((gt - out).abs().pow(2.0).mean())
        c = self.calib
        c[prefix + '.best_scale'], c[prefix + '.best_act_scale'], c[prefix + '.best_weight_scale'] = [], [], []
        for l in loss:
            i = l.index(min(l))
            c[prefix + '.best_scale'].append(pool[i])
            c[prefix + '.best_act_scale'].append(act_s[i])
            c[prefix + '.best_weight_scale'].append(w_s[i])
        return gt

    def calibrate(self, x):
        """One calibration forward with calibrate + last_calibrate open (test_quant.py:235-249).
        Returns the logits of that pass (float path with LIS softmax on float scores)."""
        W, a = self.W, self.a
        D, H = a['embed_dim'], a['num_heads']
        self.calib, self.global_distance = {}, []
        if self.a.get('input_quant', True):        # VisionTransformer(input_quant=False) has no input QAct (vit_fquant.py:524,705)
            self._calib_act('qact_input', x)
        w, b = W['patch_embed.proj.weight'], W['patch_embed.proj.bias']
        ob = _MinMax('conv_weight')
        dic = {}
        for bit in CALIB_BIT_ORDER:
            mode = 'layer_wise' if bit == 'int8' else 'channel_wise'
            ob.update(w, mode)
            dic[bit] = ob.params(x, bit, mode, [b, a['patch_size'], 0, 1, 1])
        self.calib['patch_embed.proj'] = dic
        x = F.conv2d(x, w, b, a['patch_size']).flatten(2).transpose(1, 2)
        self._calib_act('patch_embed.qact', x)
        x = torch.cat((W['cls_token'].expand(x.shape[0], -1, -1), x), dim=1)
        self._calib_act('qact_embed', x)
        self._calib_act('qact_pos', W['pos_embed'])
        x = x + W['pos_embed']
        self.calib['qact1'] = ptf_params(x)
        for i in range(a['depth']):
            p = 'blocks.%d.' % i
            B, N, C = x.shape
            h = F.layer_norm(x, (D,), W[p + 'norm1.weight'], W[p + 'norm1.bias'], self.ln_eps)
            h = self._smooth_search(p + 'attn', p + 'attn.qkv', h, ALPHA_ATTN)
            self._calib_act(p + 'attn.qact1', h)
            qkv = h.reshape(B, N, 3, H, C // H).permute(2, 0, 3, 1, 4)
            attn = (qkv[0] @ qkv[1].transpose(-2, -1)) * ((C // H) ** -0.5)
            self._calib_act(p + 'attn.qact_attn1', attn)
            attn = lis_float(attn, self.calib[p + 'attn.qact_attn1'])
            h = (attn @ qkv[2]).transpose(1, 2).reshape(B, N, C)
            self._calib_act(p + 'attn.qact2', h)
            self._calib_weight(p + 'attn.proj', W[p + 'attn.proj.weight'], h, 'linear_weight', [W[p + 'attn.proj.bias']])
            h = F.linear(h, W[p + 'attn.proj.weight'], W[p + 'attn.proj.bias'])
            self.calib[p + 'attn.qact3'] = ptf_params(h)
            x = x + h
            self.calib[p + 'qact2'] = ptf_params(x)
            h = F.layer_norm(x, (D,), W[p + 'norm2.weight'], W[p + 'norm2.bias'], self.ln_eps)
            h = self._smooth_search(p + 'mlp', p + 'mlp.fc1', h, ALPHA_MLP)
            h = F.gelu(h)
            self._calib_act(p + 'mlp.qact1', h)
            self._calib_weight(p + 'mlp.fc2', W[p + 'mlp.fc2.weight'], h, 'linear_weight', [W[p + 'mlp.fc2.bias']])
            h = F.linear(h, W[p + 'mlp.fc2.weight'], W[p + 'mlp.fc2.bias'])
            self.calib[p + 'mlp.qact2'] = ptf_params(h)
            x = x + h
            self.calib[p + 'qact4'] = ptf_params(x)
        x = F.layer_norm(x, (D,), W['norm.weight'], W['norm.bias'], self.ln_eps)[:, 0]
        self._calib_act('qact2', x)
        self._calib_weight('head', W['head.weight'], x, 'linear_weight', [W['head.bias']])
        x = F.linear(x, W['head.weight'], W['head.bias'])
        self._calib_act('act_out', x)
        return x

    # ----- quantized inference forward (THE HOT PATH) -------------------------------------------------
    def quant_forward(self, x, bit_config, taps=None):
        """model(x, bit_config) after model_close_calibrate(); model_quant() (vit_fquant.py:780-799).
        Returns fp32 logits on the int8 grid of ``act_out``.  ``taps`` (dict) receives integer codes of
        the named intermediate activations."""
        W, a, c = self.W, self.a, self.calib
        D, H, depth = a['embed_dim'], a['num_heads'], a['depth']
        hd = D // H
        assert len(bit_config) == 4 * depth + 2
        for b in bit_config:
            if b not in BIT_POOL:
                raise ValueError('%r is not in list' % (b,))   # bit_pool.index(bit) vit_fquant.py:282

        def tap(name, t):
            if taps is not None:
                taps[name] = t.to(torch.int32) if t.dtype != torch.int64 else t

        def q8(v, s):
            return torch.clamp(torch.round(v / s), -128, 127)

        # qact_input -> PatchEmbed (QConv2d k=stride=patch) -> qact     vit_fquant.py:705-715
        fp_in = not a.get('input_quant', True)     # the vit_large factory (vit_fquant.py:925): the fp32 image feeds the conv
        s_in = torch.ones(1) if fp_in else c['qact_input']
        q = x if fp_in else q8(x, s_in)
        if not fp_in:
            tap('qact_input', q)
        bit = bit_config[0]
        s_w = c['patch_embed.proj']['int%d' % bit]
        wq = weight_codes(W['patch_embed.proj.weight'], None, s_w, bit)
        P = a['patch_size']
        Bn = x.shape[0]
        if 'linear' in self.imitate:       # F.conv2d on the fake-quantised operands (layers.py:87)
            y = F.conv2d(q * s_in, (wq * s_w.reshape(-1, 1)).reshape(W['patch_embed.proj.weight'].shape), W['patch_embed.proj.bias'], stride=P)
            y = y.flatten(2).transpose(1, 2)
        else:
            cols = F.unfold(q, kernel_size=P, stride=P).transpose(1, 2)          # [B, patches, C*P*P]
            if fp_in:
                # canonical reading of F.conv2d(x_fp32, code_w * s_w, bias): the products x * code are exact in fp64; their fp64 sum,
                # times the power-of-two s_w, plus the bias, rounded to fp32 ONCE (the fp32 conv of a given platform differs from this
                # by its own accumulation order)
                acc = cols.reshape(-1, cols.shape[-1]).double() @ wq.double().t()
                y = (acc * s_w.double().reshape(1, -1).expand(1, wq.shape[0]) + W['patch_embed.proj.bias'].double()).float()
            else:
                y = qgemm(cols.reshape(-1, cols.shape[-1]), s_in, wq, s_w.reshape(-1), W['patch_embed.proj.bias'])
            y = y.reshape(Bn, -1, D)
        s_pe = c['patch_embed.qact']
        xv = q8(y, s_pe) * s_pe
        tap('patch_embed.qact', q8(y, s_pe))
        # cls concat, qact_embed, + qact_pos(pos_embed), qact1 (PTF)         vit_fquant.py:718-733
        xv = torch.cat((W['cls_token'].expand(Bn, -1, -1), xv), dim=1)
        s_e = c['qact_embed']
        xv = q8(xv, s_e) * s_e
        s_p = c['qact_pos']
        xv = xv + q8(W['pos_embed'], s_p) * s_p
        s_res = c['qact1']
        qr = q8(xv, s_res.reshape(1, 1, -1))
        tap('qact1', qr)
        xv = qr * s_res.reshape(1, 1, -1)

        for i in range(depth):
            p = 'blocks.%d.' % i
            bits = bit_config[4 * i + 1: 4 * i + 5]
            # ---- attention ------------------------------------------------------------------------
            bi = BIT_POOL.index(bits[0])
            cs = c[p + 'attn.best_scale'][bi]
            s_a0 = c[p + 'attn.best_act_scale'][bi]
            s_wq = c[p + 'attn.best_weight_scale'][bi]['int%d' % bits[0]]
            ln = self._ln(xv, s_res, W[p + 'norm1.weight'], W[p + 'norm1.bias'], s_a0 * cs)
            h = ln * (s_a0 * cs).reshape(1, 1, -1)
            q0 = q8(h / cs.reshape((1, 1, -1)), s_a0)
            tap(p + 'attn.qact0', q0)
            wq = weight_codes(W[p + 'attn.qkv.weight'], cs, s_wq, bits[0])
            N = q0.shape[1]
            y = self._linear(q0.reshape(-1, D), s_a0, wq, s_wq.reshape(-1), W[p + 'attn.qkv.bias']).reshape(Bn, N, 3 * D)
            s_q1 = c[p + 'attn.qact1']
            q1 = q8(y, s_q1)
            tap(p + 'attn.qact1', q1)
            qkv = q1.reshape(Bn, N, 3, H, hd).permute(2, 0, 3, 1, 4)
            acc = qkv[0] @ qkv[1].transpose(-2, -1)                          # exact integers
            attn = (acc * (s_q1 * s_q1)) * (hd ** -0.5)                      # (q@k^T)*scale, vit_fquant.py:316
            s_at = c[p + 'attn.qact_attn1']
            sc = q8(attn, s_at)
            tap(p + 'attn.qact_attn1', sc)
            k = self._lis_k(sc, s_at)
            tap(p + 'attn.softmax_k', k)
            o = (lis_probs(k) @ (qkv[2] * s_q1)).transpose(1, 2).reshape(Bn, N, D)   # exact (dyadic sums)
            s_a2 = c[p + 'attn.qact2']
            q2 = q8(o, s_a2)
            tap(p + 'attn.qact2', q2)
            s_wp = c[p + 'attn.proj']['int%d' % bits[1]]
            wq = weight_codes(W[p + 'attn.proj.weight'], None, s_wp, bits[1])
            y = self._linear(q2.reshape(-1, D), s_a2, wq, s_wp.reshape(-1), W[p + 'attn.proj.bias']).reshape(Bn, N, D)
            s_a3 = c[p + 'attn.qact3'].reshape(1, 1, -1)
            q3 = q8(y, s_a3)
            tap(p + 'attn.qact3', q3)
            s_b2 = c[p + 'qact2']
            qr = q8(xv + q3 * s_a3, s_b2.reshape(1, 1, -1))
            tap(p + 'qact2', qr)
            xv = qr * s_b2.reshape(1, 1, -1)
            s_res = s_b2
            # ---- MLP  (norm2 receives the ATTENTION channel scale: vit_fquant.py:464) -----------------
            bm = BIT_POOL.index(bits[2])
            cs_m = c[p + 'mlp.best_scale'][bm]
            s_m0 = c[p + 'mlp.best_act_scale'][bm]
            s_w1 = c[p + 'mlp.best_weight_scale'][bm]['int%d' % bits[2]]
            ln = self._ln(xv, s_res, W[p + 'norm2.weight'], W[p + 'norm2.bias'], s_m0 * cs)
            h = ln * (s_m0 * cs).reshape(1, 1, -1)
            q0 = q8(h / cs_m.reshape((1, 1, -1)), s_m0)
            tap(p + 'mlp.qact0', q0)
            wq = weight_codes(W[p + 'mlp.fc1.weight'], cs_m, s_w1, bits[2])
            y = self._linear(q0.reshape(-1, D), s_m0, wq, s_w1.reshape(-1), W[p + 'mlp.fc1.bias'])
            s_m1 = c[p + 'mlp.qact1']
            q1 = q8(self._gelu(y), s_m1)
            tap(p + 'mlp.qact1', q1.reshape(Bn, N, -1))
            s_w2 = c[p + 'mlp.fc2']['int%d' % bits[3]]
            wq = weight_codes(W[p + 'mlp.fc2.weight'], None, s_w2, bits[3])
            y = self._linear(q1, s_m1, wq, s_w2.reshape(-1), W[p + 'mlp.fc2.bias']).reshape(Bn, N, D)
            s_m2 = c[p + 'mlp.qact2'].reshape(1, 1, -1)
            q2 = q8(y, s_m2)
            tap(p + 'mlp.qact2', q2)
            s_b4 = c[p + 'qact4']
            qr = q8(xv + q2 * s_m2, s_b4.reshape(1, 1, -1))
            tap(p + 'qact4', qr)
            xv = qr * s_b4.reshape(1, 1, -1)
            s_res = s_b4

        # final norm (cls row only is consumed), qact2, head, act_out        vit_fquant.py:766-796
        s_f = c['qact2']
        ln = self._ln(xv[:, :1], s_res, W['norm.weight'], W['norm.bias'], s_f.expand(D))
        qf = q8(ln[:, 0] * s_f, s_f)
        tap('qact2', qf)
        bit = bit_config[-1]
        s_wh = c['head']['int%d' % bit]
        wq = weight_codes(W['head.weight'], None, s_wh, bit)
        y = self._linear(qf, s_f, wq, s_wh.reshape(-1), W['head.bias'])
        s_o = c['act_out']
        ql = q8(y, s_o)
        tap('act_out', ql)
        return ql * s_o

    def flops(self):
        """FLOPs list returned by the reference forward (MAC counts of the 4*depth+2 linear layers):
        layers_quant.py:482,329,344; vit_fquant.py:304,336,794."""
        a = self.a
        D, P = a['embed_dim'], a['patch_size']
        g = a['img_size'] // P
        N = g * g + 1
        Hd = int(D * a['mlp_ratio'])
        out = [self.in_chans * P * P * D * g * g]
        for _ in range(a['depth']):
            out += [N * D * 3 * D, N * D * D, N * D * Hd, N * Hd * D]
        out.append(D * a['num_classes'])
        return out


# --------------------------------------------------------------------------------------------------
def extract_calib(model):
    """Read the calibrated state out of a module tree that exposes the REFERENCE attribute names
    (works for the imported reference model and for the product's drop-in classes alike)."""
    def sc(q):
        return q.quantizer.scale.detach().clone().float()

    def dic(l):
        return {k: v.detach().clone().float() for k, v in l.quantizer.dic_scale.items()}

    c = {'patch_embed.proj': dic(model.patch_embed.proj),
         'patch_embed.qact': sc(model.patch_embed.qact), 'qact_embed': sc(model.qact_embed),
         'qact_pos': sc(model.qact_pos), 'qact1': sc(model.qact1), 'qact2': sc(model.qact2),
         'head': dic(model.head), 'act_out': sc(model.act_out)}
    if getattr(model, 'input_quant', True):
        c['qact_input'] = sc(model.qact_input)
    for i, blk in enumerate(model.blocks):
        p = 'blocks.%d.' % i
        for nm, m in ((p + 'attn', blk.attn), (p + 'mlp', blk.mlp)):
            c[nm + '.best_scale'] = [t.detach().clone().float() for t in m.best_scale]
            c[nm + '.best_act_scale'] = [t.detach().clone().float() for t in m.best_act_scale]
            c[nm + '.best_weight_scale'] = [{k: v.detach().clone().float() for k, v in d.items()}
                                            for d in m.best_weight_scale]
        c[p + 'attn.qact1'] = sc(blk.attn.qact1)
        c[p + 'attn.qact_attn1'] = sc(blk.attn.qact_attn1)
        c[p + 'attn.qact2'] = sc(blk.attn.qact2)
        c[p + 'attn.proj'] = dic(blk.attn.proj)
        c[p + 'attn.qact3'] = sc(blk.attn.qact3)
        c[p + 'qact2'] = sc(blk.qact2)
        c[p + 'mlp.qact1'] = sc(blk.mlp.qact1)
        c[p + 'mlp.fc2'] = dic(blk.mlp.fc2)
        c[p + 'mlp.qact2'] = sc(blk.mlp.qact2)
        c[p + 'qact4'] = sc(blk.qact4)
    return c


def flatten_calib(c):
    """flat {str: tensor} view (for .npz fixtures and comparisons)."""
    out = {}
    for k, v in c.items():
        if isinstance(v, dict):
            for b, t in v.items():
                out['%s/%s' % (k, b)] = t
        elif isinstance(v, list):
            for i, t in enumerate(v):
                if isinstance(t, dict):
                    for b, u in t.items():
                        out['%s/%d/%s' % (k, i, b)] = u
                else:
                    out['%s/%d' % (k, i)] = t
        else:
            out[k] = v
    return out


def unflatten_calib(flat):
    c = {}
    for k, v in flat.items():
        v = torch.as_tensor(v).float()
        parts = k.split('/')
        if len(parts) == 1:
            c[k] = v
        elif len(parts) == 2 and not parts[1].isdigit():
            c.setdefault(parts[0], {})[parts[1]] = v
        elif len(parts) == 2:
            lst = c.setdefault(parts[0], [])
            i = int(parts[1])
            while len(lst) <= i:
                lst.append(None)
            lst[i] = v
        else:
            lst = c.setdefault(parts[0], [])
            i = int(parts[1])
            while len(lst) <= i:
                lst.append(None)
            if lst[i] is None:
                lst[i] = {}
            lst[i][parts[2]] = v
    return c

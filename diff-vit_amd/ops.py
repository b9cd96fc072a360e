"""``torch.ops.p2vit.*``: the C-ABI entry points registered as PyTorch custom ops ("bound via
torch.library custom ops").  Each op is a thin argument adapter over ``engine.lib()``; tensors are int8 code tensors or
fp32 parameter vectors on the GPU, results are freshly allocated tensors on the same device and stream.  There is no CPU
kernel behind any of them: calling one with CPU tensors raises ``NotImplementedError`` from the dispatcher, and calling
one without the built library raises ``RuntimeError`` from ``engine.lib()``.

    torch.ops.p2vit.fake_quant(x, scale, inner, lo, hi)                         UniformQuantizer.forward   quantizer/uniform.py:82-88
    torch.ops.p2vit.quantize_patchify(images, inv_s, patch)                     qact_input + im2col        vit_fquant.py:705, layers.py:55-88
    torch.ops.p2vit.linear_requant(x, w, colscale, bias, inv_s_out)             QLinear + QAct             layers.py:133-178, 207-220
    torch.ops.p2vit.linear_gelu_requant(x, w, colscale, bias, inv_s_out)        fc1 + GELU + qact1         layers_quant.py:304-351
    torch.ops.p2vit.int_layernorm(x, s1, mask, gamma, beta, inv_out, post_mul)  QIntLayerNorm 'int'        layers.py:255-289
    torch.ops.p2vit.lis_attention(qkv, heads, s_qkv_sq, qk_scale, inv_s_attn, av_mul, x0, b, c)   vit_fquant.py:309-326, layers.py:323-376
    torch.ops.p2vit.forward(plan_handle, images, bit_config)                    VisionTransformer.forward  vit_fquant.py:780-799
"""
import ctypes as C

import torch

from . import engine as E

_LIB = torch.library.Library('p2vit', 'DEF')
_LIB.define('fake_quant(Tensor x, Tensor scale, int inner, int lo, int hi) -> Tensor')
_LIB.define('quantize_patchify(Tensor images, float inv_s, int patch) -> Tensor')
_LIB.define('linear_requant(Tensor x, Tensor w, Tensor colscale, Tensor bias, float inv_s_out) -> Tensor')
_LIB.define('linear_gelu_requant(Tensor x, Tensor w, Tensor colscale, Tensor bias, float inv_s_out) -> Tensor')
_LIB.define('int_layernorm(Tensor x, float s1, Tensor mask, Tensor gamma, Tensor beta, Tensor inv_out, Tensor post_mul) -> Tensor')
_LIB.define('lis_attention(Tensor qkv, int heads, float s_qkv_sq, float qk_scale, float inv_s_attn, float av_mul, int x0, int b, int c) -> Tensor')
_LIB.define('forward(int plan, Tensor images, int[] bit_config) -> Tensor')


def _f32(t):
    return t.contiguous().float()


def _fake_quant(x, scale, inner, lo, hi):
    x, scale = _f32(x), _f32(scale)
    out = torch.empty_like(x)
    E.check(E.lib().p2v_fake_quant_f32(E.ptr(x), x.numel(), E.ptr(scale), scale.numel(), inner, lo, hi, E.ptr(out), None,
                                       E.stream_ptr()))
    return out


def _quantize_patchify(images, inv_s, patch):
    images = _f32(images)
    B, Cin, H, W = images.shape
    k = Cin * patch * patch
    k_pad = (k + 63) // 64 * 64
    out = torch.zeros(B * (H // patch) * (W // patch), k_pad, dtype=torch.int8, device=images.device)
    E.check(E.lib().p2v_quantize_patchify(E.ptr(images), B, Cin, H, W, patch, inv_s, E.ptr(out), k_pad, E.stream_ptr()))
    return out


def _linear(kind, x, w, colscale, bias, inv_s_out):
    x = x.contiguous()
    M, K = x.shape
    N = w.shape[0]
    n_pad = (N + 127) // 128 * 128            # the GEMM stages whole 128-row weight tiles
    wp = torch.zeros(n_pad, K, dtype=torch.int8, device=x.device)
    wp[:N] = w
    w = wp
    cs = torch.zeros(n_pad, device=x.device)
    cs[:N] = colscale
    bs = torch.zeros(n_pad, device=x.device)
    bs[:N] = bias
    colscale, bias = cs, bs
    lin = E.Linear(E.ptr(w), E.ptr(colscale), E.ptr(bias))
    epi = E.Epilogue()
    epi.inv_s_out = inv_s_out
    out = torch.empty(M, N, dtype=torch.int8, device=x.device)
    E.check(E.lib().p2v_gemm_i8(kind, E.ptr(x), K, M, K, N, C.byref(lin), C.byref(epi), E.ptr(out), N, None, E.stream_ptr()))
    return out


def _int_layernorm(x, s1, mask, gamma, beta, inv_out, post_mul):
    x = x.contiguous()
    rows, Cc = x.shape
    keep = [_f32(t) for t in (mask, gamma, beta, inv_out, post_mul)]
    ln = E.Ln(s1, *[E.ptr(t) for t in keep])
    out = torch.empty_like(x)
    E.check(E.lib().p2v_int_layernorm(E.ptr(x), Cc, rows, Cc, C.byref(ln), E.ptr(out), Cc, E.stream_ptr()))
    return out


def _lis_attention(qkv, heads, s_qkv_sq, qk_scale, inv_s_attn, av_mul, x0, b, c):
    qkv = qkv.contiguous()
    B, N, D3 = qkv.shape
    D = D3 // 3
    at = E.Attn(s_qkv_sq, qk_scale, inv_s_attn, av_mul, x0, b, c)
    out = torch.empty(B, N, D, dtype=torch.int8, device=qkv.device)
    E.check(E.lib().p2v_lis_attention(E.ptr(qkv), B, N, heads, D // heads, C.byref(at), E.ptr(out), None, E.stream_ptr()))
    return out


_PLANS = {}          # handle -> FrozenPlan, filled by plan.FrozenPlan (weak registry of live plans)


def _forward(plan, images, bit_config):
    p = _PLANS.get(plan)
    if p is None:
        raise RuntimeError('p2vit::forward: %d is not a live plan handle' % plan)
    return p.forward(images, list(bit_config))


_LIB.impl('fake_quant', _fake_quant, 'CUDA')
_LIB.impl('quantize_patchify', _quantize_patchify, 'CUDA')
_LIB.impl('linear_requant', lambda x, w, cs, b, inv: _linear(E.EPI_REQUANT, x, w, cs, b, inv), 'CUDA')
_LIB.impl('linear_gelu_requant', lambda x, w, cs, b, inv: _linear(E.EPI_GELU, x, w, cs, b, inv), 'CUDA')
_LIB.impl('int_layernorm', _int_layernorm, 'CUDA')
_LIB.impl('lis_attention', _lis_attention, 'CUDA')
_LIB.impl('forward', _forward, 'CUDA')

OPS = ('fake_quant', 'quantize_patchify', 'linear_requant', 'linear_gelu_requant', 'int_layernorm', 'lis_attention', 'forward')

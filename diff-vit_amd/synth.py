"""Deterministic synthetic weights / images for the PoT-PTQ ViT path.

There is no network for checkpoints or ImageNet, and the reference cannot travel to the GPU box,
so every full-size test, fixture and benchmark regenerates its tensors from a seed with this
generator.  It is counter based and uses only 64-bit integer arithmetic (splitmix64) followed by
one int -> float64 -> float32 conversion, so the same (seed, name, shape) gives bit-identical
tensors on any host (no libm, no SIMD-width dependence -- unlike ``torch.randn``).

The shapes/keys produced follow the timm/DeiT ``state_dict`` layout the reference loads
(models/vit_fquant.py:822-828): ``cls_token``, ``pos_embed``, ``patch_embed.proj.*``,
``blocks.{i}.{norm1,attn.qkv,attn.proj,norm2,mlp.fc1,mlp.fc2}.*``, ``norm.*``, ``head.*``.
"""
import zlib

import numpy as np
import torch

_MASK = (1 << 64) - 1


def _splitmix64(x: np.ndarray) -> np.ndarray:
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & np.uint64(_MASK)
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & np.uint64(_MASK)
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & np.uint64(_MASK)
    return z ^ (z >> np.uint64(31))


def _stream(seed: int, name: str, n: int) -> np.ndarray:
    """n uint64 words for (seed, name); counter based, so order of generation never matters."""
    base = (zlib.crc32(name.encode()) * 0x100000001B3 + seed * 0x9E3779B97F4A7C15) & _MASK
    with np.errstate(over='ignore'):
        ctr = np.arange(n, dtype=np.uint64) + np.uint64(base)
        return _splitmix64(_splitmix64(ctr))


def normal(seed: int, name: str, shape, std: float = 1.0, mean: float = 0.0) -> torch.Tensor:
    """Approximately normal (Irwin-Hall, 4 x 16-bit uniforms per sample), exact integer pipeline."""
    n = int(np.prod(shape))
    w = _stream(seed, name, n)
    s = ((w & np.uint64(0xFFFF)) + ((w >> np.uint64(16)) & np.uint64(0xFFFF)) +
         ((w >> np.uint64(32)) & np.uint64(0xFFFF)) + ((w >> np.uint64(48)) & np.uint64(0xFFFF)))
    # sum of 4 U{0..65535}: mean 2*65535, var 4*(65536^2-1)/12
    z = (s.astype(np.float64) - 2.0 * 65535.0) / np.sqrt(4.0 * (65536.0**2 - 1.0) / 12.0)
    return torch.from_numpy((z * std + mean).astype(np.float32).reshape(shape))


def uniform(seed: int, name: str, shape, lo: float = 0.0, hi: float = 1.0) -> torch.Tensor:
    n = int(np.prod(shape))
    w = _stream(seed, name, n) >> np.uint64(11)  # 53 bits
    u = w.astype(np.float64) / float(1 << 53)
    return torch.from_numpy((lo + (hi - lo) * u).astype(np.float32).reshape(shape))


def images(seed: int, batch: int, img_size: int = 224, in_chans: int = 3, offset: int = 0) -> torch.Tensor:
    """ImageNet-shaped synthetic batch.  Unit-normal pixels (the reference's --mode 1 Gaussian
    calibration input, test_quant.py:214-216) with a per-image contrast in U(0.6,1.4) and a per-image,
    per-channel DC offset ~N(0,0.7) -- normalised ImageNet images have both, and without them every
    image of a random-weight ViT lands on the same top-1.  Image i of the stream is independent of the
    batch size it is requested in."""
    out = []
    for i in range(batch):
        tag = 'image/%d' % (offset + i)
        px = normal(seed, tag, (in_chans, img_size, img_size))
        gain = uniform(seed, tag + '/gain', (1, 1, 1), 0.6, 1.4)
        dc = normal(seed, tag + '/dc', (in_chans, 1, 1), 0.7)
        out.append(px * gain + dc)
    return torch.stack(out) if out else torch.zeros(0, in_chans, img_size, img_size)


ARCHS = {
    # name: (img, patch, dim, depth, heads, classes, mlp_ratio)  -- vit_fquant.py:802-933
    'micro': dict(img_size=32, patch_size=8, embed_dim=64, depth=2, num_heads=2, num_classes=10, mlp_ratio=4.0),
    'deit_tiny': dict(img_size=224, patch_size=16, embed_dim=192, depth=12, num_heads=3, num_classes=1000, mlp_ratio=4.0),
    'deit_small': dict(img_size=224, patch_size=16, embed_dim=384, depth=12, num_heads=6, num_classes=1000, mlp_ratio=4.0),
    'deit_base': dict(img_size=224, patch_size=16, embed_dim=768, depth=12, num_heads=12, num_classes=1000, mlp_ratio=4.0),
    'vit_base': dict(img_size=224, patch_size=16, embed_dim=768, depth=12, num_heads=12, num_classes=1000, mlp_ratio=4.0),
    'vit_large': dict(img_size=224, patch_size=16, embed_dim=1024, depth=24, num_heads=16, num_classes=1000, mlp_ratio=4.0),
}


def vit_state_dict(arch: dict, seed: int = 0, in_chans: int = 3) -> dict:
    """Random-init weights with non-degenerate statistics.

    The reference's default ``trunc_normal_(std=.02)`` init gives the same top-1 for every image and
    never exercises a clamp; these statistics (matrices ~3x wider, non-zero
    biases, LN gamma in U(0.5,1.5), a few outlier channels) give distinct top-1 and saturating codes.
    The two residual-branch output projections (attn.proj, mlp.fc2) are kept small so that, like a trained
    network, the residual stream dominates each block: with unit-gain random branches the net is chaotic
    and a single +-1 code flip in block 0 decorrelates half of block 11 (measured), which would make the
    full-size fixtures a test of chaos rather than of arithmetic.
    """
    D = arch['embed_dim']
    P = arch['patch_size']
    T = (arch['img_size'] // P) ** 2 + 1
    H = int(D * arch['mlp_ratio'])
    sd = {}
    # small cls token / cls position: the cls row is then dominated by what attention gathers from the
    # image, so top-1 depends on the input
    sd['cls_token'] = normal(seed, 'cls_token', (1, 1, D), 0.02)
    pos = normal(seed, 'pos_embed', (1, T, D), 0.3)
    pos[:, 0] *= 0.05
    sd['pos_embed'] = pos
    sd['patch_embed.proj.weight'] = normal(seed, 'patch_embed.proj.weight', (D, in_chans, P, P), 0.06)
    sd['patch_embed.proj.bias'] = normal(seed, 'patch_embed.proj.bias', (D,), 0.05)

    def lin(prefix, out_f, in_f, std):
        w = normal(seed, prefix + '.weight', (out_f, in_f), std)
        # a few input channels with 4x larger weights, so SmoothQuant channel scales are not all equal
        boost = uniform(seed, prefix + '.boost', (in_f,)) > 0.97
        w = w * torch.where(boost, torch.tensor(4.0), torch.tensor(1.0)).reshape(1, -1)
        sd[prefix + '.weight'] = w.contiguous()
        sd[prefix + '.bias'] = normal(seed, prefix + '.bias', (out_f,), 0.05)

    for i in range(arch['depth']):
        p = 'blocks.%d.' % i
        for nm in ('norm1', 'norm2'):
            sd[p + nm + '.weight'] = uniform(seed, p + nm + '.weight', (D,), 0.5, 1.5)
            sd[p + nm + '.bias'] = normal(seed, p + nm + '.bias', (D,), 0.05)
        lin(p + 'attn.qkv', 3 * D, D, 0.06)
        lin(p + 'attn.proj', D, D, 0.02)
        lin(p + 'mlp.fc1', H, D, 0.06)
        lin(p + 'mlp.fc2', D, H, 0.01)
    sd['norm.weight'] = uniform(seed, 'norm.weight', (D,), 0.5, 1.5)
    sd['norm.bias'] = normal(seed, 'norm.bias', (D,), 0.05)
    lin('head', arch['num_classes'], D, 0.06)
    return sd


def swin_state_dict(template, seed):
    """deterministic non-degenerate fill of a Swin state_dict (same keys/shapes as ``template``, e.g. ``model.state_dict()``):
    LN gamma in U(0.6, 1.4), LN beta N(0, 0.1), biases N(0, 0.05), relative-position tables N(0, 0.5), matrices N(0, 0.6/sqrt(fan_in)),
    the patch-embed conv N(0, 0.08).  Integer buffers (relative_position_index, attn_mask) are kept."""
    import torch
    out = {}
    for k, v in template.items():
        shp = tuple(v.shape)
        if v.dtype != torch.float32 or 'index' in k or 'mask' in k:
            out[k] = v
        elif k.endswith(('norm.weight', 'norm1.weight', 'norm2.weight')):
            out[k] = uniform(seed, k, shp, 0.6, 1.4)
        elif 'norm' in k and k.endswith('bias'):
            out[k] = normal(seed, k, shp, 0.1)
        elif k.endswith('bias'):
            out[k] = normal(seed, k, shp, 0.05)
        elif 'table' in k:
            out[k] = normal(seed, k, shp, 0.5)
        else:
            out[k] = normal(seed, k, shp, 0.6 / (shp[-1] ** 0.5) if v.dim() == 2 else 0.08)
    return out

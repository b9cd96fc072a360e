"""Local checkpoint ingestion.  The reference fetches weights over the network
(``torch.hub.load_state_dict_from_url`` vit_fquant.py:822-828; ``load_weights_from_npz`` models/utils.py:12-197): there is no
network here, so these loaders take a LOCAL path and execute nothing from the file (``weights_only=True`` / ``allow_pickle=False``).

  * ``.pth`` / ``.pt``: a DeiT/timm state_dict, optionally under the key ``'model'`` (vit_fquant.py:828) -> ``load_state_dict(strict=False)``.
  * ``.npz``: the Google Brain Flax ViT layout (``embedding/kernel``, ``Transformer/encoderblock_i/...``), converted to the timm
    parameter names like models/utils.py:95-197 does.  Parity of this conversion is pinned only by a synthetic round trip
    (no real checkpoint is available offline): "parity unpinned" with respect to a downloaded file.
"""
import numpy as np
import torch


def _n2p(w, t=True):
    """Flax -> torch layout: conv HWIO -> OIHW, dense (in, out) -> (out, in)  (models/utils.py:20-30)."""
    if w.ndim == 4 and w.shape[0] == w.shape[1] == w.shape[2] == 1:
        w = w.flatten()
    if t:
        if w.ndim == 4:
            w = w.transpose([3, 2, 0, 1])
        elif w.ndim == 3:
            w = w.transpose([2, 0, 1])
        elif w.ndim == 2:
            w = w.transpose([1, 0])
    return torch.from_numpy(np.ascontiguousarray(w)).float()


def vit_npz_to_state_dict(w, depth, prefix=''):
    """{flax name: array} -> timm-style state_dict for ``VisionTransformer`` (qkv fused as [q; k; v] rows)."""
    sd = {'patch_embed.proj.weight': _n2p(w[prefix + 'embedding/kernel']), 'patch_embed.proj.bias': _n2p(w[prefix + 'embedding/bias']),
          'cls_token': _n2p(w[prefix + 'cls'], t=False),
          'pos_embed': _n2p(w[prefix + 'Transformer/posembed_input/pos_embedding'], t=False),
          'norm.weight': _n2p(w[prefix + 'Transformer/encoder_norm/scale']), 'norm.bias': _n2p(w[prefix + 'Transformer/encoder_norm/bias'])}
    if prefix + 'head/kernel' in w:
        sd['head.weight'] = _n2p(w[prefix + 'head/kernel'])
        sd['head.bias'] = _n2p(w[prefix + 'head/bias'])
    for i in range(depth):
        b = '%sTransformer/encoderblock_%d/' % (prefix, i)
        m = b + 'MultiHeadDotProductAttention_1/'
        p = 'blocks.%d.' % i
        sd[p + 'norm1.weight'], sd[p + 'norm1.bias'] = _n2p(w[b + 'LayerNorm_0/scale']), _n2p(w[b + 'LayerNorm_0/bias'])
        sd[p + 'attn.qkv.weight'] = torch.cat([_n2p(w[m + n + '/kernel'], t=False).flatten(1).T for n in ('query', 'key', 'value')])
        sd[p + 'attn.qkv.bias'] = torch.cat([_n2p(w[m + n + '/bias'], t=False).reshape(-1) for n in ('query', 'key', 'value')])
        sd[p + 'attn.proj.weight'] = _n2p(w[m + 'out/kernel']).flatten(1)
        sd[p + 'attn.proj.bias'] = _n2p(w[m + 'out/bias'])
        for r in range(2):
            sd[p + 'mlp.fc%d.weight' % (r + 1)] = _n2p(w[b + 'MlpBlock_3/Dense_%d/kernel' % r])
            sd[p + 'mlp.fc%d.bias' % (r + 1)] = _n2p(w[b + 'MlpBlock_3/Dense_%d/bias' % r])
        sd[p + 'norm2.weight'], sd[p + 'norm2.bias'] = _n2p(w[b + 'LayerNorm_2/scale']), _n2p(w[b + 'LayerNorm_2/bias'])
    return sd


def state_dict_to_vit_npz(sd, depth, num_heads):
    """inverse of :func:`vit_npz_to_state_dict` (used by the round-trip test and to export weights in the Flax layout)."""
    D = sd['norm.weight'].shape[0]
    hd = D // num_heads
    f = lambda t: t.detach().cpu().float().numpy()                       # noqa: E731
    w = {'embedding/kernel': f(sd['patch_embed.proj.weight']).transpose(2, 3, 1, 0), 'embedding/bias': f(sd['patch_embed.proj.bias']),
         'cls': f(sd['cls_token']), 'Transformer/posembed_input/pos_embedding': f(sd['pos_embed']),
         'Transformer/encoder_norm/scale': f(sd['norm.weight']), 'Transformer/encoder_norm/bias': f(sd['norm.bias']),
         'head/kernel': f(sd['head.weight']).T, 'head/bias': f(sd['head.bias'])}
    for i in range(depth):
        b = 'Transformer/encoderblock_%d/' % i
        m = b + 'MultiHeadDotProductAttention_1/'
        p = 'blocks.%d.' % i
        w[b + 'LayerNorm_0/scale'], w[b + 'LayerNorm_0/bias'] = f(sd[p + 'norm1.weight']), f(sd[p + 'norm1.bias'])
        w[b + 'LayerNorm_2/scale'], w[b + 'LayerNorm_2/bias'] = f(sd[p + 'norm2.weight']), f(sd[p + 'norm2.bias'])
        qkv_w, qkv_b = f(sd[p + 'attn.qkv.weight']), f(sd[p + 'attn.qkv.bias'])
        for j, n in enumerate(('query', 'key', 'value')):
            w[m + n + '/kernel'] = qkv_w[j * D:(j + 1) * D].T.reshape(D, num_heads, hd)
            w[m + n + '/bias'] = qkv_b[j * D:(j + 1) * D].reshape(num_heads, hd)
        w[m + 'out/kernel'] = f(sd[p + 'attn.proj.weight']).T.reshape(num_heads, hd, D)
        w[m + 'out/bias'] = f(sd[p + 'attn.proj.bias'])
        for r in range(2):
            w[b + 'MlpBlock_3/Dense_%d/kernel' % r] = f(sd[p + 'mlp.fc%d.weight' % (r + 1)]).T
            w[b + 'MlpBlock_3/Dense_%d/bias' % r] = f(sd[p + 'mlp.fc%d.bias' % (r + 1)])
    return w


def load_checkpoint(model, path, strict=False):
    """load a LOCAL ``.pth``/``.pt`` (timm/DeiT state_dict, optionally under 'model') or Flax ``.npz`` into ``model``; returns the
    ``load_state_dict`` result.  Shapes must match the model (no position-embedding resize)."""
    if path.endswith('.npz'):
        g = np.load(path, allow_pickle=False)
        sd = vit_npz_to_state_dict({k: g[k] for k in g.files}, model.depth)
    else:
        sd = torch.load(path, map_location='cpu', weights_only=True)
        if isinstance(sd, dict) and 'model' in sd and isinstance(sd['model'], dict):
            sd = sd['model']
    own = model.state_dict()
    bad = [k for k, v in sd.items() if k in own and tuple(own[k].shape) != tuple(v.shape)]
    if bad:
        raise ValueError('checkpoint/model shape mismatch for %s' % bad[:4])
    return model.load_state_dict(sd, strict=strict)


# file names of the checkpoints the reference's factories fetch (vit_fquant.py:822-828,849-855,876-882,904-907,929-932;
# swin_quant.py:838-844,866-872,894-900): torch.hub stores a download under <hub dir>/checkpoints/<basename of the URL>
PRETRAINED_FILES = {
    'deit_tiny_patch16_224': 'deit_tiny_patch16_224-a1311bcf.pth',
    'deit_small_patch16_224': 'deit_small_patch16_224-cd65a155.pth',
    'deit_base_patch16_224': 'deit_base_patch16_224-b5f2ef4d.pth',
    'vit_base_patch16_224': 'B_16-i21k-300ep-lr_0.001-aug_medium1-wd_0.1-do_0.0-sd_0.0--imagenet2012-steps_20k-lr_0.01-res_224.npz',
    'vit_large_patch16_224': 'L_16-i21k-300ep-lr_0.001-aug_medium1-wd_0.1-do_0.1-sd_0.1--imagenet2012-steps_20k-lr_0.01-res_224.npz',
    'swin_tiny_patch4_window7_224': 'swin_tiny_patch4_window7_224.pth',
    'swin_small_patch4_window7_224': 'swin_small_patch4_window7_224.pth',
    'swin_base_patch4_window7_224': 'swin_base_patch4_window7_224.pth',
}


def pretrained_path(factory_name):
    """where ``pretrained=True`` of the reference's factory of that name keeps its download: ``torch.hub.get_dir()/checkpoints/<file>``
    (``TORCH_HOME`` moves the hub directory)."""
    import os
    return os.path.join(torch.hub.get_dir(), 'checkpoints', PRETRAINED_FILES[factory_name])


def load_pretrained(model, factory_name):
    """``pretrained=True``: the reference downloads the checkpoint into the torch-hub cache and loads it with ``strict=False``
    (vit_fquant.py:822-828).  Here the SAME cache file is loaded when it is already there (nothing is fetched: no network), with
    loaders that execute nothing from the file; a missing file is a ``FileNotFoundError`` naming the path to put it at."""
    import os
    path = pretrained_path(factory_name)
    if not os.path.exists(path):
        raise FileNotFoundError('pretrained=True: %s is not in the torch-hub cache (the reference would download it; there is no network here). '
                                'Place the file at %s or load a local checkpoint with checkpoint.load_checkpoint(model, path).'
                                % (PRETRAINED_FILES[factory_name], path))
    _verify_hub_hash(path)
    res = load_checkpoint(model, path)
    missing, unexpected = list(getattr(res, 'missing_keys', ())), list(getattr(res, 'unexpected_keys', ()))
    if missing or unexpected:          # strict=False like the reference, but not silently: a wrong file shows up here
        import warnings
        warnings.warn('pretrained=True (%s): %d missing and %d unexpected keys, e.g. %s' % (
            os.path.basename(path), len(missing), len(unexpected), (missing + unexpected)[:4]))
    return res


def _verify_hub_hash(path):
    """torch.hub file names end in ``-<first hex digits of the sha256>.<ext>`` and ``load_state_dict_from_url(check_hash=True)`` - what
    the reference's DeiT factories call (vit_fquant.py:822-828) - refuses a file whose digest does not start with them.  The same test
    on the cached file, with hashlib (nothing is fetched).  Names without such a suffix (the Swin / Google .npz files) are not checked,
    as in torch.hub."""
    import hashlib
    import os
    import re
    m = re.search(r'-([a-f0-9]{8,})\.[A-Za-z0-9]+$', os.path.basename(path))
    if not m:
        return
    h = hashlib.sha256()
    with open(path, 'rb') as f:
        for chunk in iter(lambda: f.read(1 << 20), b''):
            h.update(chunk)
    if not h.hexdigest().startswith(m.group(1)):
        raise RuntimeError('invalid hash value (expected "%s", got "%s"): %s is truncated or not the file the reference downloads'
                           % (m.group(1), h.hexdigest()[:len(m.group(1))], path))

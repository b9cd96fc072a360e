"""Calibration-state dict <-> flat {str: array} (for .npz files).

Nested format (what ``FrozenPlan`` consumes): name -> tensor | {bit_name: tensor} | [entry per bit-pool
index], where an entry is a tensor or a {bit_name: tensor} dict -- the shape of the reference's
``quantizer.scale``, ``quantizer.dic_scale`` and ``best_scale/best_act_scale/best_weight_scale`` attributes
(models/ptq/quantizer/uniform.py:21-24, models/vit_fquant.py:234-239)."""
import torch


def flatten(calib):
    out = {}
    for k, v in calib.items():
        if isinstance(v, dict):
            for b, t in v.items():
                out['%s/%s' % (k, b)] = t
        elif isinstance(v, (list, tuple)):
            for i, t in enumerate(v):
                if isinstance(t, dict):
                    for b, u in t.items():
                        out['%s/%d/%s' % (k, i, b)] = u
                else:
                    out['%s/%d' % (k, i)] = t
        else:
            out[k] = v
    return out


def unflatten(flat):
    c = {}
    for k, v in flat.items():
        v = torch.as_tensor(v).float()
        parts = k.split('/')
        if len(parts) == 1:
            c[k] = v
            continue
        if not parts[1].isdigit():
            c.setdefault(parts[0], {})[parts[1]] = v
            continue
        lst = c.setdefault(parts[0], [])
        i = int(parts[1])
        while len(lst) <= i:
            lst.append(None)
        if len(parts) == 2:
            lst[i] = v
        else:
            if lst[i] is None:
                lst[i] = {}
            lst[i][parts[2]] = v
    return c


def load_npz(path, prefix='calib/'):
    import numpy as np
    g = np.load(path)
    return unflatten({k[len(prefix):]: g[k] for k in g.files if k.startswith(prefix)})


# ------------------------------------------------------------------------------------------------------------------------
# Frozen plan on disk: architecture + calibrated state + fp32 weights in ONE .npz written and read with
# numpy only (allow_pickle stays False).  ``load_plan`` rebuilds the integer plan on a GPU without the model object and
# without recalibrating; it works for ViT/DeiT (``FrozenPlan``) and Swin (``SwinPlan``).
# ------------------------------------------------------------------------------------------------------------------------
def save_plan(path, model):
    """model: a calibrated ``VisionTransformer`` or ``SwinTransformer`` (after ``model_close_calibrate``)."""
    import json
    import numpy as np
    kind = 'swin' if 'depths' in model.arch else 'vit'
    out = {'meta': np.array(json.dumps({'kind': kind, 'arch': model.arch, 'in_chans': model.in_chans, 'format': 1,
                                        'input_quant': bool(getattr(model, 'input_quant', True))}))}
    for k, v in flatten(model.export_calib()).items():
        out['calib/' + k] = v.detach().float().cpu().numpy()
    for k, v in model.state_dict().items():
        if v.dtype == torch.float32:
            out['weight/' + k] = v.detach().cpu().numpy()
    np.savez_compressed(path, **out)


def load_plan(path, device='cuda', bits=8):
    """-> ``FrozenPlan`` (ViT/DeiT) or ``SwinPlan`` built from a file written by ``save_plan``."""
    import json
    import numpy as np
    g = np.load(path, allow_pickle=False)
    meta = json.loads(str(g['meta']))
    if meta.get('format') != 1:
        raise ValueError('%s: unknown plan file format %r' % (path, meta.get('format')))
    arch = meta['arch']
    sd = {k[len('weight/'):]: torch.from_numpy(g[k]) for k in g.files if k.startswith('weight/')}
    calib = unflatten({k[len('calib/'):]: g[k] for k in g.files if k.startswith('calib/')})
    if meta['kind'] == 'swin':
        from .swin_plan import SwinPlan
        arch['depths'], arch['num_heads'] = tuple(arch['depths']), tuple(arch['num_heads'])
        calib = {k: (v.reshape(-1) if torch.is_tensor(v) else v) for k, v in calib.items()}
        return SwinPlan(arch, sd, calib, device=device, in_chans=meta['in_chans'], bits=bits)
    from .plan import FrozenPlan
    # input_quant False = the reference's vit_large factory (vit_fquant.py:925); files written before the key existed are input_quant models
    return FrozenPlan(arch, sd, calib, device=device, in_chans=meta['in_chans'], input_quant=bool(meta.get('input_quant', True)))

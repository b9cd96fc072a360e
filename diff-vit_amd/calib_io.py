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

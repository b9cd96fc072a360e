"""Generate tests/golden/swin_winattn.npz by RUNNING THE REAL REFERENCE ``WindowAttention`` (models/swin_quant.py:53-221)
standalone on CPU:  python oracle/gen_golden_swin.py

The reference's whole Swin model cannot run a forward as shipped (SURVEY.md 8c caveat 3), but its ``WindowAttention`` module
can: this script calibrates it (quant=False, calibrate/last_calibrate=True on every QAct/QLinear inside) and runs the quantized
forward with and without a shifted-window mask, tapping every QAct.  Only inputs, calibrated scales and outputs (data) are
stored; weights come from the build-owned generator.  Same process-local ``Tensor.cuda`` identity as gen_golden.py.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_golden as G                                    # noqa: E402  (reuses the loader helpers; no reference code)

synth, oracle, GOLD = G.synth, G.oracle, G.GOLD


def main():
    torch.manual_seed(0)
    ref_config, ref_models, _ = G.import_reference()
    from models import swin_quant
    dim, heads, ws, nW, Bimg = 64, 2, 7, 3, 2
    N = ws * ws
    cfg = ref_config.Config(True, True, 'minmax')
    m = swin_quant.WindowAttention(dim, window_size=(ws, ws), num_heads=heads, qkv_bias=True, quant=False, calibrate=False, cfg=cfg)
    sd = {
        'qkv.weight': synth.normal(21, 'wa/qkv.w', (3 * dim, dim), 0.09), 'qkv.bias': synth.normal(21, 'wa/qkv.b', (3 * dim,), 0.1),
        'proj.weight': synth.normal(21, 'wa/proj.w', (dim, dim), 0.08), 'proj.bias': synth.normal(21, 'wa/proj.b', (dim,), 0.05),
        'relative_position_bias_table': synth.normal(21, 'wa/table', ((2 * ws - 1) ** 2, heads), 0.6),
    }
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and set(missing) <= {'relative_position_index'}, (missing, unexpected)
    m.eval()
    s_in = 2.0 ** -4
    codes_cal = torch.clamp(torch.round(synth.normal(21, 'wa/xcal', (Bimg * nW, N, dim), 18.0)), -128, 127)
    codes_ev = torch.clamp(torch.round(synth.normal(21, 'wa/xev', (Bimg * nW, N, dim), 18.0)), -128, 127)
    # shifted-window style mask: tokens carry a region id; pairs from different regions get -100
    region = torch.zeros(nW, N, dtype=torch.long)
    region[1, 28:] = 1
    region[2, :] = (torch.arange(N) % 7 >= 4).long() + 2 * (torch.arange(N) >= 21).long()
    mask = (region.unsqueeze(1) != region.unsqueeze(2)).float() * -100.0
    q_mods = {n: mod for n, mod in m.named_modules() if isinstance(mod, (ref_models.QAct, ref_models.QLinear))}
    with torch.no_grad():
        for mod in q_mods.values():
            mod.calibrate, mod.last_calibrate = True, True
        m(codes_cal * s_in, mask=mask)
        for mod in q_mods.values():
            mod.calibrate, mod.last_calibrate, mod.quant = False, False, True
        m.log_int_softmax.quant = True
        out = {}
        for tag, msk in (('nomask', None), ('mask', mask)):
            taps, hooks = {}, []

            def mk(name, mod):
                def hook(_m, _inp, o):
                    if name == 'log_int_softmax':
                        k = torch.where(o > 0, -torch.log2(o.clamp(min=1e-30)), torch.full_like(o, 16.0))
                        taps['softmax_k'] = k.round().to(torch.int8).numpy()
                    else:
                        taps[name] = torch.round(o / mod.quantizer.scale.reshape(oracle.act_shape(o) if o.dim() > 1 else (-1,))).to(torch.int16).numpy()
                return hook
            for name, mod in m.named_modules():
                if isinstance(mod, ref_models.QAct) or name == 'log_int_softmax':
                    hooks.append(mod.register_forward_hook(mk(name, mod)))
            y = m(codes_ev * s_in, mask=msk)
            for h in hooks:
                h.remove()
            for k, v in taps.items():
                out['taps/%s/%s' % (tag, k)] = v
            out['out/%s' % tag] = y.numpy()
    for name, mod in q_mods.items():
        if isinstance(mod, ref_models.QAct):
            out['scale/' + name] = mod.quantizer.scale.detach().float().reshape(-1).numpy()
        else:
            for bt, s in mod.quantizer.dic_scale.items():
                out['wscale/%s/%s' % (name, bt)] = s.detach().float().reshape(-1).numpy()
            out['wbit/' + name] = np.array(mod.quantizer.bit_type.name)
    out['x_cal'] = codes_cal.to(torch.int8).numpy()
    out['x_ev'] = codes_ev.to(torch.int8).numpy()
    out['s_in'] = np.float32(s_in)
    out['mask'] = mask.numpy()
    out['rel_index'] = m.relative_position_index.numpy()
    out['seed'] = np.int64(21)
    os.makedirs(GOLD, exist_ok=True)
    np.savez_compressed(os.path.join(GOLD, 'swin_winattn.npz'), **out)
    print('swin_winattn: wrote %d arrays; weight bit types:' % len(out), {n: str(out['wbit/' + n]) for n in ('qkv', 'proj')})
    for k in sorted(out):
        if k.startswith('scale/'):
            print(' ', k, out[k][:4])


if __name__ == '__main__':
    main()

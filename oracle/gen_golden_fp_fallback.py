"""Generate tests/golden/micro_vit_fp_fallback.npz by RUNNING THE REAL REFERENCE (imported from /root/reference) on CPU: the micro-ViT
of micro_vit.npz after calibration + model_quant(), evaluated with bit_config lists that contain -1 -- the reference's per-layer fp32
fallback (models/ptq/layers.py:144,171; models/vit_fquant.py:199,429-430,462-463; models/layers_quant.py:222) -- and, afterwards,
once more with [8]*10: a -1 entry flips the block's QIntLayerNorm to F.layer_norm for good, so that forward no longer equals
logits/q8.  Only outputs are stored.   Run once in the build container:  python oracle/gen_golden_fp_fallback.py
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_golden as G          # noqa: E402

CASES = {'head': [8] * 9 + [-1], 'embed': [-1] + [8] * 9, 'proj0': [8, 8, -1] + [8] * 7, 'fc2_1': [8] * 8 + [-1, 8],
         'qkv1_fc1_0': [8, 8, 8, -1, 8, -1, 8, 8, 8, 8]}


def main():
    ref = G.import_reference()
    g = np.load(os.path.join(G.GOLD, 'micro_vit.npz'))
    arch = G.synth.ARCHS['micro']
    sd = {k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith('w/')}
    x_cal, x_ev = torch.from_numpy(g['x_cal']), torch.from_numpy(g['x_ev'])
    out = {}
    for name, bc in CASES.items():
        m = G.build_ref(arch, sd, ref)                      # a fresh model per case: the norm flips are permanent
        with torch.no_grad():
            m.model_open_calibrate()
            m.model_open_last_calibrate()
            m(x_cal, plot=False)
            m.model_close_calibrate()
            m.model_quant()
            o = m(x_ev, bc, False)[0]
            after = m(x_ev, [8] * 10, False)[0]
        out['bits/' + name] = np.array(bc, dtype=np.int8)
        out['logits/' + name] = o.numpy()
        out['after_q8/' + name] = after.numpy()
        out['norm_modes/' + name] = np.array([[b.norm1.mode == 'ln', b.norm2.mode == 'ln'] for b in m.blocks])
        print(name, 'finite', bool(torch.isfinite(o).all()), 'after == logits/q8:', bool(np.array_equal(after.numpy(), g['logits/q8'])),
              out['norm_modes/' + name].tolist())
    np.savez_compressed(os.path.join(G.GOLD, 'micro_vit_fp_fallback.npz'), **out)


if __name__ == '__main__':
    main()

"""Generate tests/golden/micro_vit_dequant.npz by RUNNING THE REAL REFERENCE (imported from /root/reference) on CPU: the micro-ViT
of micro_vit.npz (same weights, calibration batch and evaluation batch, read from that fixture) after

    model_open_calibrate(); model_open_last_calibrate(); model(x_cal); model_close_calibrate(); model_quant()

in the two states in which the reference's forward leaves the all-quantized graph (models/vit_fquant.py:667-683):
  * ``dequant``:  model_dequant()  -- per-module .quant False everywhere; QIntLayerNorm stays in mode 'int' and the softmax stays
    log-int (neither is touched by model_dequant), so this is NOT the float model;
  * ``fc2_float``: model_quant() again, then blocks[0].mlp.fc2.quant = False.
Only outputs are stored.   Run once in the build container:  python oracle/gen_golden_dequant.py
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_golden as G          # noqa: E402  (import_reference / build_ref / synth)


def main():
    ref = G.import_reference()
    g = np.load(os.path.join(G.GOLD, 'micro_vit.npz'))
    arch = G.synth.ARCHS['micro']
    sd = {k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith('w/')}
    m = G.build_ref(arch, sd, ref)
    x_cal, x_ev = torch.from_numpy(g['x_cal']), torch.from_numpy(g['x_ev'])
    out = {}
    with torch.no_grad():
        m.model_open_calibrate()
        m.model_open_last_calibrate()
        m(x_cal, plot=False)
        m.model_close_calibrate()
        m.model_quant()
        q8 = m(x_ev, [8] * 10, False)[0]
        assert np.array_equal(q8.numpy(), g['logits/q8'])          # same state as the main fixture
        m.model_dequant()
        out['dequant/q8'] = m(x_ev, [8] * 10, False)[0].numpy()
        out['dequant/q4'] = m(x_ev, [4] * 10, False)[0].numpy()
        m.model_quant()
        m.blocks[0].mlp.fc2.quant = False
        out['fc2_float/q8'] = m(x_ev, [8] * 10, False)[0].numpy()
    np.savez_compressed(os.path.join(G.GOLD, 'micro_vit_dequant.npz'), **out)
    print('micro_vit_dequant: wrote', {k: v.shape for k, v in out.items()})


if __name__ == '__main__':
    main()

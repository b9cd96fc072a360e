import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import diff_vit_amd as dva
from diff_vit_amd import calib_io
arch = dva.synth.ARCHS['deit_small']
ref = calib_io.flatten(calib_io.load_npz(os.path.join(ROOT, 'tests/golden/deit_small.npz')))
for dev in ('cpu', 'cuda'):
    m = dva.deit_small_patch16_224(cfg=dva.Config()); m.load_state_dict(dva.synth.vit_state_dict(arch, 3), strict=False); m = m.to(dev).eval()
    dva.harness.calibrate_model(m, dva.synth.images(3, 2, 224).to(dev))
    mine = calib_io.flatten(m.export_calib())
    bad = [(k, int((mine[k].reshape(-1) != ref[k].reshape(-1)).sum()), ref[k].numel()) for k in ref if not torch.equal(mine[k].reshape(-1), ref[k].reshape(-1))]
    print(dev, 'mismatching tensors', len(bad), 'of', len(ref))
    for k, n, tot in bad[:12]:
        a, b = mine[k].reshape(-1), ref[k].reshape(-1)
        i = int((a != b).nonzero()[0])
        print('   %-40s %d/%d  first idx %d mine %.6g ref %.6g ratio %.6f' % (k, n, tot, i, a[i], b[i], a[i] / b[i]))

"""GPU box: the product's calibration of DeiT-S (harness.calibrate_model, where='host' and where='model') against the REAL reference's calibration
state in tests/golden/deit_small.npz: tensors equal, power-of-two exponents that moved, PTF factors (profiles/r02_calibration_agreement.txt)."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import diff_vit_amd as dva
g=np.load(os.path.join(ROOT, 'tests', 'golden', 'deit_small.npz'))
arch=dva.synth.ARCHS['deit_small']; seed=int(g['seed'])
ref={k[len('calib/'):]:g[k] for k in g.files if k.startswith('calib/')}
x=dva.synth.images(seed,int(g['n_calib']),224)
for where in ('host','model'):
    m=dva.deit_small_patch16_224(cfg=dva.Config(True,True,'minmax')); m.load_state_dict(dva.synth.vit_state_dict(arch,seed),strict=False); m=m.cuda().eval()
    out=dva.harness.calibrate_model(m,x.cuda(),where=where)[0]
    flat=dva.calib_io.flatten(m.export_calib())
    flips=0; pot_el=0; worst=0; fac=0; eq=0
    first=None
    for k,want in ref.items():
        a=flat[k].numpy().reshape(want.shape)
        eq+=int(np.array_equal(a,want))
        if np.all(np.frexp(want)[0]==0.5):
            n=int((a!=want).sum()); flips+=n; pot_el+=want.size
            if n and first is None: first=(k,n)
        else:
            worst=max(worst,float(np.abs(a/want-1).max())); fac+=int((np.round(a/a.min())!=np.round(want/want.min())).sum())
    print(where,'tensors equal',eq,'/',len(ref),'pot flips',flips,'of',pot_el,'first',first,'ptf worst rel',worst,'ptf factor diffs',fac, 'calib logits max|d|', float(np.abs(out.cpu().numpy()-g['calib_logits']).max()))

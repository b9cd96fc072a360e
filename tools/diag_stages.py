"""GPU diagnostic: first stage at which the HIP engine and the oracle disagree (DeiT-S fixture)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import p2vit_oracle as O
import diff_vit_amd as dva
g = np.load(os.path.join(ROOT, 'tests/golden/deit_small.npz'))
arch = dva.synth.ARCHS['deit_small']; seed = int(g['seed'])
sd = dva.synth.vit_state_dict(arch, seed)
calib = O.unflatten_calib({k[6:]: torch.from_numpy(g[k]) for k in g.files if k.startswith('calib/')})
plan = dva.FrozenPlan(arch, sd, calib)
x = dva.synth.images(seed, 4, 224, offset=1000)
orc = O.OracleViT(arch, sd); orc.calib = calib
bits = [8] * 50
taps = {}
ref = orc.quant_forward(x, bits, taps)
B, T, D = 4, 197, 384
stages = [(3, 'x', D, 'qact1')]
for i in range(12):
    p = 'blocks.%d.' % i; b = 3 + 7 * i
    stages += [(b + 1, 'ln', D, p + 'attn.qact0'), (b + 2, 'qkv', 3 * D, p + 'attn.qact1'), (b + 3, 'att', D, p + 'attn.qact2'),
               (b + 4, 'x', D, p + 'qact2'), (b + 5, 'ln', D, p + 'mlp.qact0'), (b + 6, 'hid', 4 * D, p + 'mlp.qact1'),
               (b + 7, 'x', D, p + 'qact4')]
nbad = 0
for stop, buf, cols, name in stages:
    plan.forward(x.cuda(), bits, stop_after=stop)
    torch.cuda.synchronize()
    got = plan.view(B, buf, B * T, cols).cpu().numpy().astype(np.int64)
    want = taps[name].reshape(B * T, cols).numpy().astype(np.int64)
    bad = np.argwhere(got != want)
    if len(bad):
        print(name, len(bad), 'of', got.size, 'first', bad[:5].tolist(), 'got', [int(got[tuple(b)]) for b in bad[:5]], 'want', [int(want[tuple(b)]) for b in bad[:5]])
        nbad += 1
        if nbad >= 3: break
    else:
        print(name, 'ok')

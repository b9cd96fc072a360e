"""host-side: oracle throughput (DeiT-S int8, 16-image forwards) against the torch thread count on this machine."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import diff_vit_amd as dva
import p2vit_oracle as O
from conftest import load_golden, golden_calib
g = load_golden('deit_small')
arch = dva.synth.ARCHS['deit_small']
o = O.OracleViT(arch, dva.synth.vit_state_dict(arch, int(g['seed'])))
o.calib = golden_calib(g, O)
bits = [8] * 50
for nb in (16, 64):
    x = dva.synth.images(1000, nb, 224)
    for th in (8, 16, 32, 64, 128):
        if th > (os.cpu_count() or 1): continue
        torch.set_num_threads(th)
        with torch.no_grad():
            o.quant_forward(x, bits)
            t = time.perf_counter(); n = 0
            while time.perf_counter() - t < 6: o.quant_forward(x, bits); n += 1
            dt = time.perf_counter() - t
        print('batch %3d threads %3d: %.2f img/s' % (nb, th, nb * n / dt), flush=True)

"""GPU box: wall time of the one-off host calibration (harness.calibrate_model where='host', DeiT-S, 2 images) at different torch thread counts,
and whether the calibrated state is identical across them.  usage: python tools/calib_threads.py"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import diff_vit_amd as dva
arch = dva.synth.ARCHS['deit_small']
sd = dva.synth.vit_state_dict(arch, 3)
x = dva.synth.images(3, 2, 224)
ref = None
print('host threads available:', os.cpu_count(), ' torch default:', torch.get_num_threads())
for th in (torch.get_num_threads(), 64, 32, 16, 8):
    torch.set_num_threads(th)
    m = dva.deit_small_patch16_224(cfg=dva.Config(True, True, 'minmax'))
    m.load_state_dict(sd, strict=False)
    m = m.cuda().eval()
    t = time.perf_counter()
    dva.harness.calibrate_model(m, x.cuda(), where='host')
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    flat = dva.calib_io.flatten(m.export_calib())
    same = ref is None or all(torch.equal(flat[k], ref[k]) for k in ref)
    ref = ref or flat
    print('threads %3d: %.2f s   identical to the first run: %s' % (th, dt, same), flush=True)

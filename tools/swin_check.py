"""GPU: Swin-B (BASELINE config 4 shape) through the drop-in surface - parity vs OracleSwin on a few images, then throughput."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import diff_vit_amd as dva
from diff_vit_amd import swin
import swin_oracle as SO
name = sys.argv[1] if len(sys.argv) > 1 else 'swin_base'
n_par = int(sys.argv[2]) if len(sys.argv) > 2 else 2
B = int(sys.argv[3]) if len(sys.argv) > 3 else 64
S = dva.synth
cfg = dva.Config(True, True, 'minmax')
make = {'swin_base': swin.swin_base_patch4_window7_224, 'swin_micro': swin.swin_micro_patch4_window7_56}[name]
m = make(cfg=cfg).eval()
img = m.arch['img_size']
m.load_state_dict(S.swin_state_dict(m.state_dict(), 5))
x = S.images(5, max(n_par, 2), img)
m.cuda()
t0 = time.time()
with torch.no_grad():
    m.model_open_calibrate(); m.model_open_last_calibrate(); m(x[:2].cuda()); m.model_close_calibrate()
    m.model_quant()
    torch.cuda.synchronize(); print('calibration %.1f s' % (time.time() - t0), flush=True)
    out = m(x[:n_par].cuda())
    torch.cuda.synchronize()
    t0 = time.time()
    ref = SO.OracleSwin(m.arch, {k: v.cpu() for k, v in m.state_dict().items()}).quant_forward(x[:n_par], m.export_calib(), 8)
    print('oracle %.1f s for %d images' % (time.time() - t0, n_par), flush=True)
    eq = torch.equal(out.cpu(), ref)
    print('HIP == oracle:', eq, 'mismatching logits', int((out.cpu() != ref).sum()), 'top1', ref.argmax(1).tolist(), out.argmax(1).tolist())
    xb = S.images(6, 8, img).repeat((B + 7) // 8, 1, 1, 1)[:B].cuda()
    for _ in range(2): m(xb)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(5): m(xb)
    torch.cuda.synchronize(); dt = (time.time() - t0) / 5
    print('%s B=%d: %.2f ms/step, %.0f img/s' % (name, B, dt * 1e3, B / dt))
sys.exit(0 if eq else 1)

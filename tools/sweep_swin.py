"""GPU: Swin-B / Swin-T forward rate per batch size and number of slices (side streams only; SwinPlan.forward(n_streams=..., slices=...)).
python tools/sweep_swin.py [swin_base|swin_tiny] BATCH[,BATCH...]"""
import contextlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import diff_vit_amd as dva

name = sys.argv[1] if len(sys.argv) > 1 else 'swin_base'
batches = [int(b) for b in (sys.argv[2] if len(sys.argv) > 2 else '16,32,64,128,256').split(',')]
with contextlib.redirect_stdout(sys.stderr):
    model = dva.harness.str2model(name)(cfg=dva.Config(True, True, 'minmax'))
model.load_state_dict(dva.synth.swin_state_dict(model.state_dict(), 1))
model = model.cuda().eval()
with torch.no_grad():
    dva.harness.calibrate_model(model, dva.synth.images(1, 2, 224).cuda())
plan = model.freeze('cuda')
base = dva.synth.images(5, 32, 224, offset=100).cuda()


def rate(x, k, slices):
    run = lambda: plan.forward(x, n_streams=max(k, 1), slices=slices)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    n = max(5, min(20, 2048 // x.shape[0]))
    t0 = time.perf_counter()
    for _ in range(n):
        run()
    torch.cuda.synchronize()
    return x.shape[0] * n / (time.perf_counter() - t0)


for B in batches:
    x = base.repeat((B + 31) // 32, 1, 1, 1)[:B].contiguous()
    res = {}
    for rep in range(2):
        for k in (1, 2, 3, 4):
            if k > 1 and B < 8 * k:
                continue
            if k == 1:
                sl = [B]
            elif k == 4:                                   # three side streams + a smaller slice on the caller's stream
                side = (B * 1000 + 3764) // 3765
                sl = [side] * 3 + [B - 3 * side]
                if sl[-1] < 1:
                    continue
            else:
                q, r = divmod(B, k)
                sl = [q + (1 if i < r else 0) for i in range(k)]
            res.setdefault(k, []).append(rate(x, min(k, 3), sl if k > 1 else None) if k > 1 else rate(x, 1, None))
    print('%s batch %4d  ' % (name, B) + '  '.join('%d: %s' % (k, ' / '.join('%.0f' % v for v in res[k])) for k in sorted(res)), flush=True)

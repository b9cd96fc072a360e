"""GPU: time of one DeiT-S / 256 step when the host reads the logits after every forward (synchronous use, e.g. harness.validate) and of
back-to-back steps, with the side streams' slices enqueued by worker threads (engine.THREADED_ENQUEUE, the default) or one after the other."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import diff_vit_amd as dva
E = dva.engine
m = dva.deit_small_patch16_224(pretrained=False, cfg=dva.Config()).cuda().eval()
m.load_state_dict(dva.synth.vit_state_dict(dva.synth.ARCHS['deit_small'], 5), strict=False)
dva.harness.calibrate_model(m, dva.synth.images(5, 2, 224).cuda())
plan = m.freeze()
bc = [8] * 50
X = dva.synth.images(5, 64, 224, offset=100).cuda().repeat(4, 1, 1, 1).contiguous()
OUT = torch.empty(256, 1000, device='cuda')


def per_step(fn, sync, n=60):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
        if sync:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


step = lambda: plan.forward_streams(X, bc, OUT, 3)
for rep in range(2):
    for threaded in (True, False):
        E.THREADED_ENQUEUE = threaded
        print('%-22s synchronous step %.3f ms   back-to-back %.3f ms' % ('worker threads' if threaded else 'one after the other', per_step(step, True), per_step(step, False)), flush=True)
E.THREADED_ENQUEUE = True
print('%-22s synchronous step %.3f ms   back-to-back %.3f ms' % ('one slice', per_step(lambda: plan.forward(X, bc, out=OUT), True), per_step(lambda: plan.forward(X, bc, out=OUT), False)))

"""GPU: is the batch-256 step bound by the host's enqueue rate?  Times the enqueue loop (no synchronisation) against the GPU time."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import diff_vit_amd as dva
arch = dva.synth.ARCHS['deit_small']
m = dva.deit_small_patch16_224(cfg=dva.Config()); m.load_state_dict(dva.synth.vit_state_dict(arch, 3), strict=False); m = m.cuda().eval()
dva.harness.calibrate_model(m, dva.synth.images(3, 2, 224).cuda())
plan = m.freeze('cuda')
bits = [8] * 50
x = dva.synth.images(1000, 32, 224).cuda().repeat(8, 1, 1, 1)
out = torch.empty(256, 1000, device='cuda')
for ns in (3, 1):
    for _ in range(10): plan.forward_streams(x, bits, out, ns)
    torch.cuda.synchronize()
    n = 100
    t0 = time.perf_counter()
    for _ in range(n): plan.forward_streams(x, bits, out, ns)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print('slices %d: host enqueue %.3f ms per step (with back-pressure), GPU-inclusive %.3f ms per step' % (ns, t_host / n * 1e3, t_all / n * 1e3), flush=True)
    one = []
    for _ in range(20):           # one step into an EMPTY queue: the pure host cost of enqueuing a step
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        plan.forward_streams(x, bits, out, ns)
        one.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    print('           host cost of one step into an empty queue: median %.3f ms' % (sorted(one)[len(one) // 2] * 1e3), flush=True)

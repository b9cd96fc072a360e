"""GPU: the batch-256 DeiT-S forward on three stream slices, eager enqueue vs HIP-graph replay of the same fork/join."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import diff_vit_amd as dva
arch = dva.synth.ARCHS['deit_small']
m = dva.deit_small_patch16_224(cfg=dva.Config()); m.load_state_dict(dva.synth.vit_state_dict(arch, 3), strict=False); m = m.cuda().eval()
dva.harness.calibrate_model(m, dva.synth.images(3, 2, 224).cuda())
plan = m.freeze('cuda')
bits = [8] * 50
B = 256
x = dva.synth.images(1000, 32, 224).cuda().repeat(8, 1, 1, 1)
out = torch.empty(B, 1000, device='cuda')
def run(fn, n=100):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n
for ns in (3, 2):
    eager = run(lambda: plan.forward_streams(x, bits, out, ns))
    ref = out.clone()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        plan.forward_streams(x, bits, out, ns)
    out.zero_()
    graph = run(g.replay)
    print('streams %d: eager %.3f ms (%.0f img/s)   graph %.3f ms (%.0f img/s)   equal=%s' % (ns, eager * 1e3, B / eager, graph * 1e3, B / graph, torch.equal(out, ref)), flush=True)

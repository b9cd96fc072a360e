"""GPU: latency of the DeiT-S forward at small batch, eager enqueue vs HIP-graph replay of the same p2v_forward call."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import diff_vit_amd as dva
arch = dva.synth.ARCHS['deit_small']
m = dva.deit_small_patch16_224(cfg=dva.Config()); m.load_state_dict(dva.synth.vit_state_dict(arch, 3), strict=False); m = m.cuda().eval()
dva.harness.calibrate_model(m, dva.synth.images(3, 2, 224).cuda())
plan = m.freeze('cuda')
bits = [8] * 50
for B in (1, 4, 8, 32, 128):
    x = dva.synth.images(1000, B, 224).cuda()
    out = torch.empty(B, 1000, device='cuda')
    for _ in range(5): plan.forward(x, bits, out=out)
    torch.cuda.synchronize(); t = time.perf_counter()
    n = 200
    for _ in range(n): plan.forward(x, bits, out=out)
    torch.cuda.synchronize(); eager = (time.perf_counter() - t) / n
    ref = out.clone()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        plan.forward(x, bits, out=out)
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    with torch.cuda.graph(g):
        plan.forward(x, bits, out=out)
    out.zero_()
    for _ in range(5): g.replay()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): g.replay()
    torch.cuda.synchronize(); graph = (time.perf_counter() - t) / n
    print('B=%3d  eager %.3f ms (%.0f img/s)   graph %.3f ms (%.0f img/s)   equal=%s' % (B, eager * 1e3, B / eager, graph * 1e3, B / graph, torch.equal(out, ref)))

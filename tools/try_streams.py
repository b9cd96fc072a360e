import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import diff_vit_amd as dva
arch = dva.synth.ARCHS['deit_small']
m = dva.deit_small_patch16_224(cfg=dva.Config()); m.load_state_dict(dva.synth.vit_state_dict(arch, 3), strict=False); m = m.cuda().eval()
dva.harness.calibrate_model(m, dva.synth.images(3, 2, 224).cuda())
plan = m.freeze('cuda')
B = 256
x = dva.synth.images(1000, 32, 224).repeat(8, 1, 1, 1).cuda()
bits = [8] * 50
ref = plan.forward(x, bits).clone()
out = torch.empty_like(ref)
for ns in (1, 2, 3, 4):
    for _ in range(5): plan.forward_streams(x, bits, out, ns)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): plan.forward_streams(x, bits, out, ns)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 20
    print('streams', ns, '%.3f ms  %.0f img/s  equal=%s' % (dt * 1e3, B / dt, torch.equal(out, ref)))
print('--- HIP graph replay ---')
for ns in (1, 2, 3, 4):
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        plan.forward_streams(x, bits, out, ns)      # warm-up on the side stream (allocates workspaces)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    try:
        with torch.cuda.graph(g):
            plan.forward_streams(x, bits, out, ns)
    except Exception as e:
        print('capture failed for', ns, repr(e)[:200]); continue
    out.zero_()
    for _ in range(3): g.replay()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): g.replay()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 20
    print('graph streams', ns, '%.3f ms  %.0f img/s  equal=%s' % (dt * 1e3, B / dt, torch.equal(out, ref)))
# host enqueue cost
t = time.perf_counter()
for _ in range(20): plan.forward(x, bits, out=out)
te = (time.perf_counter() - t) / 20
torch.cuda.synchronize()
print('host enqueue time per forward (1 stream): %.3f ms' % (te * 1e3))

"""GPU: the default bench step (DeiT-S, batch 256, FrozenPlan.forward_streams) captured ONCE into a HIP graph and replayed N times, so
that a rocprofv3 --kernel-trace of this process is GPU-bound (an eager run under the profiler is host-bound: every launch costs the
host ~15 us there).  usage: python3 tools/overlap_run.py [replays=30] [slices=default|a,b,c] [model=deit_small] [batch=256]
Prints ms per replay (unprofiled this equals bench.py's ms_per_step)."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import diff_vit_amd as dva
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
sl = [int(v) for v in sys.argv[2].split(',')] if len(sys.argv) > 2 and sys.argv[2] != 'default' else None
name = sys.argv[3] if len(sys.argv) > 3 else 'deit_small'
B = int(sys.argv[4]) if len(sys.argv) > 4 else 256
arch = dva.synth.ARCHS[name]
m = dva.harness.str2model(name)(cfg=dva.Config(True, True, 'minmax'))
m.load_state_dict(dva.synth.vit_state_dict(arch, 3), strict=False)
m = m.cuda().eval()
dva.harness.calibrate_model(m, dva.synth.images(3, 2, 224).cuda(), where='host')
plan = m.freeze('cuda')
bits = [8] * 50
x = dva.synth.images(1000, B, 224).cuda()
out = torch.empty(B, 1000, device='cuda')
for _ in range(3):
    plan.forward_streams(x, bits, out, 3, sl)
torch.cuda.synchronize()
ref = out.clone()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    plan.forward_streams(x, bits, out, 3, sl)
out.zero_()
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(n):
    g.replay()
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / n
print('slices %s: graph replay %.3f ms per step (%.0f img/s), equal to eager: %s' % (sl or plan.slice_sizes(B, 3), dt * 1e3, B / dt, torch.equal(out, ref)), flush=True)

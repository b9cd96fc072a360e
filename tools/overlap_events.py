"""GPU: per-kernel durations UNDER OVERLAP (HIP events on every slice's stream while all slices run, FrozenPlan.profile_streams) next to
the isolated per-launch times (FrozenPlan.profile) of the same slice size.  usage: python tools/overlap_events.py [slices=default|a,b,c]"""
import os, sys, collections, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import diff_vit_amd as dva
sl = [int(v) for v in sys.argv[1].split(',')] if len(sys.argv) > 1 and sys.argv[1] != 'default' else None
arch = dva.synth.ARCHS['deit_small']
m = dva.deit_small_patch16_224(cfg=dva.Config(True, True, 'minmax'))
m.load_state_dict(dva.synth.vit_state_dict(arch, 3), strict=False)
m = m.cuda().eval()
dva.harness.calibrate_model(m, dva.synth.images(3, 2, 224).cuda(), where='host')
plan = m.freeze('cuda')
bits, B = [8] * 50, 256
x = dva.synth.images(1000, B, 224).cuda()
sizes = sl or plan.slice_sizes(B, 3)
iso = collections.defaultdict(list)
for _ in range(3):
    for k, ms in plan.profile(x[:max(sizes)], bits):
        iso[k].append(ms)
ovl = collections.defaultdict(list)
walls = []
for _ in range(5):
    per, wall = plan.profile_streams(x, bits, 3, sizes, rounds=3)
    walls.append(wall)
    for p in per[:3]:
        for k, ms in p:
            ovl[k].append(ms)
print('slices', sizes, ' longest stream of the middle step: %.3f ms (median of 5)' % sorted(walls)[2])
print('%-14s %6s %12s %12s %8s' % ('kind', 'n/slice', 'isolated us', 'overlap us', 'ratio'))
tot_i = tot_o = 0.0
for k in sorted(ovl, key=lambda k: -sum(ovl[k])):
    a, b = sum(iso[k]) / len(iso[k]) * 1e3, sum(ovl[k]) / len(ovl[k]) * 1e3
    n = len(iso[k]) // 3
    tot_i += a * n; tot_o += b * n
    print('%-14s %6d %12.2f %12.2f %8.2f' % (k, n, a, b, b / a))
print('per slice: sum isolated %.3f ms, sum under overlap %.3f ms' % (tot_i / 1e3, tot_o / 1e3))

"""GPU: the sliced forward when OTHER HIP streams of the process carried work before the plan took its side streams (DeiT-S, batch 256).
python tools/stream_pool_check.py N_FOREIGN [probe|plain]  ->  one JSON line: img/s at four / three / two / one slices and what the probe of
engine.side_streams saw.  `plain` = the behaviour before round 4's probed pool (the next three streams torch hands out): queue i and queue
i + 4 share a dispatch pipe, and two busy queues on one pipe run BELOW the one-stream rate (profiles/r04_stream_pool.txt)."""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import diff_vit_amd as dva

n_foreign = int(sys.argv[1]) if len(sys.argv) > 1 else 2
plain = len(sys.argv) > 2 and sys.argv[2] == 'plain'
m = dva.deit_small_patch16_224(pretrained=False, cfg=dva.Config()).cuda().eval()
m.load_state_dict(dva.synth.vit_state_dict(dva.synth.ARCHS['deit_small'], 5), strict=False)
dva.harness.calibrate_model(m, dva.synth.images(5, 2, 224).cuda())
plan = m.freeze()
bc = [8] * 50
X = dva.synth.images(5, 64, 224, offset=100).cuda().repeat(4, 1, 1, 1).contiguous()
OUT = torch.empty(256, 1000, device='cuda')
foreign = []
for _ in range(n_foreign):                       # e.g. a copy stream, a collective's stream, another library's stream
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for _ in range(20):
            torch.zeros(1 << 20, device='cuda').add_(1.0)
    foreign.append(st)
torch.cuda.synchronize()
if plain:
    dva.engine._SIDE_STREAMS[torch.cuda.current_device()] = [torch.cuda.Stream() for _ in range(3)]


def rate(slices, n_streams, n=30):
    for _ in range(3):
        plan.forward_streams(X, bc, OUT, n_streams, slices)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        plan.forward_streams(X, bc, OUT, n_streams, slices)
    torch.cuda.synchronize()
    return round(256 * n / (time.perf_counter() - t0))


print(json.dumps({'foreign_streams': n_foreign, 'side_streams': 'plain' if plain else 'probed', 'four_slices': rate([68, 68, 68, 52], 3),
                  'three_slices': rate([92, 92, 72], 2), 'two_slices': rate([128, 128], 1), 'one_slice': rate([256], 1),
                  'probe': dva.engine.SIDE_STREAM_REPORT.get(torch.cuda.current_device())}), flush=True)

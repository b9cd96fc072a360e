"""GPU: per-op durations of one Swin-B forward (p2v_run_ops_profile), aggregated per kind / stage."""
import os, sys, collections, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import diff_vit_amd as dva
from diff_vit_amd import swin
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
S = dva.synth
m = swin.swin_base_patch4_window7_224(cfg=dva.Config(True, True, 'minmax')).eval()
m.load_state_dict(S.swin_state_dict(m.state_dict(), 5))
m.cuda()
x = S.images(5, 2, 224).cuda()
with torch.no_grad():
    m.model_open_calibrate(); m.model_open_last_calibrate(); m(x); m.model_close_calibrate(); m.model_quant()
    plan = m.freeze('cuda')
    xb = S.images(6, 8, 224).repeat(B // 8, 1, 1, 1).cuda()
    plan.forward(xb, n_streams=1)
    prof = plan.profile(xb); prof = plan.profile(xb)
epi = {0: 'requant', 1: 'gelu', 2: 'resid', 4: 'head'}
agg = collections.OrderedDict()
r = plan._recorded[(B, 0, True)] if (B, 0, True) in plan._recorded else plan._recorded[(B, 0)]
for i, (kind, e, ms) in enumerate(prof):
    o = r['ops'][i]
    key = kind + ('/' + epi.get(e, str(e)) + ' K=%d N=%d' % (o.K, o.N) if kind == 'gemm' else (' C=%d' % o.N if kind == 'layernorm' else (' T=%d H=%d' % (o.i1, o.i2) if kind == 'window_attention' else '')))
    a = agg.setdefault(key, [0, 0.0]); a[0] += 1; a[1] += ms
tot = sum(v[1] for v in agg.values())
for k, (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print('%-42s x%3d  %7.3f ms  %5.1f %%' % (k, n, ms, 100 * ms / tot))
print('total %.3f ms for %d images (single stream)' % (tot, B))

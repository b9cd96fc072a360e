"""GPU: throughput of the DeiT-S architecture at other image sizes (token counts), with the per-kernel split of one single-stream forward.
python tools/bench_tokens.py [batch]  ->  224 (197 tokens), 384 (577: the resident attention kernel's range), 448 (785) and 512 (1025: the streaming kernel)."""
import os, sys, time
from functools import partial
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import diff_vit_amd as dva

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 64
sizes = [int(s) for s in sys.argv[2].split(',')] if len(sys.argv) > 2 else [224, 384, 448, 512]
for img in sizes:
    arch = dict(img_size=img, patch_size=16, embed_dim=384, depth=12, num_heads=6, num_classes=1000, mlp_ratio=4.0)
    sd = dva.synth.vit_state_dict(arch, 5)
    m = dva.VisionTransformer(img_size=img, patch_size=16, embed_dim=384, depth=12, num_heads=6, num_classes=1000, mlp_ratio=4.0, qkv_bias=True,
                              norm_layer=partial(dva.QIntLayerNorm, eps=1e-6), input_quant=True, cfg=dva.Config())
    m.load_state_dict(sd, strict=False)
    m = m.cuda().eval()
    dva.harness.calibrate_model(m, dva.synth.images(5, 2, img).cuda())
    plan = m.freeze()
    x = dva.synth.images(5, batch, img, offset=100).cuda()
    bc = [8] * 50
    out = torch.empty(batch, 1000, device='cuda')
    for mode in ('one stream', 'sliced'):
        run = (lambda: plan.forward(x, bc)) if mode == 'one stream' else (lambda: plan.forward_streams(x, bc, out, 3))
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 10
        for _ in range(n):
            run()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print('img %d (%d tokens) batch %d %-10s %8.1f img/s  %7.3f ms' % (img, (img // 16) ** 2 + 1, batch, mode, batch / dt, dt * 1e3), flush=True)
    prof = plan.profile(x, bc)
    agg = {}
    for kind, ms in prof:
        agg[kind] = agg.get(kind, 0.0) + ms
    tot = sum(agg.values())
    print('   ' + '  '.join('%s %.2f ms (%.0f %%)' % (k, v, 100 * v / tot) for k, v in sorted(agg.items(), key=lambda kv: -kv[1])[:6]), flush=True)
    del plan, m
    torch.cuda.empty_cache()

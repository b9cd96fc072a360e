"""GPU: which batch slicing (1 .. 4 kernels in flight: k - 1 side streams + the caller's stream) is fastest for a ViT geometry and batch size.
python tools/sweep_slices.py EMBED_DIM HEADS IMG BATCH[,BATCH...]  ->  one line per batch: img/s at 1, 2, 3, 4 slices (ABAB, two passes)."""
import os, sys, time
from functools import partial
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import diff_vit_amd as dva

dim, heads, img = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
batches = [int(b) for b in sys.argv[4].split(',')]
arch = dict(img_size=img, patch_size=16, embed_dim=dim, depth=12, num_heads=heads, num_classes=1000, mlp_ratio=4.0)
sd = dva.synth.vit_state_dict(arch, 5)
m = dva.VisionTransformer(img_size=img, patch_size=16, embed_dim=dim, depth=12, num_heads=heads, num_classes=1000, mlp_ratio=4.0, qkv_bias=True,
                          norm_layer=partial(dva.QIntLayerNorm, eps=1e-6), input_quant=True, cfg=dva.Config())
m.load_state_dict(sd, strict=False)
m = m.cuda().eval()
dva.harness.calibrate_model(m, dva.synth.images(5, 2, img).cuda())
plan = m.freeze()
bc = [8] * 50
tokens = (img // 16) ** 2 + 1


def split(B, k):
    if k == 1:
        return [B]
    d = 1000 * (k - 1) + 765                      # the caller's slice ~0.77 of a side slice (FrozenPlan.slice_sizes)
    side = (B * 1000 + d - 1) // d
    last = B - (k - 1) * side
    if last < 1:
        q, r = divmod(B, k)
        return [q + (1 if i < r else 0) for i in range(k)]
    return [side] * (k - 1) + [last]


for B in batches:
    x = dva.synth.images(5, min(B, 64), img, offset=100).cuda()
    x = x.repeat((B + x.shape[0] - 1) // x.shape[0], 1, 1, 1)[:B].contiguous()
    out = torch.empty(B, 1000, device='cuda')
    res = {}
    for rep in range(2):
        for k in (1, 2, 3, 4):
            if B < 4 * k and k > 1:
                continue
            sl = split(B, k)
            run = lambda: plan.forward_streams(x, bc, out, max(k - 1, 1), slices=sl)
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            n = max(5, min(30, int(2e6 / (B * tokens))))
            t0 = time.perf_counter()
            for _ in range(n):
                run()
            torch.cuda.synchronize()
            res.setdefault(k, []).append(B * n / (time.perf_counter() - t0))
    best = max(res, key=lambda k: sum(res[k]))
    print('D %d tokens %d batch %4d  ' % (dim, tokens, B) + '  '.join('%d: %s' % (k, ' / '.join('%.0f' % v for v in res[k])) for k in sorted(res)) +
          '   best %d (default %d)' % (best, len(plan.slice_sizes(B, 3)) if B >= 96 else 1), flush=True)

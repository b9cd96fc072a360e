"""Generate tests/golden/kat_fuzz.npz by RUNNING THE REAL REFERENCE operator classes on randomised inputs (CPU):
    python oracle/gen_golden_fuzz.py
QIntLayerNorm.forward 'int' (layers.py:255-289) with zero / tiny / huge gamma, PTF input scales, power-of-two and other output
scales, in_scale_expand 1 and 4;  QIntSoftmax (layers.py:323-376) with scaling factors 2^-1 .. 2^-9, -100 masks, saturated and
all-equal rows.  Only inputs and outputs (data) are stored."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_golden as G                                    # noqa: E402

GOLD = G.GOLD


def main():
    _, ref_models, _ = G.import_reference()
    from models.ptq.bit_type import BIT_TYPE_DICT
    g = torch.Generator().manual_seed(1234)
    rnd = lambda *s, std=1.0: torch.randn(*s, generator=g) * std          # noqa: E731
    ri = lambda lo, hi, *s: torch.randint(lo, hi, s if s else (1,), generator=g)   # noqa: E731
    out = {}
    n_ln = 24
    for i in range(n_ln):
        expand = 4 if i % 6 == 5 else 1
        C = int(ri(1, 25)) * 4 * expand
        rows = int(ri(1, 9))
        nin = C // expand
        base = float(2.0 ** ri(-9, 0)) * (1.0 + 0.37 * (i % 3))
        s_in = base * 2.0 ** ri(0, 4, nin).float()
        full = s_in if expand == 1 else s_in.unsqueeze(-1).expand(-1, expand).T.reshape(-1)
        codes = torch.clamp(torch.round(rnd(1, rows, C, std=float(torch.rand(1, generator=g)) * 60 + 1)), -128, 127)
        gamma = rnd(C) * (10.0 ** float(ri(-3, 2)))
        if i % 4 == 1:
            gamma[::3] = 0.0
        if i % 4 == 2:
            gamma[1] = 1e-9
            gamma[2] = -3e4
        beta = rnd(C, std=0.5) * (100.0 if i % 5 == 3 else 1.0)
        cs = 2.0 ** ri(-3, 3, C).float()
        s_a = float(2.0 ** ri(-7, -1))
        mult = 1.0 if i % 7 else 1.3
        ln = ref_models.QIntLayerNorm(C)
        ln.weight.data, ln.bias.data, ln.mode = gamma.clone(), beta.clone(), 'int'
        class Q: pass                                                                # noqa: E701
        qi, qo = Q(), Q()
        qi.scale, qo.scale = s_in, torch.tensor([s_a * mult])
        with torch.no_grad():
            y = ln(codes * full.reshape(1, 1, -1), qi, qo, cs, expand)
        p = 'ln/%d/' % i
        out[p + 'codes'], out[p + 'in_scale'], out[p + 'out_scale'] = codes.to(torch.int8).numpy(), s_in.numpy(), (qo.scale * cs).numpy()
        out[p + 'gamma'], out[p + 'beta'], out[p + 'out'], out[p + 'expand'] = gamma.numpy(), beta.numpy(), y.numpy(), np.int64(expand)
    out['ln/n'] = np.int64(n_ln)
    n_lis = 27
    for i in range(n_lis):
        e = 1 + i % 9
        sf = torch.tensor([2.0 ** -e])
        N = int(ri(2, 66))
        codes = torch.clamp(torch.round(rnd(2, 2, N, N, std=float(torch.rand(1, generator=g)) * 70 + 0.5)), -128, 127)
        codes[0, 0, 0, :] = 17
        if i % 3 == 0:                      # shifted-window style mask: -100 added to the dequantised scores
            m = (torch.rand(N, N, generator=g) < 0.4).float() * -100.0
            m.fill_diagonal_(0.0)
            x = codes * sf + m
        else:
            x = codes * sf
        sm = ref_models.QIntSoftmax(log_i_softmax=True, bit_type=BIT_TYPE_DICT['uint4'], quantizer_str='log2')
        with torch.no_grad():
            pr = sm(x, sf)
        p = 'lis/%d/' % i
        out[p + 'x_over_sf'], out[p + 'e'], out[p + 'probs'] = torch.round(x / sf).to(torch.int32).numpy(), np.int64(e), pr.numpy()
    out['lis/n'] = np.int64(n_lis)
    np.savez_compressed(os.path.join(GOLD, 'kat_fuzz.npz'), **out)
    print('kat_fuzz: wrote %d arrays' % len(out))


if __name__ == '__main__':
    main()

"""CPU: per-operator known-answer vectors captured from the reference classes (tests/golden/kat_ops.npz)."""
import numpy as np
import torch

from conftest import load_golden


def test_uniform_quantizer_ties_and_clamps(oracle):
    g = load_golden('kat_ops')
    x, s = torch.from_numpy(g['uq/x']), torch.from_numpy(g['uq/scale'])
    for bt in ('int8', 'int4', 'uint4'):
        lo, hi = oracle.BITS[bt]
        out = oracle.fake_quant(x, s, lo, hi, (1, -1))
        assert np.array_equal(out.numpy(), g['uq/%s/out' % bt]), bt


def test_log_int_softmax(oracle):
    g = load_golden('kat_ops')
    for e in range(3, 9):
        codes = torch.from_numpy(g['lis/%d/codes' % e]).float()
        k = oracle.lis_int(codes, torch.tensor([2.0 ** -e]))
        assert np.array_equal(oracle.lis_probs(k).numpy(), g['lis/%d/probs' % e]), e


def test_int_layernorm(oracle):
    g = load_golden('kat_ops')
    for tag in ('a', 'b', 'z'):
        ex = int(g['ln/%s/expand' % tag])
        s_in = torch.from_numpy(g['ln/%s/in_scale' % tag])
        if ex != 1:      # in_scale_expand (Swin PatchMerging), layers.py:257-259
            s_in = s_in.unsqueeze(-1).expand(-1, ex).T.reshape(-1)
        codes = torch.from_numpy(g['ln/%s/codes' % tag]).float()
        out_scale = torch.from_numpy(g['ln/%s/out_scale' % tag])
        y = oracle.int_layernorm(codes * s_in.reshape(1, 1, -1), s_in, torch.from_numpy(g['ln/%s/gamma' % tag]),
                                 torch.from_numpy(g['ln/%s/beta' % tag]), out_scale)
        assert np.array_equal((y * out_scale.reshape(1, 1, -1)).numpy(), g['ln/%s/out' % tag]), tag
        assert tag == 'z' or np.abs(y.numpy()).max() > 127      # LN output is NOT clamped (layers.py:288-289)


def test_minmax_pot_search(oracle):
    g = load_golden('kat_ops')
    xa = torch.from_numpy(g['mm/act/x'])
    ob = oracle._MinMax('activation')
    ob.update(xa, 'layer_wise')
    assert np.array_equal(ob.params(xa, 'int8', 'layer_wise').numpy(), g['mm/act/scale'])
    w, b = torch.from_numpy(g['mm/w']), torch.from_numpy(g['mm/b'])
    ob = oracle._MinMax('linear_weight')
    for bt in oracle.CALIB_BIT_ORDER:
        mode = 'layer_wise' if bt == 'int8' else 'channel_wise'
        ob.update(w, mode)
        s = ob.params(xa, bt, mode, [b])
        assert np.array_equal(s.numpy().reshape(g['mm/w/%s' % bt].shape), g['mm/w/%s' % bt]), bt


def test_ptf(oracle):
    g = load_golden('kat_ops')
    s = oracle.ptf_params(torch.from_numpy(g['ptf/x']))
    assert np.array_equal(s.numpy(), g['ptf/scale'])
    r = s / s.min()
    assert set(np.unique(r.numpy()).tolist()) <= {1.0, 2.0, 4.0, 8.0}


def test_randomised_reference_vectors(oracle):
    """tests/golden/kat_fuzz.npz (oracle/gen_golden_fuzz.py, produced by the REAL reference classes): 24 QIntLayerNorm cases (zero /
    tiny / huge gamma, PTF scales, non power-of-two output scales, in_scale_expand 4) and 27 QIntSoftmax cases (sf 2^-1..2^-9, -100
    masks); NaN/inf rows (zero variance) must agree as well."""
    g = load_golden('kat_fuzz')
    for i in range(int(g['ln/n'])):
        p = 'ln/%d/' % i
        ex = int(g[p + 'expand'])
        s_in = torch.from_numpy(g[p + 'in_scale'])
        if ex != 1:
            s_in = s_in.unsqueeze(-1).expand(-1, ex).T.reshape(-1)
        codes = torch.from_numpy(g[p + 'codes']).float()
        out_scale = torch.from_numpy(g[p + 'out_scale'])
        y = oracle.int_layernorm(codes * s_in.reshape(1, 1, -1), s_in, torch.from_numpy(g[p + 'gamma']), torch.from_numpy(g[p + 'beta']), out_scale)
        got, want = (y * out_scale.reshape(1, 1, -1)).numpy(), g[p + 'out']
        assert np.array_equal(got, want, equal_nan=True), (i, int((got != want).sum()))
    for i in range(int(g['lis/n'])):
        p = 'lis/%d/' % i
        k = oracle.lis_int(torch.from_numpy(g[p + 'x_over_sf']).long(), torch.tensor([2.0 ** -int(g[p + 'e'])]))
        assert np.array_equal(oracle.lis_probs(k).numpy(), g[p + 'probs']), i

"""CPU: the oracle (canonical restatement) against vectors produced by the REAL reference
(oracle/gen_golden.py, run in the build container)."""
import numpy as np
import pytest
import torch

from conftest import golden_calib, load_golden

CFGS = ('q8', 'q4', 'qmix')


def _bits(g, tag, L):
    return {'q8': [8] * L, 'q4': [4] * L, 'qmix': [int(b) for b in g['bit_qmix']]}[tag]


def test_micro_float_forward_bitexact(micro, oracle):
    orc = oracle.OracleViT(micro['arch'], micro['sd'])
    out = orc.float_forward(micro['x_ev'])
    assert np.array_equal(out.numpy(), micro['g']['fp_logits'])       # north_star asks <= 1e-5; we get 0


def test_micro_calibration_identical(micro, oracle):
    g = micro['g']
    orc = oracle.OracleViT(micro['arch'], micro['sd'])
    with torch.no_grad():
        cal = orc.calibrate(micro['x_cal'])
    assert np.abs(cal.numpy() - g['calib_logits']).max() <= 1e-5
    flat = oracle.flatten_calib(orc.calib)
    assert len(flat) == sum(1 for k in g.files if k.startswith('calib/'))
    for k, v in flat.items():
        assert np.array_equal(v.numpy().reshape(g['calib/' + k].shape), g['calib/' + k]), k
    gd = np.array([[float(v) for v in row] for row in orc.global_distance])
    assert np.allclose(gd, g['global_distance'], rtol=1e-6, atol=0)


@pytest.mark.parametrize('tag', CFGS)
def test_micro_quant_forward_all_taps_bitexact(micro, oracle, tag):
    g = micro['g']
    orc = oracle.OracleViT(micro['arch'], micro['sd'])
    orc.calib = micro['calib']
    L = 4 * micro['arch']['depth'] + 2
    taps = {}
    out = orc.quant_forward(micro['x_ev'], _bits(g, tag, L), taps)
    assert np.array_equal(out.numpy(), g['logits/' + tag])
    assert np.array_equal(out.topk(5, 1, True, True)[1].numpy(), g['top5/' + tag])
    n = 0
    for k in g.files:
        if k.startswith('taps/%s/' % tag):
            name = k.split('/', 2)[2]
            assert np.array_equal(taps[name].numpy().reshape(g[k].shape).astype(np.int64), g[k].astype(np.int64)), name
            n += 1
    assert tag == 'q4' or n == 27
    assert orc.flops() == [int(v) for v in g['flops']]


def test_bit_config_errors(micro, oracle):
    orc = oracle.OracleViT(micro['arch'], micro['sd'])
    orc.calib = micro['calib']
    with pytest.raises(ValueError):
        orc.quant_forward(micro['x_ev'], [8] * 9 + [6])


def test_deit_small_against_reference(oracle, synth):
    """Full-size DeiT-S (BASELINE config 2 shape), weights regenerated from the seed, calibration state of
    the REAL reference.  The reference's fp32 simulation has ~1e-6/element order-dependent roundings (MKL
    K-blocked bias add in fc2, LIS sum of exp_int up to 2^50, A&S-polynomial GELU); the canonical oracle
    does not imitate them.  Every tap before the first such event is bit-equal; after it a W8A8 network
    amplifies a single +-1 code (each flipped GEMM input flips ~10% of that row's outputs), so later taps
    are compared statistically and by top-1."""
    g = load_golden('deit_small')
    arch = synth.ARCHS['deit_small']
    sd = synth.vit_state_dict(arch, int(g['seed']))
    orc = oracle.OracleViT(arch, sd)
    orc.calib = golden_calib(g, oracle)
    n = int(g['n_eval'])
    x = synth.images(int(g['seed']), n, 224, offset=1000)
    taps = {}
    out = orc.quant_forward(x, [8] * 50, taps)
    exact = ('patch_embed.qact', 'qact1', 'blocks.0.attn.qact0', 'blocks.0.attn.qact2', 'blocks.0.attn.qact3',
             'blocks.0.qact2', 'blocks.0.mlp.qact0')
    for name in exact:
        ref = g['taps/q8/' + name]
        assert np.array_equal(taps[name].numpy().reshape(ref.shape).astype(np.int64), ref.astype(np.int64)), name
    ref = g['taps/q8/blocks.0.mlp.qact2']
    d = np.abs(taps['blocks.0.mlp.qact2'].numpy().reshape(ref.shape).astype(np.int64) - ref)
    assert d.max() <= 1 and (d > 0).sum() <= 8          # first divergence: fc2 bias rounding
    s_o = float(g['calib/act_out'])
    for tag, bits in (('q8', [8] * 50), ('q4', [4] * 50), ('qmix', [int(b) for b in g['bit_qmix']])):
        o = out if tag == 'q8' else orc.quant_forward(x, bits)
        agree = int((o.argmax(1).numpy() == g['logits/' + tag].argmax(1)).sum())
        assert agree == int(g['canon_vs_ref/%s/top1_agree' % tag]), (tag, agree)
        if tag != 'q4':       # the random-weight 4-bit net has top-2 margins of 1-2 codes: top-1 is not stable there
            assert agree >= n - 1, (tag, agree)
        assert np.abs(o.numpy() - g['logits/' + tag]).max() / s_o <= 24
        # top-5 sets overlap strongly even after rounding chaos
        t5 = o.topk(5, 1, True, True)[1].numpy()
        assert np.mean([len(set(t5[i]) & set(g['top5/' + tag][i])) for i in range(n)]) >= 2.5


def test_deit_tiny_float_config1(oracle, synth):
    """BASELINE config 1 (deit_tiny fp32, no --quant): float path."""
    g = load_golden('deit_tiny_fp')
    arch = synth.ARCHS['deit_tiny']
    orc = oracle.OracleViT(arch, synth.vit_state_dict(arch, int(g['seed'])))
    out = orc.float_forward(synth.images(int(g['seed']), 4, 224, offset=1000))
    assert np.abs(out.numpy() - g['fp_logits']).max() <= 1e-5
    assert orc.flops() == [int(v) for v in g['flops']]

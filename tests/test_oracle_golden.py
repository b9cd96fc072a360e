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
    s_o = float(g['calib/act_out'].reshape(-1)[0])
    for tag, bits in (('q8', [8] * 50), ('q4', [4] * 50), ('qmix', [int(b) for b in g['bit_qmix']])):
        o = out if tag == 'q8' else orc.quant_forward(x, bits)
        agree = int((o.argmax(1).numpy() == g['logits/' + tag].argmax(1)).sum())
        # (this fixture's random head has top-2 margins of 1-2 codes at 4 bits, so the count below records the drift rather than
        # bounding it; test_deit_small_margin_top1_identical is the top-1 assertion proper)
        assert agree == int(g['canon_vs_ref/%s/top1_agree' % tag]), (tag, agree)
        if tag != 'q4':       # the random-weight 4-bit net has top-2 margins of 1-2 codes: top-1 is not stable there
            assert agree >= n - 1, (tag, agree)
        assert np.abs(o.numpy() - g['logits/' + tag]).max() / s_o <= 24
        # top-5 sets overlap strongly even after rounding chaos
        t5 = o.topk(5, 1, True, True)[1].numpy()
        assert np.mean([len(set(t5[i]) & set(g['top5/' + tag][i])) for i in range(n)]) >= 2.5


def _first_tap_diff(taps, g, tag='q8'):
    for k, v in taps.items():
        key = 'taps/%s/%s' % (tag, k)
        if key in g.files:
            n = int((v.numpy().reshape(g[key].shape).astype(np.int64) != g[key].astype(np.int64)).sum())
            if n:
                return k, n
    return None


def test_deit_small_decomposition(oracle, synth):
    """DESIGN.md section 2 as a test.  With the four canonical pieces of the oracle swapped back for the torch-CPU operations the
    reference calls (imitate = linear, gelu, lis, ln) the oracle IS the reference run of this container at DeiT-S size: 0 of 4000
    logit codes and 0 tap elements differ for [8]*50, [4]*50 and the mixed list.  Leaving exactly one piece canonical shows what that
    piece is worth: the fc2 bias order moves one code of blocks.0.mlp.qact2 and no logit; the polynomial-erf GELU alone explains the
    whole 8-bit mismatch; the fp32 exp_int sum and the LayerNorm statistics matter only at 4 bits."""
    g = load_golden('deit_small')
    arch = synth.ARCHS['deit_small']
    sd = synth.vit_state_dict(arch, int(g['seed']))
    calib = golden_calib(g, oracle)
    x = synth.images(int(g['seed']), int(g['n_eval']), 224, offset=1000)
    s_o = float(g['calib/act_out'].reshape(-1)[0])
    cfgs = {'q8': [8] * 50, 'q4': [4] * 50, 'qmix': [int(b) for b in g['bit_qmix']]}

    def run(imitate, tag):
        orc = oracle.OracleViT(arch, sd, imitate=imitate)
        orc.calib = calib
        taps = {}
        out = orc.quant_forward(x, cfgs[tag], taps)
        d = np.abs(np.round((out.numpy() - g['logits/' + tag]) / s_o))
        return int((d > 0).sum()), (_first_tap_diff(taps, g) if tag == 'q8' else None)

    every = oracle.OracleViT.IMITATE
    for tag in cfgs:
        assert run(every, tag) == (0, None), tag                     # all native: the reference, bit for bit
    drop = lambda name: tuple(a for a in every if a != name)
    n_lin, first_lin = run(drop('linear'), 'q8')
    assert n_lin == 0 and first_lin is not None and first_lin[0] == 'blocks.0.mlp.qact2' and first_lin[1] <= 8
    n_gelu, first_gelu = run(drop('gelu'), 'q8')
    assert n_gelu == int(g['canon_vs_ref/q8/logit_codes_differ']) and first_gelu is not None      # the whole 8-bit mismatch
    assert run(drop('lis'), 'q8') == (0, None) and run(drop('ln'), 'q8') == (0, None)
    assert run(drop('lis'), 'q4')[0] > 0 and run(drop('ln'), 'q4')[0] > 0


@pytest.mark.parametrize('name', ['vit_base', 'deit_tiny'])
def test_other_architectures_are_the_reference(oracle, synth, name):
    """BASELINE configs 3 / 5 (ViT-B and DeiT-B share the architecture: 768 wide, 12 heads, fc2 with a 3072-deep contraction) and the
    architecture of config 1 (DeiT-T: 192 wide, 3 heads) on the quantized path against the REAL reference (tests/golden/vit_base.npz,
    deit_tiny.npz: its calibration state, logits and top-level taps for [8]*50, [4]*50 and the mixed list).  With the four torch-CPU pieces
    switched back in the oracle IS the reference at these sizes too - 0 logit codes differ; the canonical oracle (what the HIP engine
    computes) agrees up to the platform-dependent roundings, counted in the fixture - at DeiT-T that count is ZERO for [8]*50 and [4]*50."""
    g = load_golden(name)
    arch = synth.ARCHS[name]
    sd = synth.vit_state_dict(arch, int(g['seed']))
    calib = golden_calib(g, oracle)
    x = synth.images(int(g['seed']), int(g['n_eval']), 224, offset=1000)
    s_o = float(g['calib/act_out'].reshape(-1)[0])
    cfgs = {'q8': [8] * 50, 'q4': [4] * 50, 'qmix': [int(b) for b in g['bit_qmix']]}
    ref_like = oracle.OracleViT(arch, sd, imitate=oracle.OracleViT.IMITATE)
    ref_like.calib = calib
    for tag, bits in cfgs.items():
        out = ref_like.quant_forward(x, bits)
        assert int((np.round((out.numpy() - g['logits/' + tag]) / s_o) != 0).sum()) == 0, tag
    orc = oracle.OracleViT(arch, sd)
    orc.calib = calib
    for tag in (('q8', 'q4') if name == 'deit_tiny' else ('q8',)):
        out = orc.quant_forward(x, cfgs[tag])
        d = np.abs(np.round((out.numpy() - g['logits/' + tag]) / s_o))
        assert int((d > 0).sum()) == int(g['canon_vs_ref/%s/logit_codes_differ' % tag])
        assert int((out.argmax(1).numpy() == g['logits/' + tag].argmax(1)).sum()) == int(g['canon_vs_ref/%s/top1_agree' % tag])
        if name == 'deit_tiny':
            assert np.array_equal(out.numpy(), g['logits/' + tag]), tag          # canonical oracle == REAL reference, every logit


def test_deit_small_exact_images_equal_the_reference(oracle, synth):
    """BASELINE config 2 (the headline shape) on evaluation images where no platform-dependent rounding of the reference flips a code
    (tests/golden/deit_small_exact.npz: 13 / 11 / 11 of 48 candidate images for [8]*50 / [4]*50 / the mixed list; eight kept per list):
    the canonical oracle equals the REAL reference on EVERY logit.  The GPU twin asserts the same of the HIP engine."""
    g = load_golden('deit_small_exact')
    arch = synth.ARCHS['deit_small']
    sd = synth.vit_state_dict(arch, int(g['seed']))
    orc = oracle.OracleViT(arch, sd)
    orc.calib = golden_calib(g, oracle)
    cfgs = {'q8': [8] * 50, 'q4': [4] * 50, 'qmix': [int(b) for b in g['bit_qmix']]}
    for tag in ('q8', 'q4'):
        idx = [int(i) for i in g['exact_images/' + tag]]
        assert len(idx) == 8 and int(g['n_exact/' + tag]) >= 8
        x = torch.cat([synth.images(int(g['seed']), 1, 224, offset=int(g['first_offset']) + i) for i in idx[:4]])
        out = orc.quant_forward(x, cfgs[tag])
        assert np.array_equal(out.numpy(), g['logits/' + tag][:4]), tag


def test_deit_small_margin_top1_identical(oracle, synth):
    """north_star's 'identical top-1 indices' at DeiT-S size, made testable: the head of this fixture carries one planted class per
    evaluation image (oracle/gen_golden.py::plant_head_margin), so every top-1 leads the runner-up by > 80 codes while the
    platform-dependent roundings of the reference move a logit by at most 8.  The canonical oracle (what the HIP engine computes) agrees
    with the REAL reference on all 8 images for [8]*50, [4]*50 and the mixed list; with all four pieces imitated it is bit-equal."""
    from conftest import planted_state_dict
    g = load_golden('deit_small_margin')
    arch = synth.ARCHS['deit_small']
    sd = planted_state_dict(g, synth, arch)
    calib = golden_calib(g, oracle)
    n = int(g['n_eval'])
    x = synth.images(int(g['seed']), n, 224, offset=1000)
    s_o = float(g['calib/act_out'].reshape(-1)[0])
    assert np.abs(oracle.OracleViT(arch, sd).float_forward(x).numpy() - g['fp_logits']).max() <= 1e-4
    for tag, bits in (('q8', [8] * 50), ('q4', [4] * 50), ('qmix', [int(b) for b in g['bit_qmix']])):
        ref = g['logits/' + tag]
        codes = np.round(ref / s_o)
        srt = np.sort(codes, 1)
        assert (srt[:, -1] - srt[:, -2]).min() > 60                   # the reference's own top-2 margin, in codes
        assert np.array_equal(ref.argmax(1), g['head_classes'])
        orc = oracle.OracleViT(arch, sd)
        orc.calib = calib
        out = orc.quant_forward(x, bits).numpy()
        assert np.abs(np.round((out - ref) / s_o)).max() <= 13        # canonical vs as-run: bounded drift
        assert np.array_equal(out.argmax(1), ref.argmax(1)), tag      # identical top-1: all n images
        nat = oracle.OracleViT(arch, sd, imitate=oracle.OracleViT.IMITATE)
        nat.calib = calib
        assert np.array_equal(nat.quant_forward(x, bits).numpy(), ref), tag


def test_deit_tiny_float_config1(oracle, synth):
    """BASELINE config 1 (deit_tiny fp32, no --quant): float path."""
    g = load_golden('deit_tiny_fp')
    arch = synth.ARCHS['deit_tiny']
    orc = oracle.OracleViT(arch, synth.vit_state_dict(arch, int(g['seed'])))
    out = orc.float_forward(synth.images(int(g['seed']), 4, 224, offset=1000))
    assert np.abs(out.numpy() - g['fp_logits']).max() <= 1e-5
    assert orc.flops() == [int(v) for v in g['flops']]


def test_micro_fp_input_against_reference(oracle, synth):
    """VisionTransformer(input_quant=False) - the configuration of the reference's vit_large factory (vit_fquant.py:925): the fp32
    image feeds the fake-quantised patch-embed convolution.  Fixture from the REAL reference at micro size (oracle/gen_golden.py
    micro_fp_input): calibration identical, all 26 taps and the logits of the three bit configurations bit-equal with the oracle's
    canonical reading of that convolution (fp64 sum of the exact products, one rounding)."""
    g = load_golden('micro_vit_fp_input')
    arch = dict(synth.ARCHS['micro'], input_quant=False)
    sd = {k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith('w/')}
    orc = oracle.OracleViT(arch, sd)
    assert np.array_equal(orc.float_forward(torch.from_numpy(g['x_ev'])).numpy(), g['fp_logits'])
    with torch.no_grad():
        cal = orc.calibrate(torch.from_numpy(g['x_cal']))
    assert np.abs(cal.numpy() - g['calib_logits']).max() <= 1e-5
    flat = oracle.flatten_calib(orc.calib)
    assert 'qact_input' not in flat and len(flat) == sum(1 for k in g.files if k.startswith('calib/'))
    for k, v in flat.items():
        assert np.array_equal(v.numpy().reshape(g['calib/' + k].shape), g['calib/' + k]), k
    for tag in CFGS:
        taps = {}
        out = orc.quant_forward(torch.from_numpy(g['x_ev']), _bits(g, tag, 10), taps)
        assert np.array_equal(out.numpy(), g['logits/' + tag]), tag
        n = 0
        for k in g.files:
            if k.startswith('taps/%s/' % tag):
                name = k.split('/', 2)[2]
                assert np.array_equal(taps[name].numpy().reshape(g[k].shape).astype(np.int64), g[k].astype(np.int64)), (tag, name)
                n += 1
        assert tag == 'q4' or n == 26

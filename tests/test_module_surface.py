"""CPU: the product's drop-in module surface (calibration + float path) against the REAL reference's outputs
(tests/golden) -- identical scales/exponents, float pass bit-equal (north_star asks <= 1e-5)."""
from functools import partial

import numpy as np
import pytest
import torch

from conftest import golden_calib, load_golden


def _micro_model(dva, micro):
    a = micro['arch']
    m = dva.VisionTransformer(img_size=a['img_size'], patch_size=a['patch_size'], embed_dim=a['embed_dim'], depth=a['depth'],
                              num_heads=a['num_heads'], num_classes=a['num_classes'], mlp_ratio=a['mlp_ratio'], qkv_bias=True,
                              norm_layer=partial(dva.QIntLayerNorm, eps=1e-6), input_quant=True, cfg=dva.Config(True, True, 'minmax'))
    missing, unexpected = m.load_state_dict(micro['sd'], strict=False)
    assert not missing and not unexpected
    return m.eval()


@pytest.fixture(scope='module')
def dva():
    import diff_vit_amd
    return diff_vit_amd


def test_exports_match_reference_surface(dva):
    for name in ('QAct', 'QConv2d', 'QLinear', 'QIntLayerNorm', 'QIntSoftmax', 'BIT_TYPE_DICT', 'Config', 'deit_tiny_patch16_224',
                 'deit_small_patch16_224', 'deit_base_patch16_224', 'vit_base_patch16_224', 'vit_large_patch16_224'):
        assert hasattr(dva, name), name
    assert sorted(dva.BIT_TYPE_DICT) == ['int4', 'int8', 'uint3', 'uint4', 'uint8']
    c = dva.Config()
    assert c.BIT_TYPE_W.name == 'int4' and c.BIT_TYPE_A.name == 'int8' and c.BIT_TYPE_S.name == 'uint4'
    assert c.OBSERVER_A_LN == 'ptf' and c.CALIBRATION_MODE_W == 'channel_wise' and c.INT_NORM and c.INT_SOFTMAX
    q = dva.QLinear(8, 4)
    for attr in ('quant', 'calibrate', 'last_calibrate', 'bit_type', 'observer', 'quantizer', 'module_type'):
        assert hasattr(q, attr)
    assert hasattr(q.quantizer, 'dic_scale') and hasattr(q.observer, 'max_val') and q.observer.symmetric
    with pytest.raises(AssertionError):
        dva.QIntLayerNorm((8,))


def test_float_forward_and_flops(dva, micro):
    m = _micro_model(dva, micro)
    with torch.no_grad():
        out, flops, gd = m(micro['x_ev'])
    assert np.array_equal(out.numpy(), micro['g']['fp_logits'])
    assert flops == [int(v) for v in micro['g']['flops']] and gd == []
    assert flops == m.flops()


def test_calibration_identical_to_reference(dva, micro):
    g = micro['g']
    m = _micro_model(dva, micro)
    m.model_open_calibrate()
    with torch.no_grad():
        m.model_open_last_calibrate()
        out, flops, gd = m(micro['x_cal'], plot=False)
    m.model_close_calibrate()
    m.model_quant()
    assert np.abs(out.numpy() - g['calib_logits']).max() <= 1e-5
    flat = dva.calib_io.flatten(m.export_calib())
    assert len(flat) == sum(1 for k in g.files if k.startswith('calib/'))
    for k, v in flat.items():
        assert np.array_equal(v.numpy().reshape(g['calib/' + k].shape), g['calib/' + k]), k
    gdn = np.array([[float(v) for v in row] for row in gd])
    assert gdn.shape == g['global_distance'].shape and np.allclose(gdn, g['global_distance'], rtol=1e-5, atol=0)
    # quantized forward has no CPU fallback
    with pytest.raises(RuntimeError):
        m(micro['x_ev'], [8] * 10)
    with pytest.raises(ValueError):
        m(micro['x_ev'], None)


def test_fp_input_model_calibration_identical_to_reference(dva, synth):
    """input_quant=False (the vit_large factory's configuration) through the drop-in surface: the product's calibration reproduces
    the real reference's scales of tests/golden/micro_vit_fp_input.npz; export_calib has no qact_input entry."""
    from functools import partial
    g = load_golden('micro_vit_fp_input')
    a = synth.ARCHS['micro']
    m = dva.VisionTransformer(img_size=a['img_size'], patch_size=a['patch_size'], embed_dim=a['embed_dim'], depth=a['depth'],
                              num_heads=a['num_heads'], num_classes=a['num_classes'], mlp_ratio=a['mlp_ratio'], qkv_bias=True,
                              norm_layer=partial(dva.QIntLayerNorm, eps=1e-6), input_quant=False, cfg=dva.Config(True, True, 'minmax'))
    m.load_state_dict({k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith('w/')}, strict=False)
    m.eval()
    with torch.no_grad():
        assert np.array_equal(m(torch.from_numpy(g['x_ev']))[0].numpy(), g['fp_logits'])
    out = dva.harness.calibrate_model(m, torch.from_numpy(g['x_cal']))[0]
    assert np.abs(out.numpy() - g['calib_logits']).max() <= 1e-5
    flat = dva.calib_io.flatten(m.export_calib())
    assert 'qact_input' not in flat and len(flat) == sum(1 for k in g.files if k.startswith('calib/'))
    for k, v in flat.items():
        assert np.array_equal(v.numpy().reshape(g['calib/' + k].shape), g['calib/' + k]), k
    assert dva.vit_large_patch16_224.__name__ == 'vit_large_patch16_224'


def test_module_level_ops_against_kats(dva):
    g = load_golden('kat_ops')
    for e in range(3, 9):
        sm = dva.QIntSoftmax(log_i_softmax=True, bit_type=dva.BIT_TYPE_DICT['uint4'], quantizer_str='log2')
        sf = torch.tensor([2.0 ** -e])
        p = sm(torch.from_numpy(g['lis/%d/codes' % e]).float() * sf, sf)
        assert np.array_equal(p.numpy(), g['lis/%d/probs' % e]), e
    for tag in ('a', 'b', 'z'):
        C = g['ln/%s/gamma' % tag].shape[0]
        ln = dva.QIntLayerNorm(C)
        ln.weight.data = torch.from_numpy(g['ln/%s/gamma' % tag]); ln.bias.data = torch.from_numpy(g['ln/%s/beta' % tag])
        ln.mode = 'int'
        class Q: pass
        qi, qo = Q(), Q()
        qi.scale = torch.from_numpy(g['ln/%s/in_scale' % tag]); qo.scale = torch.tensor([2.0 ** -4])
        ex = int(g['ln/%s/expand' % tag])
        full = qi.scale if ex == 1 else qi.scale.unsqueeze(-1).expand(-1, ex).T.reshape(-1)
        cs = torch.from_numpy(g['ln/%s/out_scale' % tag]) / qo.scale
        y = ln(torch.from_numpy(g['ln/%s/codes' % tag]).float() * full.reshape(1, 1, -1), qi, qo, cs, ex)
        assert np.array_equal(y.detach().numpy(), g['ln/%s/out' % tag]), tag
    M, N = dva.QIntLayerNorm(4).get_MN(torch.from_numpy(g['mn/A']))
    assert np.array_equal(M.numpy(), g['mn/M']) and np.array_equal(N.numpy(), g['mn/N'])
    from diff_vit_amd.ptq.observer import build_observer
    xa = torch.from_numpy(g['mm/act/x'])
    ob = build_observer('minmax', 'activation', dva.BIT_TYPE_DICT['int8'], 'layer_wise'); ob.update(xa)
    assert np.array_equal(ob.get_quantization_params(xa)[0].numpy(), g['mm/act/scale'])
    w, b = torch.from_numpy(g['mm/w']), torch.from_numpy(g['mm/b'])
    ob = build_observer('minmax', 'linear_weight', dva.BIT_TYPE_DICT['int4'], 'channel_wise')
    for bt in ('uint3', 'uint4', 'int4', 'int8'):
        ob.bit_type = dva.BIT_TYPE_DICT[bt]; ob.calibration_mode = 'layer_wise' if bt == 'int8' else 'channel_wise'
        ob.update(w)
        s, _ = ob.get_quantization_params(xa, others=[b])
        assert np.array_equal(s.numpy().reshape(g['mm/w/%s' % bt].shape), g['mm/w/%s' % bt]), bt
    xp = torch.from_numpy(g['ptf/x'])
    ob = build_observer('ptf', 'activation', dva.BIT_TYPE_DICT['int8'], 'channel_wise'); ob.update(xp)
    assert np.array_equal(ob.get_quantization_params(xp)[0].numpy(), g['ptf/scale'])
    from diff_vit_amd.ptq.quantizer import build_quantizer
    x, s = torch.from_numpy(g['uq/x']), torch.from_numpy(g['uq/scale'])
    for bt in ('int8', 'int4', 'uint4'):
        qz = build_quantizer('uniform', dva.BIT_TYPE_DICT[bt], None, 'activation')
        qz.scale, qz.zero_point = s, torch.zeros(16, dtype=torch.int64)
        assert np.array_equal(qz(x).numpy(), g['uq/%s/out' % bt]), bt


def test_deit_small_calibration_identical_to_reference(dva):
    """full-size DeiT-S: the batched PoT search reproduces every one of the reference's 495 calibrated tensors,
    the calibration-pass logits and the weight-MSE bookkeeping (global_distance) -- in ~2 s instead of ~1-2 min."""
    g = load_golden('deit_small')
    seed = int(g['seed'])
    m = dva.deit_small_patch16_224(cfg=dva.Config(True, True, 'minmax'))
    m.load_state_dict(dva.synth.vit_state_dict(dva.synth.ARCHS['deit_small'], seed), strict=False)
    m.eval()
    x = dva.synth.images(seed, int(g['n_calib']), 224)
    m.model_open_calibrate()
    with torch.no_grad():
        m.model_open_last_calibrate()
        out, flops, gd = m(x, plot=False)
    m.model_close_calibrate()
    m.model_quant()
    assert np.abs(out.numpy() - g['calib_logits']).max() <= 1e-5
    flat = dva.calib_io.flatten(m.export_calib())
    assert len(flat) == sum(1 for k in g.files if k.startswith('calib/')) == 495
    for k, v in flat.items():
        assert np.array_equal(v.numpy().reshape(g['calib/' + k].shape), g['calib/' + k]), k
    gdn = np.array([[float(v) for v in row] for row in gd])
    assert np.allclose(gdn, g['global_distance'], rtol=1e-5, atol=0)
    assert flops == [int(v) for v in g['flops']]


def test_local_checkpoint_ingestion(tmp_path, synth):
    """checkpoint.load_checkpoint: a DeiT-style .pth (state_dict under 'model') and a Flax-layout .npz (synthetic round trip)."""
    import diff_vit_amd as dva
    from diff_vit_amd import checkpoint
    arch = synth.ARCHS['micro']
    sd = synth.vit_state_dict(arch, 31)
    mk = lambda: dva.VisionTransformer(img_size=arch['img_size'], patch_size=arch['patch_size'], embed_dim=arch['embed_dim'],  # noqa: E731
                                       depth=arch['depth'], num_heads=arch['num_heads'], num_classes=arch['num_classes'],
                                       mlp_ratio=arch['mlp_ratio'], qkv_bias=True, cfg=dva.Config(True, True, 'minmax'))
    torch.save({'model': sd}, str(tmp_path / 'w.pth'))
    np.savez(str(tmp_path / 'w.npz'), **checkpoint.state_dict_to_vit_npz(sd, arch['depth'], arch['num_heads']))
    for name in ('w.pth', 'w.npz'):
        m = mk()
        res = checkpoint.load_checkpoint(m, str(tmp_path / name))
        assert not res.unexpected_keys
        got = m.state_dict()
        for k, v in sd.items():
            assert torch.equal(got[k], v), (name, k)
    with pytest.raises(ValueError):
        bad = dict(sd); bad['head.weight'] = torch.zeros(3, 3)
        torch.save(bad, str(tmp_path / 'bad.pth'))
        checkpoint.load_checkpoint(mk(), str(tmp_path / 'bad.pth'))


def test_bit_config_error_conventions(synth):
    """an uncalibrated model in quant state: bit_config None raises ValueError before any engine call, like bit_pool.index(None) in the
    reference (vit_fquant.py:282); a width outside the pool raises KeyError from the BIT_TYPE_DICT lookup (layers.py:174-175) once a -1
    entry has sent the forward down the module graph (see test_bit_config_minus_one_is_the_reference_fp_fallback for -1 itself)."""
    import diff_vit_amd as dva
    arch = synth.ARCHS['micro']
    m = dva.VisionTransformer(img_size=arch['img_size'], patch_size=arch['patch_size'], embed_dim=arch['embed_dim'], depth=arch['depth'],
                              num_heads=arch['num_heads'], num_classes=arch['num_classes'], mlp_ratio=arch['mlp_ratio'], qkv_bias=True,
                              input_quant=True, cfg=dva.Config(True, True, 'minmax')).eval()
    m.model_quant()
    x = torch.zeros(1, 3, arch['img_size'], arch['img_size'])
    L = 4 * arch['depth'] + 2
    with pytest.raises(ValueError):
        m(x, None)
    with pytest.raises(RuntimeError):
        m(x, [8] * L)                       # all-quantized on a CPU tensor: the engine refuses (no fallback)


def test_qintlayernorm_module_vs_randomised_reference_vectors():
    """the drop-in QIntLayerNorm class on tests/golden/kat_fuzz.npz (24 cases from the real reference, see test_oracle_kat.py)."""
    import diff_vit_amd as dva
    g = load_golden('kat_fuzz')
    for i in range(int(g['ln/n'])):
        p = 'ln/%d/' % i
        C = g[p + 'gamma'].shape[0]
        ln = dva.QIntLayerNorm(C)
        ln.weight.data, ln.bias.data, ln.mode = torch.from_numpy(g[p + 'gamma']), torch.from_numpy(g[p + 'beta']), 'int'
        class Q: pass                                                   # noqa: E701
        qi, qo = Q(), Q()
        qi.scale, qo.scale = torch.from_numpy(g[p + 'in_scale']), torch.from_numpy(g[p + 'out_scale'])
        ex = int(g[p + 'expand'])
        full = qi.scale if ex == 1 else qi.scale.unsqueeze(-1).expand(-1, ex).T.reshape(-1)
        with torch.no_grad():
            y = ln(torch.from_numpy(g[p + 'codes']).float() * full.reshape(1, 1, -1), qi, qo, None, ex)
        assert np.array_equal(y.numpy(), g[p + 'out'], equal_nan=True), i


def test_float_scale_observers_match_reference():
    """EmaObserver / OmseObserver / PercentileObserver (FQ-ViT's non power-of-two observers, selectable through
    ``--quant-method``) against outputs of the REAL reference classes (tests/golden/kat_observers.npz, oracle/gen_golden_observers.py)."""
    from conftest import load_golden
    import diff_vit_amd as dva
    from diff_vit_amd.ptq.observer import build_observer
    g = load_golden('kat_observers')
    xs = [torch.from_numpy(g['x/%d' % i]) for i in range(3)]
    w = torch.from_numpy(g['w'])
    for name in ('ema', 'omse', 'percentile'):
        for bt in ('int8', 'uint8'):
            for mode in ('layer_wise', 'channel_wise'):
                if name == 'percentile' and mode == 'channel_wise':
                    with pytest.raises(AssertionError):
                        build_observer(name, 'activation', dva.BIT_TYPE_DICT[bt], mode).update(xs[0])
                    continue
                ob = build_observer(name, 'activation', dva.BIT_TYPE_DICT[bt], mode)
                for x in xs:
                    ob.update(x)
                # the product's QAct passes the fork's extra keywords (layers.py:216); they must be accepted
                s, zp = ob.get_quantization_params(xs[-1], attn=False, attn_para=None)
                key = '%s/act/%s/%s' % (name, bt, mode)
                assert np.array_equal(np.asarray(s.numpy()), g[key + '/scale']), key
                assert np.array_equal(np.asarray(zp.numpy()), g[key + '/zp']), key
                assert np.array_equal(np.asarray(ob.max_val.numpy()), g[key + '/max']), key
                assert np.array_equal(np.asarray(ob.min_val.numpy()), g[key + '/min']), key
        ob = build_observer(name, 'linear_weight', dva.BIT_TYPE_DICT['int8'], 'layer_wise')
        ob.update(w)
        s, zp = ob.get_quantization_params(w, others=[None])
        assert np.array_equal(np.asarray(s.numpy()), g['%s/w/scale' % name]) and np.array_equal(np.asarray(zp.numpy()), g['%s/w/zp' % name]), name


def test_model_dequant_leaves_the_fused_path_like_the_reference(dva, micro):
    """model_quant(); model_dequant() (vit_fquant.py:680-683) clears the per-module flags the reference's forward reads: the
    module-by-module graph runs again (NOT the float model: QIntLayerNorm stays in mode 'int', the softmax stays log-int).
    A single module with ``.quant = False`` does the same.  Expected logits: the REAL reference in the same states
    (tests/golden/micro_vit_dequant.npz, oracle/gen_golden_dequant.py).  All on CPU: the engine is never touched."""
    from conftest import load_golden
    gd = load_golden('micro_vit_dequant')
    m = _micro_model(dva, micro)
    x = micro['x_ev']
    with torch.no_grad():
        dva.harness.calibrate_model(m, micro['x_cal'])
        assert m._fused()
        with pytest.raises(RuntimeError):
            m(x, [8] * 10)                        # quant state on a CPU tensor: the engine refuses, no fallback
        m.model_dequant()
        assert not m._fused()
        for tag, bits in (('q8', 8), ('q4', 4)):
            out = m(x, [bits] * 10)[0]
            assert np.abs(out.numpy() - gd['dequant/' + tag]).max() <= 1e-5, tag
        m.model_quant()
        assert m._fused()
        m.blocks[0].mlp.fc2.quant = False
        assert not m._fused()
        out = m(x, [8] * 10)[0]                   # module-by-module graph (torch fake-quant) with fc2 of block 0 in float
    # fake-quant graph on the int8 grid: the canonical sums vs torch-CPU sums may move single codes
    s_o = float(m.act_out.quantizer.scale)
    assert np.abs(out.numpy() - gd['fc2_float/q8']).max() <= 1.01 * s_o
    assert float((out.numpy() == gd['fc2_float/q8']).mean()) > 0.9


def test_bit_config_minus_one_is_the_reference_fp_fallback(dva, micro):
    """bit_config entries of -1 (per-layer fp32 fallback: layers.py:144,171; vit_fquant.py:199,429-430; layers_quant.py:222) run the
    module graph like the reference -- never the integer engine, never an error -- and flip the block's QIntLayerNorm to float for
    good, after which even [8]*10 stays on the module graph.  Expected logits: the REAL reference (tests/golden/
    micro_vit_fp_fallback.npz, oracle/gen_golden_fp_fallback.py).  CPU: the engine is not involved."""
    from conftest import load_golden
    gf = load_golden('micro_vit_fp_fallback')
    for name in ('head', 'embed', 'proj0', 'fc2_1', 'qkv1_fc1_0'):
        m = _micro_model(dva, micro)
        bc = [int(b) for b in gf['bits/' + name]]
        with torch.no_grad():
            dva.harness.calibrate_model(m, micro['x_cal'])
            assert m._fused()
            out = m(micro['x_ev'], bc, False)[0]                      # CPU tensor + quant state: fine, -1 is not the engine's business
            flipped = np.array([[b.norm1.mode == 'ln', b.norm2.mode == 'ln'] for b in m.blocks])
            assert np.array_equal(flipped, gf['norm_modes/' + name]), name
            s_o = float(m.act_out.quantizer.scale)
            # torch fake-quant graph vs the reference's: same ops; the canonical exact sums of LN / LIS may move single codes
            assert np.abs(out.numpy() - gf['logits/' + name]).max() <= 1.01 * s_o, name
            assert float((out.numpy() == gf['logits/' + name]).mean()) > 0.9, name
            if flipped.any():
                assert not m._fused()                                 # a float norm: [8]*10 is no longer the integer pipeline
                after = m(micro['x_ev'], [8] * 10, False)[0]
                assert np.abs(after.numpy() - gf['after_q8/' + name]).max() <= 1.01 * s_o, name
            else:
                assert m._fused()

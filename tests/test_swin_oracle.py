"""CPU: the Swin-operator oracle against vectors produced by the real reference (tests/golden/swin_winattn.npz)."""
import numpy as np
import pytest
import torch

from conftest import load_golden


def _setup(synth):
    g = load_golden('swin_winattn')
    dim, heads, ws = 64, 2, 7
    W = {
        'qkv.weight': synth.normal(21, 'wa/qkv.w', (3 * dim, dim), 0.09), 'qkv.bias': synth.normal(21, 'wa/qkv.b', (3 * dim,), 0.1),
        'proj.weight': synth.normal(21, 'wa/proj.w', (dim, dim), 0.08), 'proj.bias': synth.normal(21, 'wa/proj.b', (dim,), 0.05),
        'relative_position_bias_table': synth.normal(21, 'wa/table', ((2 * ws - 1) ** 2, heads), 0.6),
    }
    c = {k: torch.from_numpy(g['scale/' + k]) for k in ('qact1', 'qact_attn1', 'qact_table', 'qact2', 'qact3', 'qact4')}
    c['qkv'] = torch.from_numpy(g['wscale/qkv/int8'])
    c['proj'] = torch.from_numpy(g['wscale/proj/int8'])
    return g, W, c, heads, ws


def test_window_attention_matches_reference(synth):
    import swin_oracle as SO
    g, W, c, heads, ws = _setup(synth)
    assert np.array_equal(SO.relative_position_index(ws).numpy(), g['rel_index'])
    x = torch.from_numpy(g['x_ev']).float()
    for tag, mask in (('nomask', None), ('mask', torch.from_numpy(g['mask']))):
        taps = {}
        q4 = SO.window_attention_quant(x, float(g['s_in']), W, c, heads, ws, mask=mask, taps=taps)
        B_, N, C = x.shape
        ref = {k[len('taps/%s/' % tag):]: v for k, v in g.items() if k.startswith('taps/%s/' % tag)}
        # reference tap layouts: qact1 [B_,N,3C]; qact_attn1/qact2/softmax_k [B_,H,N,N]; qact_table [(2ws-1)^2,H]; qact3/4 [B_,N,C]
        for name in ('qact1', 'qact_attn1', 'qact_table', 'qact2', 'softmax_k', 'qact3', 'qact4'):
            got = taps[name].numpy().astype(np.int64)
            want = ref[name].astype(np.int64).reshape(got.shape)
            assert np.array_equal(got, want), (tag, name, int((got != want).sum()), got.size)
        out = (q4 * c['qact4']).numpy()
        assert np.array_equal(out, g['out/' + tag])


def test_window_index_and_mask_match_torch_ops():
    """window_index == roll + window_partition, and scatters back like window_reverse + roll."""
    import swin_oracle as SO
    H = W = 14
    ws, shift, C, B = 7, 3, 5, 2
    x = torch.arange(B * H * W * C, dtype=torch.float32).reshape(B, H, W, C)
    sh = torch.roll(x, shifts=(-shift, -shift), dims=(1, 2))
    win = sh.reshape(B, H // ws, ws, W // ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws * ws, C)
    idx = SO.window_index(H, W, ws, shift)
    got = x.reshape(B, H * W, C)[:, idx.reshape(-1)].reshape(B * idx.shape[0], ws * ws, C)
    assert torch.equal(got, win)
    back = torch.zeros(B, H * W, C)
    back[:, idx.reshape(-1)] = win.reshape(B, -1, C)
    assert torch.equal(back.reshape(B, H, W, C), x)
    m = SO.shifted_window_mask(H, W, ws, shift)
    assert m.shape == (4, 49, 49) and set(m.unique().tolist()) == {-100.0, 0.0}
    assert (m[0] == 0).all() and (m[3] != 0).any()
    # gather order of PatchMerging
    y = SO.patch_merge_gather(x.reshape(B, H * W, C), H, W)
    assert torch.equal(y[0, 0], torch.cat([x[0, 0, 0], x[0, 1, 0], x[0, 0, 1], x[0, 1, 1]]))


def test_swin_model_oracle_equals_module_fake_quant_graph(synth):
    """OracleSwin (functional restatement) == the Swin module surface executed op by op in torch fake-quant mode on CPU: the
    surface is composed of the reference-pinned QAct/QLinear/QIntLayerNorm/QIntSoftmax classes in swin_quant.py's call order."""
    import diff_vit_amd as dva
    from diff_vit_amd import swin
    import swin_oracle as SO
    cfg = dva.Config(True, True, 'minmax')
    m = swin.swin_micro_patch4_window7_56(cfg=cfg, num_classes=10).eval()
    m.load_state_dict(synth.swin_state_dict(m.state_dict(), 5))
    x = synth.images(5, 3, 56)
    with torch.no_grad():
        fp = m(x)
        m.model_open_calibrate(); m.model_open_last_calibrate(); m(x[:2]); m.model_close_calibrate()
        m.model_quant()
        y_mod = m.act_out(m.head(m.forward_features(x)))           # the op-by-op fake-quant graph (not the product path)
        y_or = SO.OracleSwin(m.arch, m.state_dict()).quant_forward(x, m.export_calib(), 8)
    assert fp.shape == (3, 10)
    assert torch.equal(y_mod, y_or)
    with pytest.raises(RuntimeError):                              # the product path has no CPU fallback
        m(x)

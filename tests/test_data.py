"""CPU: the real-data leg: PIL port of the reference's build_transform and ImageFolder loaders (test_quant.py:100-144,
504-534).  torchvision is not installed here, so equality with its transform is PARITY UNPINNED; what is checked is the geometry
and arithmetic the reference's pipeline specifies, and that the harness evaluates an ImageFolder tree end to end."""
import os

import numpy as np
import pytest
import torch
from PIL import Image


def _make_tree(root, n_classes=3, per_class=2, size=(300, 260)):
    rng = np.random.RandomState(0)
    for split in ('val', 'train'):
        for c in range(n_classes):
            d = os.path.join(root, split, 'n%04d' % (n_classes - c))        # names sort in reverse creation order
            os.makedirs(d)
            for i in range(per_class):
                arr = rng.randint(0, 256, (size[1], size[0], 3), dtype=np.uint8)
                Image.fromarray(arr).save(os.path.join(d, 'img_%d.png' % i))
    return root


def test_build_transform_geometry_and_normalisation():
    from diff_vit_amd.data import build_transform
    arr = np.zeros((400, 600, 3), dtype=np.uint8)
    arr[:, :, 0] = 255                                   # pure red, constant: resampling cannot change it
    t = build_transform()(Image.fromarray(arr))
    assert t.shape == (3, 224, 224) and t.dtype == torch.float32
    mean, std = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
    for ch, v in enumerate((1.0, 0.0, 0.0)):
        assert torch.allclose(t[ch], torch.full((224, 224), (v - mean[ch]) / std[ch]), atol=1e-6)
    # Resize(256) keeps the aspect (shorter side 256, longer int(256 * 600 / 400) = 384), CenterCrop(224) takes the middle:
    # a vertical edge at x = 300 of the 600-wide source lands at x = 192 of the resized image = column 112 of the crop
    edge = np.zeros((400, 600, 3), dtype=np.uint8)
    edge[:, 300:] = 255
    te = build_transform(mean=(0, 0, 0), std=(1, 1, 1))(Image.fromarray(edge))
    col = te[0, 100]
    assert float(col[:108].max()) < 0.05 and float(col[116:].min()) > 0.95
    small = build_transform(input_size=32)(Image.fromarray(arr[:32, :32]))          # input_size <= 32: no resize / crop
    assert small.shape == (3, 32, 32)
    vit = build_transform(mean=(0.5, 0.5, 0.5), std=(0.5, 0.5, 0.5), crop_pct=0.9)(Image.fromarray(arr))
    assert torch.allclose(vit[0], torch.ones(224, 224)) and torch.allclose(vit[1], -torch.ones(224, 224))


def test_image_folder_order_and_loaders(tmp_path):
    from diff_vit_amd.data import ImageFolder, build_loaders
    root = _make_tree(str(tmp_path))
    ds = ImageFolder(os.path.join(root, 'val'))
    assert ds.classes == ['n0001', 'n0002', 'n0003'] and len(ds) == 6
    assert [t for _, t in ds.samples] == [0, 0, 1, 1, 2, 2]
    assert all(os.path.basename(p) == 'img_%d.png' % (i % 2) for i, (p, _) in enumerate(ds.samples))
    val, train = build_loaders(root, 'deit_tiny', 4, 2)
    xb, yb = next(iter(val))
    assert xb.shape == (4, 3, 224, 224) and yb.tolist() == [0, 0, 1, 1]
    assert len(train) == 3 and next(iter(train))[0].shape == (2, 3, 224, 224)
    with pytest.raises(FileNotFoundError):
        ImageFolder(os.path.join(root, 'val', 'n0001'))


def test_harness_on_an_image_folder(tmp_path, capsys):
    """harness.main --real-data: float DeiT-T over a 6-image ImageFolder tree (CPU; --quant needs the GPU engine)."""
    import diff_vit_amd as dva
    root = _make_tree(str(tmp_path))
    loss, top1, top5 = dva.harness.main(['--model', 'deit_tiny', '--real-data', '--data', root, '--val-batchsize', '3', '--device', 'cpu',
                                         '--print-freq', '1'])
    out = capsys.readouterr().out
    assert 'Test: [0/2]' in out and ' * Prec@1' in out
    assert 0.0 <= top1 <= 100.0 and np.isfinite(loss)


def test_pretrained_true_reads_the_torch_hub_cache(tmp_path, monkeypatch):
    """pretrained=True (vit_fquant.py:822-828, test_quant.py:95): the file torch.hub would have downloaded is loaded from
    <TORCH_HOME>/hub/checkpoints when it is there (.pth under 'model' for DeiT, Flax .npz for ViT-B); a missing file is a
    FileNotFoundError that names the path - nothing is ever fetched."""
    import diff_vit_amd as dva
    from diff_vit_amd import checkpoint as ck
    monkeypatch.setenv('TORCH_HOME', str(tmp_path))
    with pytest.raises(FileNotFoundError) as e:
        dva.deit_tiny_patch16_224(pretrained=True)
    assert 'deit_tiny_patch16_224-a1311bcf.pth' in str(e.value) and str(tmp_path) in str(e.value)
    os.makedirs(os.path.join(str(tmp_path), 'hub', 'checkpoints'))
    arch = dva.synth.ARCHS['deit_tiny']
    sd = dva.synth.vit_state_dict(arch, 5)
    torch.save({'model': sd}, ck.pretrained_path('deit_tiny_patch16_224'))
    # torch.hub's check_hash=True (what the reference's factory calls): the digest must start with the hex suffix of the file name
    with pytest.raises(RuntimeError, match='invalid hash value'):
        dva.deit_tiny_patch16_224(pretrained=True)
    import hashlib
    digest = hashlib.sha256(open(ck.pretrained_path('deit_tiny_patch16_224'), 'rb').read()).hexdigest()
    good = 'deit_tiny_patch16_224-%s.pth' % digest[:8]
    os.rename(ck.pretrained_path('deit_tiny_patch16_224'), os.path.join(str(tmp_path), 'hub', 'checkpoints', good))
    monkeypatch.setitem(ck.PRETRAINED_FILES, 'deit_tiny_patch16_224', good)
    m = dva.deit_tiny_patch16_224(pretrained=True)
    assert all(torch.equal(m.state_dict()[k], v) for k, v in sd.items())
    # the Flax layout of the ViT-B factory (models/utils.py:12-197)
    archb = dva.synth.ARCHS['vit_base']
    sdb = dva.synth.vit_state_dict(archb, 6)
    np.savez(ck.pretrained_path('vit_base_patch16_224'), **ck.state_dict_to_vit_npz(sdb, archb['depth'], archb['num_heads']))
    mb = dva.vit_base_patch16_224(pretrained=True)
    assert all(torch.equal(mb.state_dict()[k], v) for k, v in sdb.items())
    assert set(ck.PRETRAINED_FILES) >= {'deit_small_patch16_224', 'deit_base_patch16_224', 'vit_large_patch16_224', 'swin_tiny_patch4_window7_224',
                                        'swin_small_patch4_window7_224', 'swin_base_patch4_window7_224'}
    with pytest.raises(FileNotFoundError):
        dva.swin_tiny_patch4_window7_224(pretrained=True)


def test_device_prefetcher_on_cpu_is_a_pass_through():
    """harness.DevicePrefetcher on a CPU device: the loader's batches, in order (the copy stream exists on a GPU only)."""
    import diff_vit_amd as dva
    batches = [(torch.full((2, 3), float(i)), torch.tensor([i, i + 1])) for i in range(3)]
    out = list(dva.harness.DevicePrefetcher(batches, 'cpu'))
    assert len(out) == 3 and len(dva.harness.DevicePrefetcher(batches, 'cpu')) == 3
    for (x, t), (gx, gt) in zip(batches, out):
        assert torch.equal(x, gx) and torch.equal(t, gt) and gx.device.type == 'cpu'

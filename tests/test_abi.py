"""CPU: the C-ABI library loads and exports every symbol include/p2vit.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def _declared():
    src = open(os.path.join(ROOT, 'include', 'p2vit.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(p2v_[a-z0-9_]+)\s*\(', src)))


def test_header_symbols_exported():
    import diff_vit_amd
    if not diff_vit_amd.engine.available():
        pytest.fail('libp2vit_hip.so is not built: run __graft_entry__.build()')
    lib = ctypes.CDLL(diff_vit_amd.engine.LIB_PATH)
    names = _declared()
    assert len(names) >= 17
    for n in names:
        assert hasattr(lib, n), n
    assert diff_vit_amd.engine.lib().p2v_abi_version() == diff_vit_amd.engine.P2V_ABI_VERSION


def test_error_conventions_without_gpu():
    """argument validation happens before any HIP call, so it is checkable on CPU."""
    import diff_vit_amd
    E = diff_vit_amd.engine
    L = E.lib()
    h = ctypes.c_void_p()
    bad = E.ModelDesc(E.P2V_ABI_VERSION, 224, 15, 3, 384, 12, 6, 1536, 1000)
    with pytest.raises(AssertionError):
        E.check(L.p2v_plan_create(ctypes.byref(bad), ctypes.byref(h)))
    ok = E.ModelDesc(E.P2V_ABI_VERSION, 224, 16, 3, 384, 12, 6, 1536, 1000)
    E.check(L.p2v_plan_create(ctypes.byref(ok), ctypes.byref(h)))
    assert L.p2v_workspace_bytes(h, 256) > 200e6
    lin = E.Linear(None, None, None)
    with pytest.raises(ValueError):
        E.check(L.p2v_plan_set_linear(h, 0, 6, ctypes.byref(lin)))     # 6-bit: not in bit_pool
    cfg = (ctypes.c_int8 * 50)(*([8] * 50))
    rc = L.p2v_forward(h, ctypes.c_void_p(1), 1, cfg, 49, ctypes.c_void_p(1), ctypes.c_void_p(1), 0, -1, None)
    assert rc == E.E_BITS                                                 # wrong bit_config length
    L.p2v_plan_destroy(h)


def test_plan_refuses_out_of_range_constants_at_plan_time():
    """p2v_forward launches the kernels directly, so what the per-operator entry points check (softmax constants inside the exact
    range, power-of-two REQUANT scale, GELU table size) is checked by the plan setters; the Python freeze refuses even earlier."""
    import torch
    import diff_vit_amd
    from diff_vit_amd import plan as P
    E = diff_vit_amd.engine
    L = E.lib()
    assert P.lis_consts(torch.tensor(2.0 ** -11)) == (-1420, 5544, 11710978)
    with pytest.raises(NotImplementedError):
        P.lis_consts(torch.tensor(2.0 ** -12))                              # c_int = 46.8 M >= 2^24: z leaves the exact fp32 range
    h = ctypes.c_void_p()
    ok = E.ModelDesc(E.P2V_ABI_VERSION, 224, 16, 3, 384, 12, 6, 1536, 1000)
    E.check(L.p2v_plan_create(ctypes.byref(ok), ctypes.byref(h)))
    one = ctypes.c_void_p(16)

    def block(s_attn=2.0 ** -4, inv_qkv=16.0, cells=300, inv_fc1=8.0):
        b = E.Block()
        ln = E.Ln(1.0, one, one, one, one, one, None)
        for i in range(2):
            b.ln1[i] = ln
            b.inv_s_qkv[i] = inv_qkv
            for j in range(2):
                b.ln2[i][j] = ln
        x0, bb, cc = [int(v) for v in (torch.floor(-0.6931 / torch.tensor(s_attn)), torch.floor((0.96963238 / 0.35815147) / torch.tensor(s_attn)),
                                       torch.floor((1. / 0.35815147) / torch.tensor(s_attn) ** 2))]
        b.attn = E.Attn(2.0 ** -8, 0.125, 1.0 / s_attn, 0.5, x0, bb, cc)
        for e in (b.proj_epi, b.fc2_epi):
            e.s_mid, e.s_res, e.s_next = one, one, one
        b.inv_s_fc1 = inv_fc1
        b.gelu_fc1 = E.GeluTab(one, 16.0, 100.0, cells)
        return b

    E.check(L.p2v_plan_set_block(h, 0, ctypes.byref(block())))
    with pytest.raises(NotImplementedError):                                # qact_attn1 scale 2^-12: refused when the plan is built
        E.check(L.p2v_plan_set_block(h, 0, ctypes.byref(block(s_attn=2.0 ** -12))))
    assert b'log-int-softmax constants out of range' in L.p2v_last_error()
    with pytest.raises(NotImplementedError):                                # REQUANT folds 1/scale into the column constants
        E.check(L.p2v_plan_set_block(h, 0, ctypes.byref(block(inv_qkv=12.0))))
    with pytest.raises(E.P2VError):                                         # table larger than the kernels' LDS budget
        E.check(L.p2v_plan_set_block(h, 0, ctypes.byref(block(cells=5000))))
    with pytest.raises(NotImplementedError):
        E.check(L.p2v_plan_set_block(h, 0, ctypes.byref(block(inv_fc1=3.0))))
    L.p2v_plan_destroy(h)


def test_module_cache_follows_replaced_submodules():
    """state switches walk the module tree like the reference (vit_fquant.py:667-698): a replaced head is not skipped."""
    import diff_vit_amd as dva
    a = dva.synth.ARCHS['micro']
    from functools import partial
    m = dva.VisionTransformer(img_size=a['img_size'], patch_size=a['patch_size'], embed_dim=a['embed_dim'], depth=a['depth'],
                              num_heads=a['num_heads'], num_classes=a['num_classes'], mlp_ratio=a['mlp_ratio'], qkv_bias=True,
                              norm_layer=partial(dva.QIntLayerNorm, eps=1e-6), input_quant=True, cfg=dva.Config())
    m.model_quant()
    old = m.head
    m.head = dva.QLinear(a['embed_dim'], 7)
    assert m._qmods is None and m._plan is None
    m.model_quant()
    assert m.head.quant and m.head in m._q_modules() and old not in m._q_modules()
    m.blocks[0].attn.proj = dva.QLinear(a['embed_dim'], a['embed_dim'])     # a nested replacement: caught by the next state switch
    m.model_open_calibrate()
    assert m.blocks[0].attn.proj.calibrate


def test_head_dims_and_token_limits():
    """what p2v_plan_create accepts: head_dim 32 / 48 / 64 / 80 / 96 / 128, up to 4096 tokens per image (the resident attention kernel covers 608 / 544 / 384
    of them, the streaming kernel the rest)."""
    import diff_vit_amd
    E = diff_vit_amd.engine
    L = E.lib()
    assert [L.p2v_resident_tokens(h) for h in (32, 48, 64, 80, 96, 128, 16, 112, 160)] == [608, 608, 608, 608, 544, 384, 0, 0, 0]
    assert [L.p2v_max_tokens(h) for h in (32, 48, 64, 80, 96, 128, 16, 112, 160)] == [4096] * 6 + [0, 0, 0]
    import ctypes
    for dim, heads, img, ok in ((256, 2, 224, True), (256, 2, 320, True), (192, 2, 224, True), (192, 2, 384, True), (320, 4, 384, True), (128, 2, 1040, False),
                                (192, 4, 384, True), (224 * 2, 4, 224, False), (64, 4, 224, False), (160, 2, 224, True), (144, 3, 224, True), (200, 5, 224, False)):
        h = ctypes.c_void_p()
        d = E.ModelDesc(E.P2V_ABI_VERSION, img, 16, 3, dim, 1, heads, 4 * dim, 10)
        rc = L.p2v_plan_create(ctypes.byref(d), ctypes.byref(h))
        assert (rc == 0) == ok, (dim, heads, img, rc, L.p2v_last_error())
        if rc == 0:
            L.p2v_plan_destroy(h)


def test_tuning_switches():
    """p2v_set_tuning: known switches with values in range are accepted, everything else is an argument error."""
    import diff_vit_amd
    E = diff_vit_amd.engine
    L = E.lib()
    for name, good, bad in ((b'ln_gemm_version', 3, 4), (b'ln_rows', 4, 0), (b'attn_waves', 8, 9), (b'gemm_tile', 128, 64)):
        assert L.p2v_set_tuning(name, good) == 0
        assert L.p2v_set_tuning(name, bad) == E.E_ARG
    assert L.p2v_set_tuning(b'gemm_tile', 0) == 0 and L.p2v_set_tuning(b'ln_gemm_version', 2) == 0 and L.p2v_set_tuning(b'ln_gemm', 1) == 0 and L.p2v_set_tuning(b'ln_generic', 0) == 0
    assert L.p2v_set_tuning(b'no_such_switch', 1) == E.E_ARG and L.p2v_set_tuning(None, 1) == E.E_ARG
    assert b'unknown switch' in L.p2v_last_error() or b'null name' in L.p2v_last_error()


def test_custom_ops_registered_and_gpu_only():
    """torch.ops.p2vit.* exist after import and have no CPU kernel (no silent fallback)."""
    import torch
    import diff_vit_amd as dva
    for name in dva.ops.OPS:
        assert hasattr(torch.ops.p2vit, name), name
    with pytest.raises(NotImplementedError):
        torch.ops.p2vit.fake_quant(torch.zeros(4), torch.ones(1), 1, -128, 127)
    with pytest.raises(NotImplementedError):
        torch.ops.p2vit.int_layernorm(torch.zeros(2, 64, dtype=torch.int8), 1.0, *[torch.ones(64)] * 5)


def test_run_ops_and_swin_entry_validation_without_gpu():
    """p2v_run_ops / window attention / merge / avgpool reject bad records before any HIP call."""
    import diff_vit_amd
    E = diff_vit_amd.engine
    L = E.lib()
    assert L.p2v_run_ops((E.Op * 1)(), 0, None) == 0                        # empty sequence
    op = E.Op()
    op.kind = 99
    with pytest.raises(E.P2VError):
        E.check(L.p2v_run_ops((E.Op * 1)(op), 1, None))
    assert b'unknown op kind' in L.p2v_last_error()
    one = ctypes.c_void_p(16)
    wa = E.WinAttn(2.0 ** -4, 0.1767767, 2.0 ** -3, 2.0 ** -5, 2.0 ** -4, 2.0 ** -3, -12, 43, 714, one, one, None, 7, 4)
    with pytest.raises(NotImplementedError):                                # head_dim 64: only 32 is instantiated
        E.check(L.p2v_window_attention(one, 1, 196, 4, 64, ctypes.byref(wa), one, None, None))
    wa.s_q2 = 0.3                                                           # not a power of two
    with pytest.raises(NotImplementedError):
        E.check(L.p2v_window_attention(one, 1, 196, 4, 32, ctypes.byref(wa), one, None, None))
    with pytest.raises(AssertionError):                                     # odd feature map
        E.check(L.p2v_patch_merge_gather(one, 1, 7, 7, 64, one, None))
    with pytest.raises(NotImplementedError):
        E.check(L.p2v_avgpool_quant(one, 1, 49, 6, 1.0, 1.0, one, None))


def test_int4_tile_packing_layout():
    """engine.pack_int4_tiles against a literal reading of the layout in include/p2vit.h (p2v_linear.packed4)."""
    import numpy as np
    import torch
    import diff_vit_amd as dva
    g = torch.Generator().manual_seed(3)
    w = torch.randint(-8, 8, (256, 128), generator=g, dtype=torch.int8)
    p = dva.engine.pack_int4_tiles(w).numpy()
    assert p.shape == (2, 2, 128, 32) and p.dtype == np.uint8
    wn = w.numpy().astype(np.int64)
    for (t, kt, r, c, j) in ((0, 0, 0, 0, 0), (1, 1, 127, 3, 7), (0, 1, 9, 2, 5), (1, 0, 77, 1, 3), (0, 0, 24, 0, 6)):
        chunk = p[t, kt, r, (c ^ ((r >> 3) & 3)) * 8:(c ^ ((r >> 3) & 3)) * 8 + 8]
        k0 = kt * 64 + c * 16
        lo_k, hi_k = (k0 + j, k0 + 4 + j) if j < 4 else (k0 + 8 + (j - 4), k0 + 12 + (j - 4))
        row = t * 128 + r
        assert chunk[j] == ((wn[row, lo_k] & 15) | ((wn[row, hi_k] & 15) << 4)), (t, kt, r, c, j)
    # the kernel's widening: (byte << 4) & 0xF0 / byte & 0xF0 as int8 = 16 x code
    b = p[0, 0, 5, :8].astype(np.uint8)
    even = ((b.astype(np.uint16) << 4) & 0xF0).astype(np.uint8).view(np.int8)
    odd = (b & 0xF0).view(np.int8)
    c0 = 0 ^ ((5 >> 3) & 3)
    assert c0 == 0
    assert np.array_equal(even[:4], 16 * wn[5, 0:4]) and np.array_equal(odd[:4], 16 * wn[5, 4:8])
    assert np.array_equal(even[4:], 16 * wn[5, 8:12]) and np.array_equal(odd[4:], 16 * wn[5, 12:16])
    # the packed FRAGMENT-order copy of the fused LayerNorm+GEMM kernels (p2v_linear.w_frag with packed4 = 1, ABI 4)
    f = dva.engine.fragment_order_packed4(w).numpy()
    assert f.shape == (2, 4, 4, 2, 32, 8) and f.dtype == np.uint8
    for (t, wave, ks, h, r, j) in ((0, 0, 0, 0, 0, 0), (1, 3, 3, 1, 31, 7), (0, 2, 1, 1, 9, 5), (1, 1, 2, 0, 17, 3)):
        row, k0 = 128 * t + 32 * wave + r, 32 * ks + 16 * h
        lo_k, hi_k = (k0 + j, k0 + 4 + j) if j < 4 else (k0 + 8 + (j - 4), k0 + 12 + (j - 4))
        assert f[t, wave, ks, h, r, j] == ((wn[row, lo_k] & 15) | ((wn[row, hi_k] & 15) << 4)), (t, wave, ks, h, r, j)


def test_late_import_warns_about_hardware_queues():
    """GPU_MAX_HW_QUEUES=8 is only a default the package can set BEFORE the HIP runtime initialises: an import that comes too late (HIP up,
    variable unset) warns and records `in_effect: False`; a user setting or an early import does not."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k != 'GPU_MAX_HW_QUEUES'}
    late = ("import sys, warnings; sys.path.insert(0, %r); import torch; torch.cuda.is_initialized = lambda: True\n"
            "with warnings.catch_warnings(record=True) as w:\n"
            "    warnings.simplefilter('always'); import diff_vit_amd as d\n"
            "print(int(any('GPU_MAX_HW_QUEUES' in str(x.message) for x in w)), int(d.HW_QUEUES['in_effect']), d.HW_QUEUES['set_by'])\n" % root)
    r = subprocess.run([sys.executable, '-c', late], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0 and r.stdout.split()[:2] == ['1', '0'], r.stdout + r.stderr
    r = subprocess.run([sys.executable, '-c', late], capture_output=True, text=True, env=dict(env, GPU_MAX_HW_QUEUES='6'), timeout=300)
    assert r.returncode == 0 and r.stdout.split()[:3] == ['0', '1', 'environment'], r.stdout + r.stderr
    early = "import sys; sys.path.insert(0, %r); import diff_vit_amd as d; print(int(d.HW_QUEUES['in_effect']), d.HW_QUEUES['value'])" % root
    r = subprocess.run([sys.executable, '-c', early], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0 and r.stdout.split() == ['1', '8'], r.stdout + r.stderr

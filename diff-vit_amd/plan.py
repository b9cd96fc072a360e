"""Freeze a calibrated PoT-PTQ ViT into the integer plan the HIP engine executes.

The reference keeps its calibrated state in Python attributes (``quantizer.scale``, ``dic_scale``,
``best_scale/best_act_scale/best_weight_scale`` -- models/vit_fquant.py:234-239,273-278) and re-derives
everything on every forward: weights are fake-quantised per call (models/ptq/layers.py:177,
quantizer/uniform.py:82-88), LN multipliers and softmax constants are recomputed per call
(layers.py:255-289,334-351).  Freezing does that work once, with the same fp32 operations, and uploads
the result; ``forward`` then is one C call (``p2v_forward``).

Input is a *calibration dict* (nested: name -> tensor | {bit_name: tensor} | [per-bit-pool-index ...]),
the format ``export_calib`` produces from the module surface; plus the fp32 ``state_dict``.
"""
import ctypes as C
import warnings
import weakref

import numpy as np
import torch

from . import engine as E

BIT_POOL = (4, 8)                                # models/vit_fquant.py:33
BOUNDS = {4: (-8, 7), 8: (-128, 127)}            # models/ptq/bit_type.py:17-27 for int4 / int8
GEMM_N_PAD, GEMM_K_PAD = 128, 64


def _is_pot(t):
    m, _ = torch.frexp(t.detach().float().reshape(-1))
    return bool(torch.all(m == 0.5))


def _need_pot(name, t):
    if not _is_pot(t):
        # the kernels replace x / s by exact multiplications with 1/s; only valid for PoT scales
        raise NotImplementedError('%s is not a power of two (observer other than the PoT minmax?)' % name)
    return t.detach().float()


def _q8(v, s):
    return torch.clamp(torch.round(v / s), -128, 127)


def lis_consts(sf):
    """x0_int, b_int, c_int with the reference's fp32 expressions (models/ptq/layers.py:334-351)."""
    sf = sf.detach().float().reshape(())
    x0 = torch.floor(-0.6931 / sf)
    b = torch.floor((0.96963238 / 0.35815147) / sf)
    c = torch.floor((1. / 0.35815147) / sf**2)
    x0, b, c = int(x0), int(b), int(c)
    # the kernels keep z = r (r + b) + c exactly in fp32 and the exp table in int64 (include/p2vit.h, Limits): refuse at plan time
    if not (-(1 << 12) <= x0 < 0 and 0 <= b < (1 << 23) and 0 < c < (1 << 24)):
        raise NotImplementedError('log-int-softmax input scale %g is outside the exact range of the attention kernel '
                                  '(x0_int %d, b_int %d, c_int %d; scales from 2^-11 up are covered)' % (float(sf), x0, b, c))
    return x0, b, c


class FrozenPlan:
    """Device-resident integer plan + handle of the C plan object."""

    def __init__(self, arch, state_dict, calib, device='cuda', in_chans=3, input_quant=True):
        self.arch = dict(arch)
        self.input_quant = bool(input_quant)      # False: the fp32 image feeds the patch-embed convolution (vit_fquant.py:705, :925)
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise RuntimeError('FrozenPlan needs a GPU device: the quantized forward runs on the HIP engine only (no CPU path)')
        if self.device.index is None:
            self.device = torch.device('cuda', torch.cuda.current_device())
        self.in_chans = in_chans
        self._keep = []          # tensors whose device pointers the C plan borrows
        self._handle = C.c_void_p()
        self._ws = None
        self._ws_batch = 0
        a = self.arch
        self.D, self.depth, self.H = a['embed_dim'], a['depth'], a['num_heads']
        self.hidden = int(self.D * a['mlp_ratio'])
        self.patches = (a['img_size'] // a['patch_size']) ** 2
        self.tokens = self.patches + 1
        self.n_layers = 4 * self.depth + 2
        L = E.lib()
        desc = E.ModelDesc(E.P2V_ABI_VERSION, a['img_size'], a['patch_size'], in_chans, self.D, self.depth, self.H,
                           self.hidden, a['num_classes'])
        E.check(L.p2v_plan_create(C.byref(desc), C.byref(self._handle)))
        W = {k: v.detach().float().cpu() for k, v in state_dict.items()}
        # every upload and the plan-time fold of p2v_plan_set_block (a hipMalloc + copies) happen with the plan's device current,
        # whichever device the caller has selected (a plan for cuda:1 built while cuda:0 is current)
        with torch.cuda.device(self.device):
            self._build(W, calib)
        from . import ops                      # torch.ops.p2vit.forward(handle, images, bit_config)
        ops._PLANS[self.handle] = weakref.proxy(self)

    @property
    def handle(self):
        """integer value of the C plan pointer (the ``plan`` argument of ``torch.ops.p2vit.forward``)."""
        return int(self._handle.value or 0)

    # ---------------------------------------------------------------------------------------------
    def _dev(self, t, dtype=torch.float32):
        t = t.detach().to(dtype).contiguous().to(self.device)
        self._keep.append(t)
        return t

    def _linear(self, layer, w, cs, s_x, dic, bias, pack4=True):
        """QLinear/QConv2d weight for both bit widths (layers.py:173-178; uniform.py:82-88).  ``pack4`` False keeps 4-bit codes one
        per byte (the fp32-input patch embedding reads unpacked codes)."""
        L = E.lib()
        w2 = w.reshape(w.shape[0], -1)
        if cs is not None:
            w2 = w2 * cs.reshape((1, -1))                    # weight_smoothed, vit_fquant.py:285
        N, K = w2.shape
        n_pad = (N + GEMM_N_PAD - 1) // GEMM_N_PAD * GEMM_N_PAD
        k_pad = (K + GEMM_K_PAD - 1) // GEMM_K_PAD * GEMM_K_PAD
        for bits in BIT_POOL:
            name = 'int%d' % bits
            s_w = _need_pot('%d.%s weight scale' % (layer, name), dic[name]).reshape(-1)
            lo, hi = BOUNDS[bits]
            codes = torch.clamp(torch.round(w2 / s_w.reshape(-1, 1)), lo, hi)
            wp = torch.zeros(n_pad, k_pad, dtype=torch.int8)
            wp[:N, :K] = codes.to(torch.int8)
            cs_col = torch.zeros(n_pad)
            cs_col[:N] = (s_x.reshape(-1) * s_w).expand(N)
            bp = torch.zeros(n_pad)
            if bias is not None:
                bp[:N] = bias
            # 4-bit weights travel packed, two codes per byte (half the HBM bytes of BASELINE config 5), as LDS tile images
            packed = bits == 4 and pack4
            wdev = self._dev(E.pack_int4_tiles(wp), torch.uint8) if packed else self._dev(wp, torch.int8)
            lin = E.Linear(E.ptr(wdev), E.ptr(self._dev(cs_col)), E.ptr(self._dev(bp)), None, 1 if packed else 0)
            E.check(L.p2v_plan_set_linear(self._handle, layer, bits, C.byref(lin)))

    def _ln(self, in_scale, gamma, beta, out_scale, post_mul):
        """Constants of QIntLayerNorm mode 'int' (layers.py:255-289)."""
        in_scale = in_scale.reshape(-1).float()
        D = gamma.numel()
        s1 = in_scale.min()
        mask = torch.round(in_scale / s1).expand(D) if in_scale.numel() == 1 else torch.round(in_scale / s1)
        # the kernel keeps sum(x_q^2) exactly in 32 unsigned bits: C * (128 * mask)^2 < 2^32 (PTF masks are 1, 2, 4 or 8: ptf.py:96-134)
        if float(mask.min()) < 1 or D * (128.0 * float(mask.max())) ** 2 >= 2.0 ** 32:
            raise NotImplementedError('LayerNorm input scale ratios up to %g over %d channels exceed the exact 32-bit statistics' % (float(mask.max()), D))
        out_scale = _need_pot('LN out scale', out_scale.reshape(-1).expand(D).contiguous())
        post_mul = _need_pot('LN post multiplier', post_mul.reshape(-1).expand(D).contiguous())
        return E.Ln(float(s1), E.ptr(self._dev(mask)), E.ptr(self._dev(gamma)), E.ptr(self._dev(beta)),
                    E.ptr(self._dev(1.0 / out_scale)), E.ptr(self._dev(post_mul)))

    def _build(self, W, c):
        L = E.lib()
        a, D = self.arch, self.D
        hd = D // self.H
        # ---- stem ------------------------------------------------------------------------------
        if self.input_quant:
            s_in = _need_pot('qact_input', c['qact_input'])
            self._linear(0, W['patch_embed.proj.weight'], None, s_in, c['patch_embed.proj'], W['patch_embed.proj.bias'])
        else:      # no input QAct: colscale = s_w, unpacked codes; p2v_plan_set_embed(inv_s_input = 0) selects the fp32-image kernel
            s_in = torch.ones(1)
            self._linear(0, W['patch_embed.proj.weight'], None, s_in, c['patch_embed.proj'], W['patch_embed.proj.bias'], pack4=False)
        s_pe = _need_pot('patch_embed.qact', c['patch_embed.qact'])
        s_e = _need_pot('qact_embed', c['qact_embed'])
        s_p = _need_pot('qact_pos', c['qact_pos'])
        s_res = c['qact1'].detach().float().reshape(-1)                       # PTF: arbitrary fp32
        pos_deq = (_q8(W['pos_embed'], s_p) * s_p).reshape(self.tokens, D)
        cls_v = _q8(W['cls_token'].reshape(1, D), s_e) * s_e + pos_deq[0:1]   # vit_fquant.py:718-725
        cls_codes = _q8(cls_v, s_res.reshape(1, -1)).reshape(D)
        epi = E.Epilogue()
        epi.inv_s_pe = float(1.0 / s_pe)
        epi.pe_to_embed = float(s_pe / s_e)
        epi.s_embed = float(s_e)
        epi.s_next = E.ptr(self._dev(s_res))
        epi.pos_deq = E.ptr(self._dev(pos_deq))
        epi.patches = self.patches
        E.check(L.p2v_plan_set_embed(self._handle, float(1.0 / s_in) if self.input_quant else 0.0, C.byref(epi),
                                     E.ptr(self._dev(cls_codes, torch.int8))))
        # ---- blocks ----------------------------------------------------------------------------
        for i in range(self.depth):
            p = 'blocks.%d.' % i
            blk = E.Block()
            cs_a, s_a0 = c[p + 'attn.best_scale'], c[p + 'attn.best_act_scale']
            cs_m, s_m0 = c[p + 'mlp.best_scale'], c[p + 'mlp.best_act_scale']
            g1, b1 = W[p + 'norm1.weight'], W[p + 'norm1.bias']
            g2, b2 = W[p + 'norm2.weight'], W[p + 'norm2.bias']
            s_b2 = c[p + 'qact2'].detach().float().reshape(-1)
            s_b4 = c[p + 'qact4'].detach().float().reshape(-1)
            s_q1 = _need_pot(p + 'attn.qact1', c[p + 'attn.qact1'])
            s_at = _need_pot(p + 'attn.qact_attn1', c[p + 'attn.qact_attn1'])
            s_a2 = _need_pot(p + 'attn.qact2', c[p + 'attn.qact2'])
            s_m1 = _need_pot(p + 'mlp.qact1', c[p + 'mlp.qact1'])
            for bi in range(2):
                csb = _need_pot(p + 'attn.channel_scale', cs_a[bi])
                sab = _need_pot(p + 'attn.qact0', s_a0[bi])
                out_scale = sab * csb                                          # layers.py:264-265
                blk.ln1[bi] = self._ln(s_res, g1, b1, out_scale, out_scale / csb / sab)
                blk.inv_s_qkv[bi] = float(1.0 / s_q1)
                for bm in range(2):
                    csm = _need_pot(p + 'mlp.channel_scale', cs_m[bm])
                    smb = _need_pot(p + 'mlp.qact0', s_m0[bm])
                    o2 = smb * csb                                             # attention's scale: vit_fquant.py:464
                    blk.ln2[bi][bm] = self._ln(s_b2, g2, b2, o2, o2 / csm / smb)
            # qkv / fc1 depend on the bit-pool index through best_* (identical when |alpha_pool| == 1)
            self._linear_per_bit(1 + 4 * i, W[p + 'attn.qkv.weight'], cs_a, s_a0, c[p + 'attn.best_weight_scale'], W[p + 'attn.qkv.bias'])
            x0, bb, cc = lis_consts(s_at)
            blk.attn = E.Attn(float(s_q1 * s_q1), float(np.float32(hd ** -0.5)), float(1.0 / s_at), float(s_q1 / s_a2), x0, bb, cc)
            self._linear(2 + 4 * i, W[p + 'attn.proj.weight'], None, s_a2, c[p + 'attn.proj'], W[p + 'attn.proj.bias'])
            pe = E.Epilogue()
            pe.s_mid = E.ptr(self._dev(c[p + 'attn.qact3'].reshape(-1)))
            pe.s_res = E.ptr(self._dev(s_res))
            pe.s_next = E.ptr(self._dev(s_b2))
            blk.proj_epi = pe
            self._linear_per_bit(3 + 4 * i, W[p + 'mlp.fc1.weight'], cs_m, s_m0, c[p + 'mlp.best_weight_scale'], W[p + 'mlp.fc1.bias'])
            blk.inv_s_fc1 = float(1.0 / s_m1)
            blk.gelu_fc1 = E.gelu_table(float(1.0 / s_m1), self.device)      # exact GELU -> qact1 threshold table (cached per scale)
            self._linear(4 + 4 * i, W[p + 'mlp.fc2.weight'], None, s_m1, c[p + 'mlp.fc2'], W[p + 'mlp.fc2.bias'])
            fe = E.Epilogue()
            fe.s_mid = E.ptr(self._dev(c[p + 'mlp.qact2'].reshape(-1)))
            fe.s_res = E.ptr(self._dev(s_b2))
            fe.s_next = E.ptr(self._dev(s_b4))
            blk.fc2_epi = fe
            E.check(L.p2v_plan_set_block(self._handle, i, C.byref(blk)))
            if L.p2v_plan_block_prefolded(self._handle, i) != 1:        # same results either way: the kernels then fold per workgroup
                warnings.warn('block %d: LayerNorm constants not folded at plan time: %s' % (i, (L.p2v_last_error() or b'').decode()))
            s_res = s_b4
        # ---- head ------------------------------------------------------------------------------
        s_f = _need_pot('qact2', c['qact2'])
        s_o = _need_pot('act_out', c['act_out'])
        fl = self._ln(s_res, W['norm.weight'], W['norm.bias'], s_f, s_f / s_f)
        E.check(L.p2v_plan_set_head(self._handle, C.byref(fl), float(1.0 / s_o), float(s_o)))
        self._linear(self.n_layers - 1, W['head.weight'], None, s_f, c['head'], W['head.bias'])
        self.s_out = float(s_o)

    def _linear_per_bit(self, layer, w, cs_list, s_x_list, dic_list, bias):
        """qkv / fc1: SmoothQuant state is looked up per bit-pool index (vit_fquant.py:282-292)."""
        L = E.lib()
        N, K = w.shape
        n_pad = (N + GEMM_N_PAD - 1) // GEMM_N_PAD * GEMM_N_PAD
        k_pad = (K + GEMM_K_PAD - 1) // GEMM_K_PAD * GEMM_K_PAD
        for bi, bits in enumerate(BIT_POOL):
            name = 'int%d' % bits
            w2 = w * cs_list[bi].reshape((1, -1))
            s_w = _need_pot('%d.%s weight scale' % (layer, name), dic_list[bi][name]).reshape(-1)
            s_x = _need_pot('%d act scale' % layer, s_x_list[bi]).reshape(-1)
            lo, hi = BOUNDS[bits]
            wp = torch.zeros(n_pad, k_pad, dtype=torch.int8)
            wp[:N, :K] = torch.clamp(torch.round(w2 / s_w.reshape(-1, 1)), lo, hi).to(torch.int8)
            cs_col = torch.zeros(n_pad)
            cs_col[:N] = (s_x * s_w).expand(N)
            bp = torch.zeros(n_pad)
            bp[:N] = bias
            # qkv / fc1 follow a LayerNorm: a second copy in MFMA-fragment order feeds the fused LayerNorm+GEMM kernel
            wdev = self._dev(E.pack_int4_tiles(wp), torch.uint8) if bits == 4 else self._dev(wp, torch.int8)
            frag = None
            if self.D <= 384:        # 4-bit layers: the fragment copy travels packed too (two codes per byte, ABI 4)
                frag = E.ptr(self._dev(E.fragment_order_packed4(wp), torch.uint8) if bits == 4 else self._dev(E.fragment_order(wp), torch.int8))
            lin = E.Linear(E.ptr(wdev), E.ptr(self._dev(cs_col)), E.ptr(self._dev(bp)), frag, 1 if bits == 4 else 0)
            E.check(L.p2v_plan_set_linear(self._handle, layer, bits, C.byref(lin)))

    # ---------------------------------------------------------------------------------------------
    def workspace(self, batch):
        if self._ws is None or self._ws_batch < batch:
            n = E.lib().p2v_workspace_bytes(self._handle, batch)
            self._ws = torch.empty(n, dtype=torch.uint8, device=self.device)
            self._ws_batch = batch
        return self._ws

    def _check(self, images, bit_config):
        """the reference's input checks (layers_quant.py:437-439, vit_fquant.py:282) plus what the C ABI trusts: the geometry in
        the plan descriptor, a device pointer on THIS plan's GPU.  Shared by ``forward``, ``forward_streams`` and ``profile``
        (a wrong-sized or foreign tensor would otherwise be an out-of-bounds read in ``k_quantize_patchify``)."""
        a = self.arch
        if images.dim() != 4 or images.shape[2] != a['img_size'] or images.shape[3] != a['img_size']:
            raise AssertionError("Input image size (%d*%d) doesn't match model (%d*%d)." % (
                images.shape[2] if images.dim() > 2 else -1, images.shape[3] if images.dim() > 3 else -1, a['img_size'], a['img_size']))
        if images.shape[1] != self.in_chans:
            raise AssertionError('expected %d input channels' % self.in_chans)
        if not images.is_cuda:
            raise RuntimeError('the quantized forward runs on the HIP engine only: move the input to the GPU')
        if images.device != self.device:
            raise RuntimeError('images live on %s, the frozen plan on %s' % (images.device, self.device))
        if images.shape[0] < 1:
            raise AssertionError('empty batch')
        if bit_config is None:
            raise ValueError('None is not in list')        # bit_pool.index(None), vit_fquant.py:282
        cfg = (C.c_int8 * len(bit_config))(*[int(b) if -128 <= int(b) <= 127 else 127 for b in bit_config])
        return images.contiguous().float(), cfg

    def _check_out(self, out, B):
        # the C ABI writes B x classes floats through the raw pointer
        if tuple(out.shape) != (B, self.arch['num_classes']) or out.device != self.device or out.dtype != torch.float32 or not out.is_contiguous():
            raise AssertionError('out must be a contiguous fp32 [%d, %d] tensor on %s' % (B, self.arch['num_classes'], self.device))

    def forward(self, images, bit_config, stop_after=-1, out=None, taps=None):
        """images: fp32 [B,C,H,W] on the plan's device -> fp32 logits [B, classes] (int8 grid * act_out scale).
        ``taps`` (dict): filled with 'qkv_output' / 'fc1_output' -> list of fp32 [B, tokens, 3D] / [B, tokens, hidden] tensors per
        block, the layer outputs before qact1 / GELU that the reference keeps for its analysis scripts (vit_fquant.py:301,
        layers_quant.py:326); written by the GEMM epilogues of the same launches."""
        images, cfg = self._check(images, bit_config)
        B = images.shape[0]
        with torch.cuda.device(self.device):
            ws = self.workspace(B)
            if out is None:
                out = torch.empty(B, self.arch['num_classes'], dtype=torch.float32, device=self.device)
            else:
                self._check_out(out, B)
            if taps is None:
                E.check(E.lib().p2v_forward(self._handle, E.ptr(images), B, cfg, len(bit_config), E.ptr(out), E.ptr(ws),
                                            ws.numel(), stop_after, E.stream_ptr(self.device)))
            else:
                qkv = [torch.empty(B, self.tokens, 3 * self.D, dtype=torch.float32, device=self.device) for _ in range(self.depth)]
                fc1 = [torch.empty(B, self.tokens, self.hidden, dtype=torch.float32, device=self.device) for _ in range(self.depth)]
                pq = (C.c_void_p * self.depth)(*[t.data_ptr() for t in qkv])
                pf = (C.c_void_p * self.depth)(*[t.data_ptr() for t in fc1])
                E.check(E.lib().p2v_forward_taps(self._handle, E.ptr(images), B, cfg, len(bit_config), E.ptr(out), E.ptr(ws), ws.numel(),
                                                 pq, pf, E.stream_ptr(self.device)))
                taps['qkv_output'], taps['fc1_output'] = qkv, fc1
        return out

    def slice_sizes(self, batch, n_streams=3):
        """Batch slices of ``forward_streams`` for up to ``n_streams`` side streams plus the caller's own stream (256 -> 68 + 68 + 68 + 52).
        Four kernels in flight on four hardware queues is what the device sustains - a fourth SIDE stream (five streams with the caller's)
        collapses to 57 k img/s, see ``engine.side_streams`` - and the caller's stream carries ~0.77 of a side slice.

        How many slices pays depends on the work per slice (profiles/r04_sweep_slices.txt, same-call sweeps over widths 192 / 384 / 768,
        197 / 577 tokens and batches of 16 ... 512): for the models whose blocks run the fused LayerNorm+GEMM kernels (widths up to 384) one
        slice per ~8 k rows (batch x tokens) at width 384 - proportionally more rows for narrower models - and at least 32 images per slice
        (the attention launch has images x heads workgroups): DeiT-S 32 images -> one slice, 64 -> two (+4 %), 128 -> three (+15 %),
        192 and more -> four (+20 %); 64 images at 577 tokens -> two (+29 %).  The MFMA-bound wide models gain from exactly TWO balanced slices
        on side streams (ViT-B b512 36.1 k img/s against 35.2 k on one stream and 35.8 k on three; DeiT-B W4 40.1 / 38.2 / 39.1 k)."""
        if n_streams <= 1 or batch < 2:
            return [batch]
        if self.D > 384:
            if batch < 64:
                return [batch]
            return [(batch + 1) // 2, batch // 2]
        target = 8192 * 384 // self.D                                   # rows (batch x tokens) per slice
        k = min((2 * batch * self.tokens + target) // (2 * target), batch // 32, n_streams + 1)
        if k <= 1:
            return [batch]
        d = 1000 * (k - 1) + 765                                        # the caller's slice is ~0.77 of a side slice (52 : 68)
        side = (batch * 1000 + d - 1) // d
        return [side] * (k - 1) + [batch - (k - 1) * side]

    def forward_streams(self, images, bit_config, out, n_streams=3, slices=None):
        """Same result as ``forward``; the batch is cut into contiguous slices (``slice_sizes`` or an explicit list) that run on
        their own workspaces, round robin over ``n_streams`` side HIP streams and the caller's own stream (the slice after the side streams' runs there).  Images are
        independent, so this is only a scheduling choice: kernels of one slice (e.g. a VALU-bound GELU epilogue) overlap latency-
        or MFMA-bound phases of another slice's kernels."""
        images, cfg = self._check(images, bit_config)
        B = images.shape[0]
        n_streams = min(n_streams, E.compute_side_streams(self.device))      # (one less while an input pipeline copies on engine.copy_stream)
        sizes = list(slices) if slices is not None else self.slice_sizes(B, n_streams)
        if sum(sizes) != B or min(sizes) < 1:
            raise AssertionError('slices %r do not cover a batch of %d' % (sizes, B))
        self._check_out(out, B)
        if len(sizes) == 1:
            return self.forward(images, bit_config, out=out)
        with torch.cuda.device(self.device):
            n_side = min(len(sizes), max(n_streams, 1))
            self._streams = E.side_streams(self.device, n_side)         # the process's shared side streams of this device, never new ones
            if getattr(self, '_ws_multi', None) is None or len(self._ws_multi) < len(sizes):
                self._ws_multi = (getattr(self, '_ws_multi', None) or []) + [None] * (len(sizes) - len(getattr(self, '_ws_multi', None) or []))
            cur = torch.cuda.current_stream(self.device)
            L = E.lib()
            used = []
            lo = hi = 0
            per_stream = [[] for _ in range(n_side + 1)]             # the p2v_forward calls of every stream, in slice order
            for i, n_i in enumerate(sizes):
                lo, hi = hi, hi + n_i
                j = i % (n_side + 1)                                 # round robin over the side streams and the caller's stream
                st = self._streams[j] if j < n_side else cur
                n = L.p2v_workspace_bytes(self._handle, n_i)
                if self._ws_multi[i] is None or self._ws_multi[i].numel() < n:
                    self._ws_multi[i] = torch.empty(n, dtype=torch.uint8, device=self.device)
                if st is not cur and st not in used:
                    st.wait_stream(cur)
                    used.append(st)
                per_stream[j].append((E.ptr(images[lo:hi]), n_i, E.ptr(out[lo:hi]), E.ptr(self._ws_multi[i]), self._ws_multi[i].numel(),
                                      C.c_void_p(st.cuda_stream)))
            n_cfg, handle, dev_index = len(bit_config), self._handle, self.device.index

            def enqueue(calls, worker):
                if worker:
                    torch.cuda.set_device(dev_index)                 # (the current device is a per-thread setting)
                for xi, n_i, oi, ws, ws_n, st_ptr in calls:
                    E.check(L.p2v_forward(handle, xi, n_i, cfg, n_cfg, oi, ws, ws_n, -1, st_ptr))

            # One host thread per side stream (the C call releases the interpreter lock; p2v_forward only reads the plan): the four launch
            # sequences reach their queues together instead of one after the other - a synchronous step (forward, then read the logits) takes
            # 2.61 instead of 2.79 ms at DeiT-S / 256, and back-to-back steps 2.43 instead of 2.48 ms (profiles/r04_threaded_enqueue.txt)
            if E.THREADED_ENQUEUE and not torch.cuda.is_current_stream_capturing():
                futs = [E.enqueue_pool().submit(enqueue, calls, True) for calls in per_stream[:n_side] if calls]
                enqueue(per_stream[n_side], False)
                for f in futs:
                    f.result()
            else:
                for calls in per_stream:
                    enqueue(calls, False)
            for st in used:
                cur.wait_stream(st)
        return out

    def profile(self, images, bit_config):
        """per-launch times (ms, HIP events on the launch stream) of one forward: [(kind_name, ms), ...]; the last entry, 'event_gap', is an
        interval with no launch in it - what the event pair adds to every interval."""
        images, cfg = self._check(images, bit_config)
        B = images.shape[0]
        with torch.cuda.device(self.device):
            ws = self.workspace(B)
            out = torch.empty(B, self.arch['num_classes'], dtype=torch.float32, device=self.device)
            n_max = 7 * self.depth + 10
            ms = (C.c_float * n_max)()
            kind = (C.c_int32 * n_max)()
            n = E.lib().p2v_forward_profile(self._handle, E.ptr(images), B, cfg, len(bit_config), E.ptr(out), E.ptr(ws), ws.numel(),
                                            E.stream_ptr(self.device), ms, kind, n_max)
        if n < 0:
            E.check(n)
        return [(E.KERNEL_KINDS[kind[i]], float(ms[i])) for i in range(min(n, n_max))]

    def profile_streams(self, images, bit_config, n_streams=3, slices=None, rounds=3):
        """per-launch times UNDER OVERLAP: the slices of ``forward_streams`` run concurrently on their streams, each with HIP events
        between its launches (``p2v_forward_profile_begin`` / ``_end``); ``rounds`` consecutive steps are enqueued back to back so that
        the middle one runs in the steady state.  Returns (per-slice list of [(kind_name, ms), ...] of the middle round, wall ms of it).
        A batch that runs as ONE slice has no overlap: the isolated profile is returned."""
        images, cfg = self._check(images, bit_config)
        B = images.shape[0]
        n_streams = min(n_streams, E.compute_side_streams(self.device))
        sizes = list(slices) if slices is not None else self.slice_sizes(B, n_streams)
        if len(sizes) == 1:
            per = self.profile(images, bit_config)
            return [per], sum(ms for _, ms in per)
        out = torch.empty(B, self.arch['num_classes'], dtype=torch.float32, device=self.device)
        L = E.lib()
        n_max = 7 * self.depth + 10
        with torch.cuda.device(self.device):
            self.forward_streams(images, bit_config, out, n_streams, sizes)          # streams and workspaces exist
            torch.cuda.synchronize(self.device)
            cur = torch.cuda.current_stream(self.device)
            n_side = min(len(sizes), max(n_streams, 1))
            tokens = []
            try:
                for r in range(rounds):
                    lo = hi = 0
                    row = []
                    tokens.append(row)
                    for i, n_i in enumerate(sizes):
                        lo, hi = hi, hi + n_i
                        j = i % (n_side + 1)
                        st = self._streams[j] if j < n_side else cur
                        tok = C.c_void_p()
                        E.check(L.p2v_forward_profile_begin(self._handle, E.ptr(images[lo:hi]), n_i, cfg, len(bit_config), E.ptr(out[lo:hi]),
                                                            E.ptr(self._ws_multi[i]), self._ws_multi[i].numel(), C.c_void_p(st.cuda_stream), C.byref(tok)))
                        row.append(tok)
                res = []
                for row in tokens:
                    per = []
                    for j, tok in enumerate(row):
                        ms = (C.c_float * n_max)()
                        kind = (C.c_int32 * n_max)()
                        row[j] = None                                                     # ended (also when the call below fails)
                        n = L.p2v_forward_profile_end(tok, ms, kind, n_max)
                        if n < 0:
                            E.check(n)
                        per.append([(E.KERNEL_KINDS[kind[i]], float(ms[i])) for i in range(min(n, n_max))])
                    res.append(per)
            finally:                         # a failed begin / end must not leak the tokens (events + bookkeeping) already handed out
                for row in tokens:
                    for tok in row:
                        if tok is not None:
                            L.p2v_forward_profile_end(tok, None, None, 0)
        mid = res[len(res) // 2]
        return mid, max(sum(ms for _, ms in per) for per in mid)

    def view(self, batch, name, rows, cols):
        """int8 view of a workspace activation buffer (parity tests)."""
        off = E.lib().p2v_workspace_view(self._handle, batch, name.encode())
        if off < 0:
            raise KeyError(name)
        return self._ws[off: off + rows * cols].view(torch.int8).reshape(rows, cols)

    def __del__(self):
        try:
            from . import ops
            ops._PLANS.pop(int(self._handle.value or 0), None)
        except Exception:
            pass
        try:
            if self._handle:
                E.lib().p2v_plan_destroy(self._handle)
                self._handle = C.c_void_p()
        except Exception:
            pass

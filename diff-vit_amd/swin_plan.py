"""Frozen integer plan of a calibrated Swin model: int8 weight codes and per-channel requant constants on the GPU, and a forward
that is a fixed sequence of HIP kernels called through the C ABI (``p2v_quantize_patchify``, ``p2v_gemm_i8``,
``p2v_int_layernorm``, ``p2v_window_attention``, ``p2v_patch_merge_gather``, ``p2v_avgpool_quant``).  Window partition, cyclic
shift and their inverses are index tables consumed by the attention kernel; nothing runs on the CPU and nothing falls back.

Call order = swin_quant.py (forward_features :790-811, SwinTransformerBlock.forward :351-399, PatchMerging.forward :438-461).
"""
import ctypes as C

import numpy as np
import torch

from . import engine as E
from .swin import relative_position_index, shifted_window_regions  # noqa: F401


def _pad128(n):
    return (n + 127) // 128 * 128


def _lis_consts(sf):
    """x0_int, b_int, c_int of the I-BERT polynomial in fp32 (layers.py:334-351); range-checked like the ViT plan's."""
    from .plan import lis_consts
    return lis_consts(torch.tensor(float(sf), dtype=torch.float32))


def _window_index(H, W, ws, shift):
    hh = (torch.arange(H) + shift) % H
    ww = (torch.arange(W) + shift) % W
    grid = hh[:, None] * W + ww[None, :]
    return grid.reshape(H // ws, ws, W // ws, ws).permute(0, 2, 1, 3).reshape(-1, ws * ws)


class SwinPlan:
    def __init__(self, arch, state_dict, calib, device='cuda', in_chans=3, bits=8):
        self.arch, self.device, self.in_chans, self.bits = dict(arch), torch.device(device), in_chans, bits
        if self.device.type != 'cuda':
            raise RuntimeError('SwinPlan needs a GPU device: the quantized forward has no CPU path')
        if self.device.index is None:
            self.device = torch.device('cuda', torch.cuda.current_device())
        E.lib()
        self._keep = []
        self.W = {k: v.detach().float().cpu() for k, v in state_dict.items() if v.dtype == torch.float32}
        self.c = calib
        self._recorded = {}
        with torch.cuda.device(self.device):      # uploads and GELU-table builds on the plan's device, whatever the caller's current one
            self._build()

    # ---- helpers -------------------------------------------------------------------------------------------------------
    def _dev(self, t, dtype=torch.float32):
        t = t.to(dtype).contiguous().to(self.device)
        self._keep.append(t)
        return t

    def _vec(self, v, n):
        """per-channel fp32 vector of length n, padded to a multiple of 128 (whole tiles are staged)."""
        v = torch.as_tensor(v, dtype=torch.float32).reshape(-1)
        if v.numel() == 1:
            v = v.expand(n)
        out = torch.ones(_pad128(n))
        out[:n] = v
        return self._dev(out)

    def _pot(self, name):
        s = float(self.c[name].reshape(-1)[0])
        m, _ = np.frexp(s)
        if self.c[name].numel() != 1 or m != 0.5:
            raise NotImplementedError('%s: the engine needs a power-of-two layer-wise scale, got %r' % (name, self.c[name][:4]))
        return s

    def _linear(self, name, s_x, bias=True, k_pad=None, frag=False):
        """int8-stored weight codes [n_pad][k_pad], colscale = s_x * s_w, bias; ``frag``: also the MFMA-fragment-order copy the fused
        LayerNorm+GEMM kernel streams (qkv / fc1 of stages up to 384 channels)."""
        bt = 'int%d' % self.bits
        w = self.W[name + '.weight']
        w2 = w.reshape(w.shape[0], -1)
        N, K = w2.shape
        s_w = self.c[name][bt].reshape(-1)
        lo, hi = (-128, 127) if self.bits == 8 else (-8, 7)
        codes = torch.clamp(torch.round(w2 / s_w.reshape(-1, 1)), lo, hi)
        kp = k_pad or (K + 63) // 64 * 64          # the GEMM walks 64-deep k-tiles: extra weight columns are zero
        wp = torch.zeros(_pad128(N), kp, dtype=torch.int8)
        wp[:N, :K] = codes.to(torch.int8)
        cs = torch.zeros(_pad128(N))
        cs[:N] = (torch.tensor(float(s_x)) * s_w.expand(N)).float()
        b = torch.zeros(_pad128(N))
        if bias:
            b[:N] = self.W[name + '.bias']
        d = dict(w=self._dev(wp, torch.int8), cs=self._dev(cs), b=self._dev(b), N=N, K=kp)
        d['wf'] = self._dev(E.fragment_order(wp), torch.int8) if frag else None
        d['lin'] = E.Linear(E.ptr(d['w']), E.ptr(d['cs']), E.ptr(d['b']), E.ptr(d['wf']) if frag else None, 0)
        return d

    def _ln(self, prefix, s_in_vec, s_out, C_):
        """QIntLayerNorm 'int' constants: mask = round(in_scale / min), inv_out = 1 / out_scale; the following QAct has the LN's
        own output scale, so post_mul = 1."""
        s_in_vec = torch.as_tensor(s_in_vec, dtype=torch.float32).reshape(-1)
        if s_in_vec.numel() == 1:
            s_in_vec = s_in_vec.expand(C_)
        s1 = s_in_vec.min()
        mask_max = float(torch.round(s_in_vec / s1).max())
        if C_ * (128.0 * mask_max) ** 2 >= 2.0 ** 32:      # the kernel keeps sum(x_q^2) exactly in 32 unsigned bits
            raise NotImplementedError('LayerNorm input scale ratios up to %g over %d channels exceed the exact 32-bit statistics' % (mask_max, C_))
        t = [self._dev(torch.round(s_in_vec / s1)), self._dev(self.W[prefix + '.weight']), self._dev(self.W[prefix + '.bias']),
             self._dev(torch.full((C_,), 1.0 / float(s_out))), self._dev(torch.ones(C_))]
        ln = E.Ln(float(s1), *[E.ptr(x) for x in t])
        # the constants folded once here instead of by every workgroup of every launch (p2v_ln_prefold: gamma / out_scale, beta / out_scale and
        # the fast-chain tests; the wide stages' stand-alone LayerNorm folds for only 16 - 32 rows per workgroup otherwise)
        L = E.lib()
        nbytes = L.p2v_ln_prefold_bytes(C_)
        buf = torch.empty(nbytes // 4, dtype=torch.float32, device=self.device)
        self._keep.append(buf)
        E.check(L.p2v_ln_prefold(C.byref(ln), C_, E.ptr(buf), nbytes))
        return ln

    # ---- plan ----------------------------------------------------------------------------------------------------------
    def _build(self):
        a, c = self.arch, self.c
        P, D0, ws = a['patch_size'], a['embed_dim'], a['window_size']
        g = a['img_size'] // P
        self.g = g
        self.s_in = self._pot('qact_input')
        kpatch = self.in_chans * P * P
        self.k_patch = (kpatch + 63) // 64 * 64
        self.pe = self._linear('patch_embed.proj', self.s_in, k_pad=self.k_patch)
        self.s_pe_b = self._pot('patch_embed.qact_before_norm')
        s_pe = self._pot('patch_embed.qact')
        self.pe_ln = self._ln('patch_embed.norm', torch.tensor([self.s_pe_b]), s_pe, D0)
        s_res = torch.full((D0,), s_pe)
        self.stages = []
        H = g
        for li, depth in enumerate(a['depths']):
            Cc = D0 * 2 ** li
            heads = a['num_heads'][li]
            if Cc // heads != 32:
                raise NotImplementedError('window attention kernel: head_dim must be 32 (got %d)' % (Cc // heads))
            wsz = min(ws, H)
            blocks = []
            for bi in range(depth):
                p = 'layers.%d.blocks.%d.' % (li, bi)
                shift = 0 if (bi % 2 == 0 or H <= ws) else ws // 2
                s1 = self._pot(p + 'qact1')
                b = dict(C=Cc, heads=heads, H=H, ws=wsz, shift=shift)
                b['ln1'] = self._ln(p + 'norm1', s_res, s1, Cc)
                s_q1 = self._pot(p + 'attn.qact1')
                b['qkv'] = self._linear(p + 'attn.qkv', s1, frag=Cc <= 384)
                b['inv_s_qkv'] = 1.0 / s_q1
                s_q2 = self._pot(p + 'attn.qact2')
                s_q3 = self._pot(p + 'attn.qact3')
                tab = torch.clamp(torch.round(self.W[p + 'attn.relative_position_bias_table'] / self._pot(p + 'attn.qact_table')), -128, 127)
                idx = _window_index(H, H, wsz, shift)
                reg = shifted_window_regions(H, H, wsz, shift) if shift else None
                x0, bb, cc = _lis_consts(s_q2)
                b['tab'], b['idx'] = self._dev(tab, torch.int8), self._dev(idx, torch.int32)
                b['reg'] = self._dev(reg, torch.int8) if reg is not None else None
                b['wa'] = E.WinAttn(s_q1, float(np.float32(32 ** -0.5)), self._pot(p + 'attn.qact_attn1'), self._pot(p + 'attn.qact_table'),
                                    s_q2, s_q3, x0, bb, cc, E.ptr(b['tab']), E.ptr(b['idx']),
                                    E.ptr(b['reg']) if b['reg'] is not None else None, wsz, idx.shape[0])
                b['proj'] = self._linear(p + 'attn.proj', s_q3)
                s_b2 = c[p + 'qact2'].reshape(-1)
                b['proj_epi'] = self._resid_epi(self._pot(p + 'attn.qact4'), s_res, s_b2, Cc, b['proj'])
                s3 = self._pot(p + 'qact3')
                b['ln2'] = self._ln(p + 'norm2', s_b2, s3, Cc)
                b['fc1'] = self._linear(p + 'mlp.fc1', s3, frag=Cc <= 384)
                s_m1 = self._pot(p + 'mlp.qact1')
                b['inv_s_fc1'] = 1.0 / s_m1
                b['fc2'] = self._linear(p + 'mlp.fc2', s_m1)
                s_b4 = c[p + 'qact4'].reshape(-1)
                b['fc2_epi'] = self._resid_epi(self._pot(p + 'mlp.qact2'), s_b2, s_b4, Cc, b['fc2'])
                s_res = s_b4.clone()
                blocks.append(b)
            st = dict(blocks=blocks, C=Cc, H=H, merge=None)
            if li < len(a['depths']) - 1:
                p = 'layers.%d.downsample.' % li
                sd1 = self._pot(p + 'qact1')
                sd2 = c[p + 'qact2'].reshape(-1)
                red = self._linear(p + 'reduction', sd1, bias=False)
                # single PTF requant through the RESID epilogue: s_mid = s_next and an all-zero residual make it
                # Q(Q(y; s) * s; s) = Q(y; s)   (|code * eps| << 0.5)
                st['merge'] = dict(ln=self._ln(p + 'norm', s_res.repeat(4), sd1, 4 * Cc), red=red,
                                   epi=self._resid_epi(sd2, torch.ones(2 * Cc), sd2, 2 * Cc, red))
                s_res = sd2.clone()
                H //= 2
            self.stages.append(st)
        Cl = D0 * 2 ** (len(a['depths']) - 1)
        self.C_last, self.H_last = Cl, H
        self.s_f = self._pot('qact2')
        self.fin_ln = self._ln('norm', s_res, self.s_f, Cl)
        self.s_pool = self._pot('qact3')
        self.head = self._linear('head', self.s_pool)
        self.s_out = self._pot('act_out')

    def _resid_epi(self, s_mid, s_res, s_next, n, lin=None):
        """constants of a RESID epilogue; with the layer's weights (``lin``) also the pre-folded table of ``p2v_resid_prefold`` - used
        only when the library proves it gives the reference's codes for these constants (all 65 536 numerators of every channel)."""
        e = E.Epilogue()
        t = [self._vec(s_mid, n), self._vec(s_res, n), self._vec(s_next, n)]
        e.s_mid, e.s_res, e.s_next = [C.cast(E.ptr(x), C.c_void_p) for x in t]
        if lin is not None:
            L = E.lib()
            nbytes = L.p2v_resid_prefold_bytes(n)
            tab = torch.empty(nbytes // 4, dtype=torch.float32, device=self.device)
            usable = C.c_int(0)
            torch.cuda.synchronize(self.device)                       # the vectors above were uploaded on torch's stream
            E.check(L.p2v_resid_prefold(C.byref(lin['lin']), C.byref(e), n, E.ptr(tab), nbytes, C.byref(usable), None))
            if usable.value:
                self._keep.append(tab)
                e.resid_tab = C.cast(E.ptr(tab), C.c_void_p)
            self.resid_prefolded = getattr(self, 'resid_prefolded', []) + [bool(usable.value)]
        return e

    # ---- forward -------------------------------------------------------------------------------------------------------
    def _record(self, B, fused=True):
        """the launch sequence of one forward at batch B as a ``p2v_op`` array with its own activation buffers (built once
        per batch size and stream slot, replayed by ``p2v_run_ops``).  ``fused``: norm1 + qkv and norm2 + fc1 of stages up to 384
        channels run as ONE launch each (``P2V_OP_LN_GEMM``: the LayerNorm output stays in LDS); the tap replay records the
        unfused sequence, whose LayerNorm outputs exist in HBM."""
        a = self.arch
        P, g = a['patch_size'], self.g
        dev = self.device
        ops, taps, keep = [], [], []

        def buf(rows, cols, zero=False):
            t_ = (torch.zeros if zero else torch.empty)(rows, cols, dtype=torch.int8, device=dev)
            keep.append(t_)
            return t_

        def pad64(n):
            return (n + 63) // 64 * 64

        def gemm(kind, x, lin, epi, out):
            o = E.Op()
            o.kind, o.epi, o.inp, o.out = E.OP_GEMM, kind, E.ptr(x), E.ptr(out)
            o.M, o.K, o.N, o.lda, o.ldo = x.shape[0], x.shape[1], lin['N'], x.shape[1], lin['N']
            o.lin, o.ep = lin['lin'], epi
            ops.append(o)
            return out

        def ln_gemm(kind, x, ln, Cc, lin, epi, out):
            o = E.Op()
            o.kind, o.epi, o.inp, o.out = E.OP_LN_GEMM, kind, E.ptr(x), E.ptr(out)
            o.M, o.K, o.N, o.lda, o.ldo = x.shape[0], Cc, lin['N'], x.shape[1], lin['N']
            o.lin, o.ep, o.ln = lin['lin'], epi, ln
            ops.append(o)
            return out

        def lnorm(x, ln, Cc, out):
            o = E.Op()
            o.kind, o.inp, o.out = E.OP_LAYERNORM, E.ptr(x), E.ptr(out)
            o.M, o.N, o.lda, o.ldo, o.ln = x.shape[0], Cc, x.shape[1], out.shape[1], ln
            ops.append(o)
            return out

        def epi_req(inv_s):
            e = E.Epilogue()
            e.inv_s_out = inv_s
            return e

        def epi_res(src, residual):
            e = E.Epilogue()
            e.s_mid, e.s_res, e.s_next, e.residual, e.resid_tab = src.s_mid, src.s_res, src.s_next, E.ptr(residual), src.resid_tab
            return e

        o = E.Op()
        patches = buf(B * g * g, self.k_patch, zero=True)
        o.kind, o.out = E.OP_PATCHIFY, E.ptr(patches)
        o.i0, o.i1, o.i2, o.i3, o.i4, o.i5, o.f0 = B, self.in_chans, a['img_size'], a['img_size'], P, self.k_patch, 1.0 / self.s_in
        ops.append(o)
        D0 = a['embed_dim']
        pe = gemm(E.EPI_REQUANT, patches, self.pe, epi_req(1.0 / self.s_pe_b), buf(B * g * g, D0))
        x = lnorm(pe, self.pe_ln, D0, buf(B * g * g, D0))
        taps.append((len(ops), 'patch_embed.qact', x))
        for li, stg in enumerate(self.stages):
            T, Cc = stg['H'] * stg['H'], stg['C']
            rows = B * T
            x2 = buf(rows, Cc)
            ln = buf(rows, pad64(Cc))
            qkv = buf(rows, 3 * Cc)
            att = buf(rows, pad64(Cc))
            hid = buf(rows, 4 * Cc)
            for bi, b in enumerate(stg['blocks']):
                p = 'layers.%d.blocks.%d.' % (li, bi)
                e_fc1 = epi_req(b['inv_s_fc1'])
                e_fc1.gelu = E.gelu_table(b['inv_s_fc1'], self.device)       # exact GELU -> qact1 threshold table (cached per scale)
                fuse = (fused and b['qkv']['wf'] is not None and E.lib().p2v_ln_gemm_fusable(E.EPI_REQUANT, Cc, b['qkv']['N'], 0) == 1
                        and E.lib().p2v_ln_gemm_fusable(E.EPI_GELU, Cc, b['fc1']['N'], e_fc1.gelu.cells if e_fc1.gelu.table else 0) == 1)
                if fuse:
                    ln_gemm(E.EPI_REQUANT, x, b['ln1'], Cc, b['qkv'], epi_req(b['inv_s_qkv']), qkv)
                else:
                    lnorm(x, b['ln1'], Cc, ln)
                    taps.append((len(ops), p + 'qact1', ln))
                    gemm(E.EPI_REQUANT, ln, b['qkv'], epi_req(b['inv_s_qkv']), qkv)
                o = E.Op()
                o.kind, o.inp, o.out = E.OP_WINATTN, E.ptr(qkv), E.ptr(att)
                o.i0, o.i1, o.i2, o.i3 = B, T, b['heads'], 32
                o.wa = b['wa']
                o.wa.out_stride = att.shape[1]
                ops.append(o)
                gemm(E.EPI_RESID, att, b['proj'], epi_res(b['proj_epi'], x), x2)
                taps.append((len(ops), p + 'qact2', x2))
                if fuse:
                    ln_gemm(E.EPI_GELU, x2, b['ln2'], Cc, b['fc1'], e_fc1, hid)
                else:
                    lnorm(x2, b['ln2'], Cc, ln)
                    gemm(E.EPI_GELU, ln, b['fc1'], e_fc1, hid)
                gemm(E.EPI_RESID, hid, b['fc2'], epi_res(b['fc2_epi'], x2), x)
                taps.append((len(ops), p + 'qact4', x))
            if stg['merge'] is not None:
                m, H = stg['merge'], stg['H']
                r2 = B * (H // 2) * (H // 2)
                gathered = buf(r2, 4 * Cc)
                o = E.Op()
                o.kind, o.inp, o.out = E.OP_MERGE, E.ptr(x), E.ptr(gathered)
                o.i0, o.i1, o.i2, o.i3 = B, H, H, Cc
                ops.append(o)
                lnm = lnorm(gathered, m['ln'], 4 * Cc, buf(r2, 4 * Cc))
                x = gemm(E.EPI_RESID, lnm, m['red'], epi_res(m['epi'], buf(r2, 2 * Cc, zero=True)), buf(r2, 2 * Cc))
                taps.append((len(ops), 'layers.%d.downsample.qact2' % li, x))
        fin = lnorm(x, self.fin_ln, self.C_last, buf(x.shape[0], self.C_last))
        taps.append((len(ops), 'qact2', fin))
        pooled = buf(B, self.C_last)
        o = E.Op()
        o.kind, o.inp, o.out = E.OP_AVGPOOL, E.ptr(fin), E.ptr(pooled)
        o.i0, o.i1, o.i2, o.f0, o.f1 = B, self.H_last * self.H_last, self.C_last, self.s_f, 1.0 / self.s_pool
        ops.append(o)
        taps.append((len(ops), 'qact3', pooled))
        e = E.Epilogue()
        e.inv_s_out, e.s_out = 1.0 / self.s_out, self.s_out
        o = E.Op()
        o.kind, o.epi, o.inp = E.OP_GEMM, E.EPI_HEAD, E.ptr(pooled)
        o.M, o.K, o.N, o.lda, o.ldo = B, self.C_last, self.head['N'], self.C_last, self.head['N']
        o.lin, o.ep = self.head['lin'], e
        ops.append(o)
        arr = (E.Op * len(ops))(*ops)
        return dict(ops=arr, n=len(ops), taps=taps, keep=keep)

    def _replay(self, images, slot, taps=None, profile=False):
        with torch.cuda.device(self.device):       # the recorded pointers and the launch stream belong to the plan's GPU
            return self._replay_on_device(images, slot, taps, profile)

    def _replay_on_device(self, images, slot, taps, profile):
        B = images.shape[0]
        key = (B, slot, taps is None)
        if key not in self._recorded:
            self._recorded[key] = self._record(B, fused=taps is None)
        r = self._recorded[key]
        out = torch.empty(B, self.head['N'], dtype=torch.float32, device=self.device)
        r['ops'][0].inp = E.ptr(images)
        r['ops'][r['n'] - 1].out = E.ptr(out)
        L = E.lib()
        if profile:
            ms = (C.c_float * r['n'])()
            E.check(L.p2v_run_ops_profile(r['ops'], r['n'], E.stream_ptr(self.device), ms))
            return out, list(ms)
        if taps is None:
            E.check(L.p2v_run_ops(r['ops'], r['n'], E.stream_ptr(self.device)))
            return out
        done = 0
        for upto, name, t_ in r['taps']:          # replay in segments and copy the tapped buffers (they are reused later)
            E.check(L.p2v_run_ops(C.cast(C.byref(r['ops'], done * C.sizeof(E.Op)), C.POINTER(E.Op)), upto - done, E.stream_ptr(self.device)))
            taps[name] = t_[:, :t_.shape[1]].clone()
            done = upto
        E.check(L.p2v_run_ops(C.cast(C.byref(r['ops'], done * C.sizeof(E.Op)), C.POINTER(E.Op)), r['n'] - done, E.stream_ptr(self.device)))
        return out

    def _check_images(self, images):
        a = self.arch
        images = images.contiguous().float()
        if images.device != self.device:
            raise RuntimeError('images must live on %s' % self.device)
        if tuple(images.shape[1:]) != (self.in_chans, a['img_size'], a['img_size']):
            raise AssertionError("Input image size (%d*%d) doesn't match model (%d*%d)." % (images.shape[2], images.shape[3], a['img_size'], a['img_size']))
        return images

    def forward(self, images, taps=None, n_streams=3, slices=None):
        """images fp32 [B, in_chans, S, S] on the plan's device -> logits fp32 [B, classes] (act_out grid).  One C call replays
        the recorded launch sequence; a large batch runs as ``n_streams`` contiguous slices on their own HIP streams (images are
        independent), like the ViT plan (three: Swin-B at 256 images 25.1 k img/s against 24.6 k on two, same call, profiles/r04_slices.txt).
        ``slices``: explicit slice sizes; slices beyond ``n_streams`` run on the caller's stream."""
        images = self._check_images(images)
        B = images.shape[0]
        n_streams = min(n_streams, E.compute_side_streams(self.device))       # (one less while an input pipeline copies on engine.copy_stream)
        if taps is not None or n_streams <= 1 or (slices is None and B < 16 * n_streams):
            return self._replay(images, 0, taps)
        if slices is None:
            step = (B + n_streams - 1) // n_streams
            slices = [min(step, B - i * step) for i in range(n_streams) if i * step < B]
        if sum(slices) != B or min(slices) < 1:
            raise AssertionError('slices %r do not cover a batch of %d' % (list(slices), B))
        self._streams = E.side_streams(self.device, n_streams)          # the process's shared side streams of this device (engine.side_streams)
        cur = torch.cuda.current_stream(self.device)
        parts, lo = [], 0
        for i, n_i in enumerate(slices):
            parts.append(images[lo:lo + n_i])
            lo += n_i
        for st in self._streams[:min(n_streams, len(slices))]:
            st.wait_stream(cur)

        def on_side_stream(i, worker):
            if worker:
                torch.cuda.set_device(self.device)     # (current device and current stream are per-thread settings)
            with torch.cuda.stream(self._streams[i]):
                return self._replay(parts[i], i + 1)

        # one host thread per side stream once the launch sequences are recorded (the replay is one C call of ~1 ms of host time per slice),
        # like FrozenPlan.forward_streams; slices beyond the side streams run on the caller's own stream (slot 0: its single-stream record)
        n_side = min(n_streams, len(slices))
        recorded = all((parts[i].shape[0], i + 1, True) in self._recorded for i in range(n_side))
        if E.THREADED_ENQUEUE and recorded and not torch.cuda.is_current_stream_capturing():
            futs = [E.enqueue_pool().submit(on_side_stream, i, True) for i in range(n_side)]
            tail = [self._replay(xi, 0) for xi in parts[n_side:]]
            outs = [f.result() for f in futs] + tail
        else:
            outs = [on_side_stream(i, False) for i in range(n_side)] + [self._replay(xi, 0) for xi in parts[n_side:]]
        for st in self._streams[:min(n_streams, len(slices))]:
            cur.wait_stream(st)
        for o in outs:
            o.record_stream(cur)
        return torch.cat(outs, 0)

    def profile(self, images):
        """per-op durations (ms, HIP events on the launch stream) of one single-stream forward: [(kind, epilogue, ms), ...]."""
        images = self._check_images(images)
        _, ms = self._replay(images, 0, profile=True)
        r = self._recorded[(images.shape[0], 0, True)]
        names = ('patchify', 'gemm', 'layernorm', 'window_attention', 'merge_gather', 'avgpool', 'ln_gemm')
        return [(names[r['ops'][i].kind], int(r['ops'][i].epi), ms[i]) for i in range(r['n'])]

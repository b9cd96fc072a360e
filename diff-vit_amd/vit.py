"""ViT / DeiT graph of the PoT-PTQ path behind the reference's module surface.

Mirror of the reference's models/vit_fquant.py (Attention :57-349, Block :352-484, VisionTransformer :487-799,
factories :802-933) and models/layers_quant.py (Mlp :141-351, PatchEmbed :354-492): same class names, constructor
arguments, attribute names (``qact0 qkv qact1 qact_attn1 log_int_softmax qact2 proj qact3 channel_scale best_scale
best_act_scale best_weight_scale qkv_output fc1_output`` ...), state-dict keys (timm/DeiT compatible) and
``forward(x, bit_config=None, plot=False, hessian_statistic=False) -> (logits, FLOPs, global_distance)``.

Two execution regimes:
  * float / calibration (``model.quant`` False or a calibrate flag open): the torch graph below -- float ops on the
    tensors' own device, with the SmoothQuant power-of-two search and the observers; this is the one-off pass.
  * quantized inference (after ``model_close_calibrate(); model_quant()``): ``forward`` freezes the calibrated state
    into an integer plan once (``plan.FrozenPlan``) and hands every batch to the fused HIP engine through the C ABI
    (one ``p2v_forward`` call, 7 kernels per block).  There is no eager fallback: without the built library or on a
    CPU tensor the call raises.
"""
from collections import OrderedDict
from functools import partial

import torch
import torch.nn.functional as F
from torch import nn

from .ptq import BIT_TYPE_DICT, QAct, QConv2d, QIntLayerNorm, QIntSoftmax, QLinear
from .ptq.observer import round_ln

__all__ = ['deit_tiny_patch16_224', 'deit_small_patch16_224', 'deit_base_patch16_224', 'vit_base_patch16_224',
           'vit_large_patch16_224']

alpha_pool = [0.35]        # vit_fquant.py:32
mlp_alpha_pool = [0.5]     # layers_quant.py:14
bit_pool = [4, 8]          # vit_fquant.py:33


def _qact(cfg, quant, calibrate, ln=False):
    return QAct(quant=quant, calibrate=calibrate, bit_type=cfg.BIT_TYPE_A,
                calibration_mode=cfg.CALIBRATION_MODE_A_LN if ln else cfg.CALIBRATION_MODE_A,
                observer_str=cfg.OBSERVER_A_LN if ln else cfg.OBSERVER_A,
                quantizer_str=cfg.QUANTIZER_A_LN if ln else cfg.QUANTIZER_A)


def _qlinear(cfg, i, o, quant, calibrate, bias=True):
    return QLinear(i, o, bias=bias, quant=quant, calibrate=calibrate, bit_type=cfg.BIT_TYPE_W,
                   calibration_mode=cfg.CALIBRATION_MODE_W, observer_str=cfg.OBSERVER_W, quantizer_str=cfg.QUANTIZER_W)


class _SmoothQuantMixin:
    """SmoothQuant with power-of-two channel scales, shared by Attention (qkv) and Mlp (fc1):
    channel_scale = 2^round(log2(max|x|^a / max|W|^(1-a))) (vit_fquant.py:199-293, layers_quant.py:222-313)."""

    def _smooth_linear(self, x, lin, qact0, pool, global_distance, bit_config, extra):
        if self.channel_scale is None or bit_config == -1:
            gmax = torch.abs(x).max(axis=1).values.max(axis=0).values
            wmax = torch.abs(lin.weight).max(axis=0).values
            cs_pool, loss_pool, act_scale, act_zp, w_scale, w_zp = [], [[], []], [], [], [], []
            if self.channel_scale is None:
                self.best_scale, self.best_act_scale, self.best_act_zp = [], [], []
                self.best_weight_scale, self.best_weight_zp = [], []
            for alpha in pool:
                cs = 2**round_ln(gmax**alpha / (wmax**(1 - alpha)), 'round')
                cs_pool.append(cs)
                x_s = x / cs.reshape((1, 1, -1))
                w_s = lin.weight * cs.reshape((1, -1))
                gt = F.linear(x_s, w_s, lin.bias)
                mid = qact0(x_s)
                if qact0.last_calibrate and bit_config != -1:
                    act_scale.append(qact0.quantizer.scale)
                    act_zp.append(qact0.quantizer.zero_point)
                    lin(mid, global_distance, bit_config, w_s, **extra)
                    w_scale.append(lin.quantizer.dic_scale)
                    w_zp.append(lin.quantizer.dic_zero_point)
                    qact0.calibrate, qact0.quant = False, True
                    mid = qact0(x_s)
                    lin.calibrate, lin.quant = False, True
                    for j, bit in enumerate(bit_pool):
                        out = lin(mid, global_distance, bit, w_s, **extra)
                        loss_pool[j].append((gt - out).abs().pow(2.0).mean())
                    qact0.quant, qact0.calibrate = False, True
                    lin.quant, lin.calibrate = False, True
            if qact0.last_calibrate and bit_config != -1:
                for loss in loss_pool:
                    i = loss.index(min(loss))
                    self.channel_scale = cs_pool[i]
                    self.best_scale.append(cs_pool[i])
                    self.best_act_scale.append(act_scale[i])
                    self.best_act_zp.append(act_zp[i])
                    self.best_weight_scale.append(w_scale[i])
                    self.best_weight_zp.append(w_zp[i])
            return gt
        i = bit_pool.index(bit_config)                      # ValueError for widths outside the pool
        self.channel_scale = self.best_scale[i]
        cs = self.channel_scale.to(x.device)                # the calibrated state may live on the host (calibrate_model where='host')
        x_s = x / cs.reshape((1, 1, -1))
        w_s = lin.weight * cs.reshape((1, -1))
        qact0.quantizer.scale = self.best_act_scale[i]
        qact0.quantizer.zero_point = self.best_act_zp[i]
        lin.quantizer.dic_scale = self.best_weight_scale[i]
        lin.quantizer.dic_zero_point = self.best_weight_zp[i]
        return lin(qact0(x_s), global_distance, bit_config, w_s, **extra)


class Attention(nn.Module, _SmoothQuantMixin):

    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0.0, proj_drop=0.0, quant=False,
                 calibrate=False, cfg=None):
        super().__init__()
        self.num_heads = num_heads
        head_dim = dim // num_heads
        self.calibrate = calibrate
        self.scale = qk_scale or head_dim**-0.5
        self.qkv = _qlinear(cfg, dim, dim * 3, quant, calibrate, bias=qkv_bias)
        self.qact0 = _qact(cfg, quant, calibrate)
        self.qact1 = _qact(cfg, quant, calibrate)
        self.qact2 = _qact(cfg, quant, calibrate)
        self.proj = _qlinear(cfg, dim, dim, quant, calibrate)
        self.qact3 = _qact(cfg, quant, calibrate, ln=True)
        self.qact_attn1 = _qact(cfg, quant, calibrate)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj_drop = nn.Dropout(proj_drop)
        self.log_int_softmax = QIntSoftmax(log_i_softmax=cfg.INT_SOFTMAX, quant=quant, calibrate=calibrate,
                                           bit_type=cfg.BIT_TYPE_S, calibration_mode=cfg.CALIBRATION_MODE_S,
                                           observer_str=cfg.OBSERVER_S, quantizer_str=cfg.QUANTIZER_S)
        self.channel_scale = None
        self.qkv_output = None

    def forward(self, x, FLOPs, global_distance, atten_bit_config, plot=False, quant=False, smoothquant=True,
                hessian_statistic=False):
        self.atten_bit_config = atten_bit_config
        B, N, C = x.shape
        bit_config = atten_bit_config[0] if atten_bit_config else None
        extra = dict(attn=False, attn_para=[self.num_heads, C, self.scale])
        if smoothquant and not hessian_statistic:
            x = self._smooth_linear(x, self.qkv, self.qact0, alpha_pool, global_distance, bit_config, extra)
        else:
            x = self.qkv(self.qact0(x), global_distance, bit_config, None, **extra)
        self.qkv_output = x.detach().clone()
        B, N, M = x.shape
        FLOPs.append(N * C * M)
        x = self.qact1(x, **extra)
        qkv = x.reshape(B, N, 3, self.num_heads, C // self.num_heads).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        attn = (q @ k.transpose(-2, -1)) * self.scale
        attn = self.qact_attn1(attn)
        attn = self.log_int_softmax(attn, self.qact_attn1.quantizer.scale)
        attn = self.attn_drop(attn)
        x = (attn @ v).transpose(1, 2).reshape(B, N, C)
        x = self.qact2(x)
        bit_config = atten_bit_config[1] if atten_bit_config else None
        x = self.proj(x, global_distance, bit_config)
        FLOPs.append(N * C * x.shape[2])
        x = self.qact3(x)
        return self.proj_drop(x)

    def get_requant_scale(self):
        bt = 'int' + str(self.atten_bit_config[1])
        return (self.qact2.quantizer.scale * self.proj.quantizer.dic_scale[BIT_TYPE_DICT[bt].name]) / self.qact3.quantizer.scale


class Mlp(nn.Module, _SmoothQuantMixin):

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.0, quant=False,
                 calibrate=False, cfg=None):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.qact0 = _qact(cfg, quant, calibrate)
        self.fc1 = _qlinear(cfg, in_features, hidden_features, quant, calibrate)
        self.act = act_layer()
        self.qact1 = _qact(cfg, quant, calibrate)
        self.fc2 = _qlinear(cfg, hidden_features, out_features, quant, calibrate)
        self.qact2 = _qact(cfg, quant, calibrate, ln=True)
        self.drop = nn.Dropout(drop)
        self.channel_scale = None
        self.fc1_output = None

    def forward(self, x, FLOPs, global_distance, ffn_bit_config, plot=False, quant=True, smoothquant=True,
                activation=None, hessian_statistic=False):
        B, N, C = x.shape
        bit_config = ffn_bit_config[0] if ffn_bit_config else None
        if smoothquant and not hessian_statistic:
            x = self._smooth_linear(x, self.fc1, self.qact0, mlp_alpha_pool, global_distance, bit_config, {})
        else:
            x = self.fc1(self.qact0(x), global_distance, bit_config, None)
        self.fc1_output = x.detach().clone()
        FLOPs.append(N * C * x.shape[2])
        x = self.act(x)
        x = self.qact1(x, asymmetric=False)
        x = self.drop(x)
        B, N, C = x.shape
        bit_config = ffn_bit_config[1] if ffn_bit_config else None
        x = self.fc2(x, global_distance, bit_config)
        FLOPs.append(N * C * x.shape[2])
        x = self.qact2(x)
        return self.drop(x)


class PatchEmbed(nn.Module):
    """Image to Patch Embedding (layers_quant.py:354-492)."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, norm_layer=None, quant=False,
                 calibrate=False, cfg=None):
        super().__init__()
        img_size = (img_size, img_size) if isinstance(img_size, int) else tuple(img_size)
        patch_size = (patch_size, patch_size) if isinstance(patch_size, int) else tuple(patch_size)
        self.img_size = img_size
        self.patch_size = patch_size
        self.grid_size = (img_size[0] // patch_size[0], img_size[1] // patch_size[1])
        self.num_patches = self.grid_size[0] * self.grid_size[1]
        self.proj = QConv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size, quant=quant,
                            calibrate=calibrate, bit_type=cfg.BIT_TYPE_W, calibration_mode=cfg.CALIBRATION_MODE_W,
                            observer_str=cfg.OBSERVER_W, quantizer_str=cfg.QUANTIZER_W)
        if norm_layer:
            self.qact_before_norm = _qact(cfg, quant, calibrate)
            self.norm = norm_layer(embed_dim)
        else:
            self.qact_before_norm = nn.Identity()
            self.norm = nn.Identity()
        self.qact = _qact(cfg, quant, calibrate)

    def forward(self, x, FLOPs, bit_config):
        B, C, H, W = x.shape
        assert H == self.img_size[0] and W == self.img_size[1], \
            f"Input image size ({H}*{W}) doesn't match model ({self.img_size[0]}*{self.img_size[1]})."
        x = self.proj(x, bit_config)
        B, M, H, W = x.shape
        FLOPs.append(C * self.patch_size[0] * self.patch_size[0] * M * H * W)
        x = x.flatten(2).transpose(1, 2)
        x = self.qact_before_norm(x)
        if isinstance(self.norm, nn.Identity):
            x = self.norm(x)
        else:
            x = self.norm(x, self.qact_before_norm.quantizer, self.qact.quantizer)
        return self.qact(x)


class Block(nn.Module):

    def __init__(self, dim, num_heads, mlp_ratio=4.0, qkv_bias=False, qk_scale=None, drop=0.0, attn_drop=0.0,
                 drop_path=0.0, act_layer=nn.GELU, norm_layer=nn.LayerNorm, quant=False, calibrate=False, cfg=None):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale, attn_drop=attn_drop,
                              proj_drop=drop, cfg=cfg)
        self.drop_path = nn.Identity()          # inference path: stochastic depth is the identity in eval mode
        self.qact2 = _qact(cfg, quant, calibrate, ln=True)
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop,
                       quant=quant, calibrate=calibrate, cfg=cfg)
        self.qact4 = _qact(cfg, quant, calibrate, ln=True)

    def forward(self, x, last_quantizer=None, FLOPs=None, global_distance=None, local_bit_config=None, plot=False,
                quant=False, hessian_statistic=False):
        FLOPs = [] if FLOPs is None else FLOPs
        global_distance = [] if global_distance is None else global_distance
        atten_bit_config = local_bit_config[0:2] if local_bit_config else None
        if atten_bit_config is not None and -1 in atten_bit_config:
            self.norm1.mode = 'ln'
        x = self.qact2(x + self.drop_path(self.attn(
            self.norm1(x, last_quantizer, self.attn.qact0.quantizer, self.attn.channel_scale), FLOPs, global_distance,
            atten_bit_config, plot=False, quant=quant, hessian_statistic=hessian_statistic)))
        ffn_bit_config = local_bit_config[2:4] if local_bit_config else None
        if ffn_bit_config is not None and -1 in ffn_bit_config:
            self.norm2.mode = 'ln'
        # NOTE the attention's channel scale is passed here, not the MLP's (reference quirk, vit_fquant.py:464)
        y = self.norm2(x, self.qact2.quantizer, self.mlp.qact0.quantizer, self.attn.channel_scale)
        y = self.mlp(y, FLOPs, global_distance, ffn_bit_config, plot, quant, hessian_statistic=hessian_statistic)
        return self.qact4(x + self.drop_path(y))


class VisionTransformer(nn.Module):

    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=1000, embed_dim=768, depth=12,
                 num_heads=12, mlp_ratio=4.0, qkv_bias=True, qk_scale=None, representation_size=None, drop_rate=0.0,
                 attn_drop_rate=0.0, drop_path_rate=0.0, hybrid_backbone=None, norm_layer=None, quant=False,
                 calibrate=False, input_quant=False, cfg=None):
        super().__init__()
        if hybrid_backbone is not None:
            raise NotImplementedError('HybridEmbed is unused by the reference factories (layers_quant.py:495-542)')
        self.num_classes = num_classes
        self.num_features = self.embed_dim = embed_dim
        norm_layer = norm_layer or partial(nn.LayerNorm, eps=1e-6)
        self.cfg = cfg
        self.quant = False
        self.input_quant = input_quant
        self.arch = dict(img_size=img_size, patch_size=patch_size, embed_dim=embed_dim, depth=depth, num_heads=num_heads,
                         num_classes=num_classes, mlp_ratio=mlp_ratio)
        self.in_chans = in_chans
        if input_quant:
            self.qact_input = _qact(cfg, quant, calibrate)
        self.patch_embed = PatchEmbed(img_size=img_size, patch_size=patch_size, in_chans=in_chans, embed_dim=embed_dim,
                                      quant=quant, calibrate=calibrate, cfg=cfg)
        num_patches = self.patch_embed.num_patches
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, num_patches + 1, embed_dim))
        self.pos_drop = nn.Dropout(p=drop_rate)
        self.qact_embed = _qact(cfg, quant, calibrate)
        self.qact_pos = _qact(cfg, quant, calibrate)
        self.qact1 = _qact(cfg, quant, calibrate, ln=True)
        self.blocks = nn.ModuleList([
            Block(dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale,
                  drop=drop_rate, attn_drop=attn_drop_rate, norm_layer=norm_layer, quant=quant, calibrate=calibrate,
                  cfg=cfg) for _ in range(depth)])
        self.depth = depth
        self.norm = norm_layer(embed_dim)
        self.qact2 = _qact(cfg, quant, calibrate)
        if representation_size:
            self.num_features = representation_size
            self.pre_logits = nn.Sequential(OrderedDict([('fc', nn.Linear(embed_dim, representation_size)), ('act', nn.Tanh())]))
        else:
            self.pre_logits = nn.Identity()
        self.head = _qlinear(cfg, self.num_features, num_classes, quant, calibrate) if num_classes > 0 else nn.Identity()
        self.act_out = _qact(cfg, quant, calibrate)
        nn.init.trunc_normal_(self.pos_embed, std=0.02)
        nn.init.trunc_normal_(self.cls_token, std=0.02)
        self.apply(self._init_weights)
        self._plan = None
        self._qmods = None
        self._lnmods = None
        self.capture_taps = False      # True: quantized forwards also fill blocks[i].attn.qkv_output / .mlp.fc1_output (fp32)

    def _init_weights(self, m):
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=0.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    @torch.jit.ignore
    def no_weight_decay(self):
        return {'pos_embed', 'cls_token'}

    def get_classifier(self):
        return self.head

    # ---- state switches (vit_fquant.py:667-698) ----------------------------------------------------------------
    def __setattr__(self, name, value):
        # replacing a submodule (e.g. swapping the head) invalidates the cached module lists and the frozen plan; the reference walks
        # self.modules() on every state switch (vit_fquant.py:667-698)
        if isinstance(value, nn.Module) and '_qmods' in self.__dict__:
            self.__dict__['_qmods'] = self.__dict__['_lnmods'] = self.__dict__['_plan'] = self.__dict__['_flops'] = None
        super().__setattr__(name, value)

    def _q_modules(self, refresh=False):
        if self._qmods is None or refresh:    # cached for the per-forward check only; every state switch below walks the tree again
            self._qmods = [m for m in self.modules() if type(m) in (QConv2d, QLinear, QAct, QIntSoftmax)]
            self._lnmods = None
        return self._qmods

    def model_quant(self, flag='on'):
        if flag == 'on':
            self.quant = True
        for m in self._q_modules(refresh=True):
            m.quant = True
        if self.cfg.INT_NORM and flag != 'off':
            for m in self.modules():
                if type(m) is QIntLayerNorm:
                    m.mode = 'int'
        self._plan = None

    def model_dequant(self):
        for m in self._q_modules(refresh=True):
            m.quant = False

    def model_open_calibrate(self):
        for m in self._q_modules(refresh=True):
            m.calibrate = True
        self._plan = None

    def model_open_last_calibrate(self):
        for m in self._q_modules(refresh=True):
            m.last_calibrate = True

    def model_close_calibrate(self):
        for m in self._q_modules(refresh=True):
            m.calibrate = False

    def load_state_dict(self, *a, **k):
        self._plan = None                      # frozen integer weights are invalidated
        return super().load_state_dict(*a, **k)

    # ---- frozen integer plan -----------------------------------------------------------------------------------
    def export_calib(self):
        """calibration state in the nested-dict format of ``plan.FrozenPlan`` / ``calib_io``."""
        def sc(q):
            if q.quantizer.scale is None:
                raise RuntimeError('the model has not been calibrated: run the calibration sequence (harness.calibrate_model) before a quantized forward')
            return q.quantizer.scale.detach().float().cpu()

        def dic(l):
            return {k: v.detach().float().cpu() for k, v in l.quantizer.dic_scale.items()}

        c = {'patch_embed.proj': dic(self.patch_embed.proj),
             'patch_embed.qact': sc(self.patch_embed.qact), 'qact_embed': sc(self.qact_embed),
             'qact_pos': sc(self.qact_pos), 'qact1': sc(self.qact1), 'qact2': sc(self.qact2), 'head': dic(self.head),
             'act_out': sc(self.act_out)}
        if self.input_quant:       # input_quant=False (the reference's vit_large factory, vit_fquant.py:925): no input QAct to export
            c['qact_input'] = sc(self.qact_input)
        for i, blk in enumerate(self.blocks):
            p = 'blocks.%d.' % i
            for nm, m in ((p + 'attn', blk.attn), (p + 'mlp', blk.mlp)):
                c[nm + '.best_scale'] = [t.detach().float().cpu() for t in m.best_scale]
                c[nm + '.best_act_scale'] = [t.detach().float().cpu() for t in m.best_act_scale]
                c[nm + '.best_weight_scale'] = [{k: v.detach().float().cpu() for k, v in d.items()} for d in m.best_weight_scale]
            c[p + 'attn.qact1'] = sc(blk.attn.qact1)
            c[p + 'attn.qact_attn1'] = sc(blk.attn.qact_attn1)
            c[p + 'attn.qact2'] = sc(blk.attn.qact2)
            c[p + 'attn.proj'] = dic(blk.attn.proj)
            c[p + 'attn.qact3'] = sc(blk.attn.qact3)
            c[p + 'qact2'] = sc(blk.qact2)
            c[p + 'mlp.qact1'] = sc(blk.mlp.qact1)
            c[p + 'mlp.fc2'] = dic(blk.mlp.fc2)
            c[p + 'mlp.qact2'] = sc(blk.mlp.qact2)
            c[p + 'qact4'] = sc(blk.qact4)
        return c

    def freeze(self, device=None):
        """build (once) the integer plan the HIP engine executes; called lazily by the first quantized forward."""
        from .plan import FrozenPlan
        if not (self.cfg.INT_NORM and self.cfg.INT_SOFTMAX):
            raise NotImplementedError('the HIP engine implements the ptf=True, lis=True configuration')
        dev = device or self.cls_token.device
        sd = {k: v for k, v in self.state_dict().items()}
        self._plan = FrozenPlan(self.arch, sd, self.export_calib(), device=dev, in_chans=self.in_chans, input_quant=self.input_quant)
        return self._plan

    def flops(self):
        """the FLOPs list the reference appends layer by layer (MAC counts; layers_quant.py:482,329,344;
        vit_fquant.py:304,336,794).  A function of the architecture only: computed once, a fresh list per call (callers may mutate it)."""
        cached = self.__dict__.get('_flops')
        if cached is None:
            a = self.arch
            D, P = a['embed_dim'], a['patch_size']
            g = a['img_size'] // P
            N, Hd = g * g + 1, int(D * a['mlp_ratio'])
            cached = [self.in_chans * P * P * D * g * g]
            for _ in range(self.depth):
                cached += [N * D * 3 * D, N * D * D, N * D * Hd, N * Hd * D]
            cached.append(D * self.num_classes)
            self.__dict__['_flops'] = cached
        return list(cached)

    # ---- forward -----------------------------------------------------------------------------------------------
    def forward_features(self, x, FLOPs, global_distance, bit_config, global_plot, hessian_statistic=False):
        B = x.shape[0]
        if self.input_quant:
            x = self.qact_input(x)
        patch_bit = bit_config[0] if bit_config else None
        x = self.patch_embed(x, FLOPs, patch_bit)
        x = torch.cat((self.cls_token.expand(B, -1, -1), x), dim=1)
        x = self.qact_embed(x)
        x = x + self.qact_pos(self.pos_embed)
        x = self.qact1(x)
        x = self.pos_drop(x)
        for i, blk in enumerate(self.blocks):
            local_bit_config = bit_config[i * 4 + 1:i * 4 + 5] if bit_config else None
            last_quantizer = self.qact1.quantizer if i == 0 else self.blocks[i - 1].qact4.quantizer
            x = blk(x, last_quantizer, FLOPs, global_distance, local_bit_config, False, self.quant, hessian_statistic)
        x = self.norm(x, self.blocks[-1].qact4.quantizer, self.qact2.quantizer)[:, 0]
        x = self.qact2(x)
        return self.pre_logits(x)

    def _fused(self):
        """the fused engine replaces the module-by-module graph only in the state ``model_quant()`` leaves behind: the model
        flag AND every Q-module's own ``.quant`` set (the reference's forward reads the per-module flags, so
        ``model_dequant()`` or a single ``m.quant = False`` sends it down the float branch of that module) AND every
        QIntLayerNorm still in mode 'int' (a ``-1`` entry flips its block's norm to float for good, vit_fquant.py:429-430)."""
        if self._lnmods is None:
            self._lnmods = [m for m in self.modules() if type(m) is QIntLayerNorm]
        return (self.quant and all(m.quant and not m.calibrate for m in self._q_modules())
                and all(m.mode == 'int' for m in self._lnmods))

    def forward(self, x, bit_config=None, plot=False, hessian_statistic=False):
        # bit_config == -1 is the reference's per-layer fp32 fallback (layers.py:144,171: plain F.linear / F.conv2d; for qkv / fc1
        # the SmoothQuant branch `channel_scale == None or bit_config == -1`, vit_fquant.py:199, layers_quant.py:222; the block's
        # norm flips to F.layer_norm, vit_fquant.py:429-430,462-463).  Such a model is partly float by request: it runs the module
        # graph below, as in the reference, not the integer engine.
        has_fp = bit_config is not None and any(int(b) == -1 for b in bit_config)
        if self._fused() and not hessian_statistic and not has_fp:
            # ---- THE HOT PATH: fused HIP engine -----------------------------------------------------------------
            if bit_config is None:
                raise ValueError('None is not in list')          # bit_pool.index(None), vit_fquant.py:282
            if self._plan is None:
                self.freeze(x.device if x.is_cuda else None)
            if not self.capture_taps:
                bits = [int(b) for b in bit_config]
                if x.shape[0] >= 64:       # large batches: contiguous slices on up to four HIP streams (same logits, +4 ... +29 %: FrozenPlan.slice_sizes)
                    out = torch.empty(x.shape[0], self.num_classes, dtype=torch.float32, device=self._plan.device)
                    return self._plan.forward_streams(x, bits, out, 3), self.flops(), []
                return self._plan.forward(x, bits), self.flops(), []
            # activation taps for the analysis scripts (cka_utility.py:44-47): written by the GEMM epilogues of the same launches
            taps = {}
            out = self._plan.forward(x, [int(b) for b in bit_config], taps=taps)
            for blk, tq, tf in zip(self.blocks, taps['qkv_output'], taps['fc1_output']):
                blk.attn.qkv_output, blk.mlp.fc1_output = tq, tf
            return out, self.flops(), []
        FLOPs, global_distance = [], []
        x = self.forward_features(x, FLOPs, global_distance, bit_config, plot, hessian_statistic)
        B, C = x.shape
        head_bit = bit_config[-1] if bit_config else None
        x = self.head(x, global_distance, head_bit)
        FLOPs.append(C * x.shape[1])
        x = self.act_out(x)
        return x, FLOPs, global_distance


def _factory(name, embed_dim, depth, num_heads, input_quant=True):
    def make(pretrained=False, quant=False, calibrate=False, cfg=None, **kwargs):
        if cfg is None:
            from .config import Config
            cfg = Config()
        model = VisionTransformer(patch_size=16, embed_dim=embed_dim, depth=depth, num_heads=num_heads, mlp_ratio=4,
                                  qkv_bias=True, norm_layer=partial(QIntLayerNorm, eps=1e-6), quant=quant,
                                  calibrate=calibrate, input_quant=input_quant, cfg=cfg, **kwargs)
        if pretrained:       # vit_fquant.py:822-828: the checkpoint of the torch-hub cache (never fetched here: checkpoint.load_pretrained)
            from .checkpoint import load_pretrained
            load_pretrained(model, name)
        return model
    make.__name__ = name
    return make


deit_tiny_patch16_224 = _factory('deit_tiny_patch16_224', 192, 12, 3)
deit_small_patch16_224 = _factory('deit_small_patch16_224', 384, 12, 6)
deit_base_patch16_224 = _factory('deit_base_patch16_224', 768, 12, 12)
vit_base_patch16_224 = _factory('vit_base_patch16_224', 768, 12, 12)
vit_large_patch16_224 = _factory('vit_large_patch16_224', 1024, 24, 16, input_quant=False)

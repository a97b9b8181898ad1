"""EfficientNet-B3 features trunk (the module timm.create_model('efficientnet_b3', features_only=True,
out_indices=(0,1,2,3,4)) returns at /root/reference/foundation_model/src/model.py:283-288) on the
gfx950 kernels. Parameter containers are ordinary nn.Conv2d / nn.BatchNorm2d objects so state-dict
keys, shapes and initialisation match timm's; their forward() is never called.
"""
import math
import weakref
from typing import List

import torch
import torch.nn as nn

from . import ops

# (type, repeats, kernel, stride, expand, out_channels) after B3 width/depth scaling
B3_STAGES = [
    ("ds", 2, 3, 1, 1, 24),
    ("ir", 3, 3, 2, 6, 32),
    ("ir", 3, 5, 2, 6, 48),
    ("ir", 5, 3, 2, 6, 96),
    ("ir", 5, 5, 1, 6, 136),
    ("ir", 6, 5, 2, 6, 232),
    ("ir", 2, 3, 1, 6, 384),
]
B3_STEM = 40
B3_FEATURE_STAGES = (0, 1, 2, 4, 6)
B3_FEATURE_DIMS = [24, 32, 48, 136, 384]


class _Cfg:
    pass


# per-trunk launch tables of the batched weight cast (ctypes arrays: kept off the module so that copying or pickling a
# model never meets them)
_CASTERS = weakref.WeakKeyDictionary()


class SqueezeExcite(nn.Module):
    def __init__(self, chs, rd):
        super().__init__()
        self.conv_reduce = nn.Conv2d(chs, rd, 1, bias=True)
        self.conv_expand = nn.Conv2d(rd, chs, 1, bias=True)


class MBConv(nn.Module):
    """Depthwise-separable ('ds') or inverted-residual ('ir') block; NHWC in, NHWC out."""

    def __init__(self, kind, cin, cout, k, s, e):
        super().__init__()
        self.kind, self.k, self.s = kind, k, s
        self.has_skip = cin == cout and s == 1
        mid = cin * e if kind == "ir" else cin
        if kind == "ir":
            self.conv_pw = nn.Conv2d(cin, mid, 1, bias=False)
            self.bn1 = nn.BatchNorm2d(mid)
            self.conv_dw = nn.Conv2d(mid, mid, k, s, (k - 1) // 2, groups=mid, bias=False)
            self.bn2 = nn.BatchNorm2d(mid)
            self.se = SqueezeExcite(mid, int(round(cin * 0.25)))
            self.conv_pwl = nn.Conv2d(mid, cout, 1, bias=False)
            self.bn3 = nn.BatchNorm2d(cout)
        else:
            self.conv_dw = nn.Conv2d(cin, cin, k, s, (k - 1) // 2, groups=cin, bias=False)
            self.bn1 = nn.BatchNorm2d(cin)
            self.se = SqueezeExcite(cin, int(round(cin * 0.25)))
            self.conv_pw = nn.Conv2d(cin, cout, 1, bias=False)
            self.bn2 = nn.BatchNorm2d(cout)

    def weight_specs(self, want_t):
        """(matrix, want copy, want transpose, pad) per GEMM weight, in the order (expand,) project, SE-expand; the
        first entries are cast to the compute dtype, the SE one stays fp32 (transposed only)."""
        w2 = ops._w2d
        gemm = ([(w2(self.conv_pw.weight), True, want_t, None), (w2(self.conv_pwl.weight), True, want_t, None)]
                if self.kind == "ir" else [(w2(self.conv_pw.weight), True, want_t, None)])
        return gemm, (w2(self.se.conv_expand.weight), False, True, None)

    def forward(self, x, pre=None):
        cfg = _Cfg()
        cfg.kind, cfg.k, cfg.s, cfg.has_skip = self.kind, self.k, self.s, self.has_skip
        cfg.pre = pre
        se = self.se
        if self.kind == "ir":
            cfg.bn_exp, cfg.bn_dw, cfg.bn_proj = ops.BNState(self.bn1), ops.BNState(self.bn2), ops.BNState(self.bn3)
            return ops.MBConvFn.apply(x, self.conv_pw.weight, self.bn1.weight, self.bn1.bias, self.conv_dw.weight,
                                      self.bn2.weight, self.bn2.bias, se.conv_reduce.weight, se.conv_reduce.bias,
                                      se.conv_expand.weight, se.conv_expand.bias, self.conv_pwl.weight,
                                      self.bn3.weight, self.bn3.bias, cfg)
        cfg.bn_exp, cfg.bn_dw, cfg.bn_proj = None, ops.BNState(self.bn1), ops.BNState(self.bn2)
        return ops.MBConvFn.apply(x, None, None, None, self.conv_dw.weight, self.bn1.weight, self.bn1.bias,
                                  se.conv_reduce.weight, se.conv_reduce.bias, se.conv_expand.weight,
                                  se.conv_expand.bias, self.conv_pw.weight, self.bn2.weight, self.bn2.bias, cfg)


class EfficientNetB3Features(nn.Module):
    """forward(x: [B,3,H,W] float32) -> list of 5 feature maps, NCHW-shaped views of NHWC storage with
    [24,32,48,136,384] channels at strides 2/4/8/16/32, in the compute dtype."""

    def __init__(self, compute_dtype=torch.float32):
        super().__init__()
        self.compute_dtype = compute_dtype
        self.conv_stem = nn.Conv2d(3, B3_STEM, 3, 2, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(B3_STEM)
        stages = []
        cin = B3_STEM
        for (kind, reps, k, s, e, cout) in B3_STAGES:
            blocks = []
            for i in range(reps):
                blocks.append(MBConv(kind, cin, cout, k, s if i == 0 else 1, e))
                cin = cout
            stages.append(nn.Sequential(*blocks))
        self.blocks = nn.Sequential(*stages)
        self.feature_dims = list(B3_FEATURE_DIMS)
        self.reset_parameters()

    def reset_parameters(self):
        # timm's efficientnet_init_weights: conv ~ N(0, sqrt(2/fan_out)), BN weight 1 / bias 0, conv bias 0
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                fan_out = m.kernel_size[0] * m.kernel_size[1] * m.out_channels // m.groups
                nn.init.normal_(m.weight, 0.0, math.sqrt(2.0 / fan_out))
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)

    def _cast_block_weights(self):
        """Every block's GEMM weights in the compute dtype (+ transposes when a backward will follow) and the SE
        expand weights transposed, in two launches for the whole trunk. -> one dict per block, in block order."""
        want_t = torch.is_grad_enabled()
        blocks = [blk for stage in self.blocks for blk in stage]
        gemm_specs, se_specs, spans = [], [], []
        for blk in blocks:
            g, s_ = blk.weight_specs(want_t)
            spans.append((len(gemm_specs), len(g)))
            gemm_specs += g
            se_specs.append(s_)
        key = (self.compute_dtype, want_t)
        casters = _CASTERS.setdefault(self, {})
        cached = casters.get(key)
        if cached is None or not cached[0].matches(gemm_specs) or not cached[1].matches(se_specs):
            cached = (ops.WeightCaster(gemm_specs, self.compute_dtype), ops.WeightCaster(se_specs, torch.float32))
            casters[key] = cached
        gw, sw = cached[0](), cached[1]()
        out = []
        for (start, n), se_w in zip(spans, sw):
            d = {"se": se_w, "proj": gw[start + n - 1]}
            if n == 2:
                d["exp"] = gw[start]
            out.append(d)
        return out

    def forward_nhwc(self, x) -> List[torch.Tensor]:
        cfg = _Cfg()
        cfg.dtype, cfg.bn = self.compute_dtype, ops.BNState(self.bn1)
        h = ops.StemFn.apply(x, self.conv_stem.weight, self.bn1.weight, self.bn1.bias, cfg)
        pre = iter(self._cast_block_weights())
        feats = []
        for si, stage in enumerate(self.blocks):
            for blk in stage:
                h = blk(h, next(pre))
            if si in B3_FEATURE_STAGES:
                feats.append(h)
        return feats

    def forward(self, x) -> List[torch.Tensor]:
        return [f.permute(0, 3, 1, 2) for f in self.forward_nhwc(x)]

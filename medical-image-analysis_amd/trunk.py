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
# per-trunk BatchNorm-folded inference weights: (key of data pointers and versions of every source tensor, tensors)
_FOLDED = weakref.WeakKeyDictionary()


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

    # ------------------------------------------------------------------------------------------------
    # Inference (eval mode, no autograd): BatchNorm is a fixed per-channel affine, so it is folded into the 1x1
    # convolution in front of it (weights scaled per output channel, shift as the GEMM bias) and SiLU / the residual
    # run in that GEMM's epilogue: the expansion writes the activated tensor, the projection writes the block output.
    # Per block: expansion GEMM, depthwise conv, pooled sums, SE gate, projection GEMM; no BatchNorm launches, no
    # separate apply/residual pass, no prologue arithmetic in the depthwise kernel. (In training the statistics of a
    # layer exist only after its convolution has finished, so there the normalisation rides in the CONSUMER's prologue.)
    # ------------------------------------------------------------------------------------------------
    def _fold_tensors(self):
        tens = [self.conv_stem.weight]
        mods = [self.bn1] + [m for stage in self.blocks for blk in stage for m in blk.modules()
                             if isinstance(m, (nn.Conv2d, nn.BatchNorm2d))]
        for m in mods:
            if isinstance(m, nn.BatchNorm2d):
                tens += [m.weight, m.bias, m.running_mean, m.running_var]
            else:
                tens += [m.weight] + ([m.bias] if m.bias is not None else [])
        return tens

    def _build_folded(self):
        dt = self.compute_dtype

        def fold(bn):
            sc = (bn.weight.detach().float() * torch.rsqrt(bn.running_var.float() + bn.eps))
            return sc.contiguous(), (bn.bias.detach().float() - bn.running_mean.float() * sc).contiguous()

        def scaled(w2d, sc, pad=None):
            w = w2d.detach().float() * sc[:, None]
            if pad and pad > w.shape[1]:
                w = torch.nn.functional.pad(w, (0, pad - w.shape[1]))
            return w.to(dt).contiguous()

        w2 = ops._w2d
        s0, b0 = fold(self.bn1)
        out = [scaled(w2(self.conv_stem.weight), s0, 32), b0]
        for stage in self.blocks:
            for blk in stage:
                if blk.kind == "ir":
                    s1, b1 = fold(blk.bn1)
                    s2, b2 = fold(blk.bn2)
                    s3, b3 = fold(blk.bn3)
                    wexp, wproj = scaled(w2(blk.conv_pw.weight), s1), scaled(w2(blk.conv_pwl.weight), s3)
                else:
                    wexp, b1 = None, None
                    s2, b2 = fold(blk.bn1)
                    s3, b3 = fold(blk.bn2)
                    wproj = scaled(w2(blk.conv_pw.weight), s3)
                cm = blk.conv_dw.weight.shape[0]
                out += [wexp, b1, s2, b2, wproj, b3,
                        blk.conv_dw.weight.detach().float().reshape(cm, blk.k * blk.k).contiguous(),
                        w2(blk.se.conv_reduce.weight).detach().float().contiguous(),
                        blk.se.conv_reduce.bias.detach().float().contiguous(),
                        w2(blk.se.conv_expand.weight).detach().float().t().contiguous(),
                        blk.se.conv_expand.bias.detach().float().contiguous()]
        return out

    def _folded(self):
        """-> flat list of inference tensors: stem (weights, bias), then per block (wexp, b1, s2, b2, wproj, b3, wdw,
        wr, br, weT, be). Rebuilt when a source tensor was written or replaced; a rebuild writes INTO the tensors
        handed out before (same addresses), so a captured HIP graph that reads them follows weight updates."""
        cached = _FOLDED.get(self)
        tens = cached[2] if cached is not None else self._fold_tensors()
        # in-place writes by torch bump the version counters; this library's own kernels (AdamW, train-mode BatchNorm
        # finalize, graph replays of a training step) write through raw pointers and bump ops.WeightEpoch instead;
        # module.to()/cuda() replace the storages and come through _apply
        try:
            key = (self.compute_dtype, tuple([t._version for t in tens]), tens[0].data_ptr(), tens[-1].data_ptr(),
                   ops.WeightEpoch.n)
        except RuntimeError:     # inference tensors keep no version counter: nothing to compare, fold again
            return self._build_folded()
        if cached is not None and cached[0] == key:
            return cached[1]
        fresh = self._build_folded()
        if cached is not None and cached[0][0] == key[0] and len(cached[1]) == len(fresh) and all(
                (a is None) == (b is None) and (a is None or (a.shape == b.shape and a.device == b.device))
                for a, b in zip(cached[1], fresh)):
            for a, b in zip(cached[1], fresh):
                if a is not None:
                    a.copy_(b)
            fresh = cached[1]
        _FOLDED[self] = (key, fresh, tens)
        return fresh

    def _apply(self, fn, *args, **kwargs):
        _FOLDED.pop(self, None)      # .to() / .cuda() / .float(): the folded copies refer to the old storages
        return super()._apply(fn, *args, **kwargs)

    def refresh_inference_weights(self):
        """Bring the BatchNorm-folded inference weights up to date (a no-op when nothing changed). HIP-graph replays
        of the eval forward call this first: the fold is host-side bookkeeping, not part of the captured launches."""
        if not self.training:
            self._folded()

    def _infer_nhwc(self, x) -> List[torch.Tensor]:
        fw = self._folded()
        B = x.shape[0]
        col, OH, OW = ops.stem_im2col(x, self.compute_dtype)
        h, _ = ops.gemm_nt(col, fw[0], bias=fw[1], epi_act=ops.ACT_SILU)
        h = h.view(B, OH, OW, -1)
        feats, i = [], 2
        for si, stage in enumerate(self.blocks):
            for blk in stage:
                wexp, b1, s2, b2, wproj, b3, wdw, wr, br, weT, be = fw[i:i + 11]
                i += 11
                Bh, H, W, Cin = h.shape
                if wexp is not None:
                    a1, _ = ops.gemm_nt(h.view(Bh * H * W, Cin), wexp, bias=b1, epi_act=ops.ACT_SILU)
                    dw_in = a1.view(Bh, H, W, -1)
                else:
                    dw_in = h
                y2, _ = ops.dwconv_fwd(dw_in, wdw, blk.k, blk.s)
                _, OH2, OW2, Cm = y2.shape
                M2 = Bh * OH2 * OW2
                pool_part = ops.img_colsum(y2, None, s2, b2, ops.ACT_SILU)
                _, _, gate = ops.se_gate_fwd(pool_part, OH2 * OW2, wr, br, weT, be)
                out, _ = ops.gemm_nt(y2.view(M2, Cm), wproj, pscale=s2, pshift=b2, pro_act=ops.ACT_SILU, gate=gate,
                                     rows_per_img=OH2 * OW2, bias=b3, add=h.view(M2, -1) if blk.has_skip else None)
                h = out.view(Bh, OH2, OW2, -1)
            if si in B3_FEATURE_STAGES:
                feats.append(h)
        return feats

    def forward_nhwc(self, x) -> List[torch.Tensor]:
        if not self.training and not torch.is_grad_enabled() and x.is_cuda:
            bns = self.__dict__.get("_bn_list")
            if bns is None:      # plain attribute (not a registered submodule list): the layers are fixed at construction
                bns = self.__dict__["_bn_list"] = [m for m in self.modules() if isinstance(m, nn.BatchNorm2d)]
            if not any(m.training for m in bns):     # a layer left in train mode needs batch statistics
                return self._infer_nhwc(x)
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

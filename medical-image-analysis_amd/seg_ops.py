"""Segmentation-decoder arithmetic on the gfx950 kernels: 3x3 conv + BatchNorm + ReLU blocks (im2col lowering onto
the MFMA GEMMs), bilinear upsample + skip concat, the final 1x1 conv to NCHW fp32 logits, and the
cross-entropy + Dice / IoU loss. Each autograd Function is a forward kernel sequence plus its hand-written backward."""
import torch

from . import ops
from ._lib import lib
from .head_ops import _seed, colsum
from .ops import ACT_NONE, ACT_RELU, _DT, _p, _stream


def im2col3x3(x, *, scale=None, shift=None, act=ACT_NONE, gate=None):
    B, H, W, C = x.shape
    col = torch.empty((B * H * W, 9 * C), dtype=x.dtype, device=x.device)
    lib().mia_im2col3x3(_DT[x.dtype], _p(x), B, H, W, C, _p(scale), _p(shift), act, _p(gate), _p(col), _stream())
    return col


def col2im3x3(dcol, shape):
    B, H, W, C = shape
    dx = torch.empty(shape, dtype=dcol.dtype, device=dcol.device)
    lib().mia_col2im3x3(_DT[dcol.dtype], _p(dcol), B, H, W, C, _p(dx), _stream())
    return dx


def _conv_weight_matrix(w, dtype, want_t):
    """[Co, Ci, 3, 3] fp32 master -> ([Co, 9*Ci], [9*Ci, Co] or None) in the compute dtype, tap-major columns."""
    co, ci = w.shape[0], w.shape[1]
    m = w.detach().permute(0, 2, 3, 1).reshape(co, 9 * ci).contiguous()
    return ops.cast_weight(m, dtype, want_t=want_t)


def _weight_grad(dwm, like):
    co, ci = like.shape[0], like.shape[1]
    return dwm.view(co, 3, 3, ci).permute(0, 3, 1, 2).contiguous()


class DecoderBlockFn(torch.autograd.Function):
    """conv1(3x3, bias) -> bn1 -> ReLU -> Dropout2d -> conv2(3x3, bias) -> bn2 -> ReLU on an NHWC tensor.
    The first BatchNorm apply, ReLU and the channel-dropout mask are folded into the second conv's im2col pass."""

    @staticmethod
    def forward(ctx, x, w1, b1, g1, be1, w2, b2, g2, be2, cfg):
        x = x.contiguous()
        B, H, W, Cin = x.shape
        Co = w1.shape[0]
        M = B * H * W
        dt = x.dtype
        training = cfg.bn1.training
        need_bwd = any(ctx.needs_input_grad)
        rows = ops.part_rows((M + 63) // 64) if training else 0
        col1 = im2col3x3(x)
        w1m, w1mT = _conv_weight_matrix(w1, dt, need_bwd)
        y1, part = ops.gemm_nt(col1, w1m, stats_rows=rows)
        st1 = ops.bn_finalize_fwd(part, M, cfg.bn1, conv_bias=b1.detach())
        mask = None
        if cfg.training and cfg.pdrop > 0:
            mask = torch.empty((B, Co), dtype=torch.float32, device=x.device)
            lib().mia_channel_dropout_mask(_p(mask), B, Co, float(cfg.pdrop), _seed(), ops.seed_offset(), _stream())
        col2 = im2col3x3(y1.view(B, H, W, Co), scale=st1[0], shift=st1[1], act=ACT_RELU, gate=mask)
        w2m, w2mT = _conv_weight_matrix(w2, dt, need_bwd)
        y2, part = ops.gemm_nt(col2, w2m, stats_rows=rows)
        st2 = ops.bn_finalize_fwd(part, M, cfg.bn2, conv_bias=b2.detach())
        out = ops.affine2(y2, st2[0], st2[1], act=ACT_RELU).view(B, H, W, Co)
        ctx.training, ctx.shape = training, (B, H, W, Cin, Co)
        ctx.w1mT, ctx.w2mT = w1mT, w2mT
        ctx.save_for_backward(col1, y1, st1, mask, col2, y2, st2, w1, g1, w2, g2)
        return out

    @staticmethod
    def backward(ctx, dout):
        col1, y1, st1, mask, col2, y2, st2, w1, g1, w2, g2 = ctx.saved_tensors
        B, H, W, Cin, Co = ctx.shape
        M = B * H * W
        training = ctx.training
        # bn2 + ReLU backward (the masked gradient goes to a fresh tensor: the incoming one is autograd's)
        g, part = ops.bn_act_bwd_reduce(dout.contiguous().view(M, Co), y2, st2, ACT_RELU)
        c2 = ops.bn_finalize_bwd(part, M, g2, st2, training)
        dy2 = ops.affine2(g, c2[2], c2[4], y=y2, b=c2[3], out=g)
        dw2 = ops.gemm_tn(dy2, col2)
        dcol2, _ = ops.gemm_nt(dy2, ctx.w2mT)
        ga = col2im3x3(dcol2, (B, H, W, Co)).view(M, Co)
        # Dropout2d mask, ReLU and bn1 backward
        ga, part = ops.bn_act_bwd_reduce(ga, y1, st1, ACT_RELU, gate=mask, rows_per_img=H * W, out=ga)
        c1 = ops.bn_finalize_bwd(part, M, g1, st1, training)
        dy1 = ops.affine2(ga, c1[2], c1[4], y=y1, b=c1[3], out=ga)
        dw1 = ops.gemm_tn(dy1, col1)
        dx = None
        if ctx.needs_input_grad[0]:
            dcol1, _ = ops.gemm_nt(dy1, ctx.w1mT)
            dx = col2im3x3(dcol1, (B, H, W, Cin))
        # conv-bias gradients (column sums of dy1 / dy2) come out of the BatchNorm backward finalize in closed form
        return (dx, _weight_grad(dw1, w1), c1[5], c1[0], c1[1], _weight_grad(dw2, w2), c2[5], c2[0], c2[1], None)


class UpsampleConcatFn(torch.autograd.Function):
    """F.interpolate(x, size, 'bilinear', align_corners=False) then cat([x, skip], channel) on NHWC tensors."""

    @staticmethod
    def forward(ctx, x, skip, H, W):
        x = x.contiguous()
        B, h, w, C1 = x.shape
        C2 = 0
        if skip is not None:
            skip = skip.contiguous()
            C2 = skip.shape[-1]
            assert skip.shape[:3] == (B, H, W) and skip.dtype == x.dtype, (skip.shape, x.shape)
        out = torch.empty((B, H, W, C1 + C2), dtype=x.dtype, device=x.device)
        lib().mia_upsample_concat_fwd(_DT[x.dtype], _p(x), B, h, w, C1, _p(skip), C2, H, W, _p(out), _stream())
        ctx.dims = (B, h, w, C1, C2, H, W)
        return out

    @staticmethod
    def backward(ctx, dout):
        B, h, w, C1, C2, H, W = ctx.dims
        dout = dout.contiguous()
        dx = torch.empty((B, h, w, C1), dtype=dout.dtype, device=dout.device)
        dskip = torch.empty((B, H, W, C2), dtype=dout.dtype, device=dout.device) if C2 else None
        lib().mia_upsample_concat_bwd(_DT[dout.dtype], _p(dout), B, h, w, C1, C2, H, W, _p(dx), _p(dskip), _stream())
        return dx, dskip, None, None


class PixelLinearFn(torch.autograd.Function):
    """nn.Conv2d(C, n, 1) with bias on an NHWC map -> NCHW fp32 logits [B, n, H, W]."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        x = x.contiguous()
        B, H, W, C = x.shape
        n = weight.shape[0]
        wt = weight.detach().reshape(n, C).contiguous()
        logits = torch.empty((B, n, H, W), dtype=torch.float32, device=x.device)
        lib().mia_pixel_linear_fwd(_DT[x.dtype], _p(x), B, H * W, C, n, _p(wt), _p(bias.detach() if bias is not None else None),
                                   _p(logits), _stream())
        ctx.has_bias = bias is not None
        ctx.save_for_backward(x, weight)
        return logits

    @staticmethod
    def backward(ctx, dl):
        x, weight = ctx.saved_tensors
        B, H, W, C = x.shape
        n = weight.shape[0]
        dl = dl.contiguous().float()
        wt = weight.detach().reshape(n, C).contiguous()
        rows = max(1, min(512, (B * H * W + 1023) // 1024))
        partial = torch.empty((rows, n * C + n), dtype=torch.float32, device=x.device)
        dx = torch.empty_like(x)
        lib().mia_pixel_linear_bwd(_DT[x.dtype], _p(x), _p(dl), B, H * W, C, n, _p(wt), _p(dx), _p(partial), rows, _stream())
        sums = colsum(partial)
        dw = sums[:n * C].view_as(weight)
        return dx, dw, (sums[n * C:] if ctx.has_bias else None)


class _Cfg:
    pass


def decoder_block_forward(m, x):
    cfg = _Cfg()
    cfg.bn1, cfg.bn2 = ops.BNState(m.bn1), ops.BNState(m.bn2)
    cfg.pdrop, cfg.training = float(m.dropout.p), m.training
    return DecoderBlockFn.apply(x, m.conv1.weight, m.conv1.bias, m.bn1.weight, m.bn1.bias, m.conv2.weight, m.conv2.bias,
                                m.bn2.weight, m.bn2.bias, cfg)


def decoder_forward(m, feats_nhwc, input_size):
    """UNetDecoder / EnhancedUNetDecoder forward (src/model.py:216-237, :450-468) on NHWC trunk features."""
    feats = list(feats_nhwc)[::-1]
    x = None
    cbams = getattr(m, "cbam_modules", None)
    for i, blk in enumerate(m.decoder_blocks):
        if i == 0:
            x = blk(feats[0])
        else:
            skip = feats[i]
            x = blk(UpsampleConcatFn.apply(x, skip, skip.shape[1], skip.shape[2]))
        if cbams is not None:
            x = cbams[i](x)
    x = UpsampleConcatFn.apply(x, None, int(input_size[0]), int(input_size[1]))
    return PixelLinearFn.apply(x, m.final_conv.weight, m.final_conv.bias)


# ------------------------------------------------------------------------------------------------
class SegLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, class_weights, iou, smooth, ce_w, ov_w):
        logits = logits.float().contiguous()
        target = target.long().contiguous()
        B, n, H, W = logits.shape
        HW = H * W
        assert target.shape == (B, H, W), (target.shape, logits.shape)
        cw = class_weights.detach().float().contiguous() if class_weights is not None else None
        splits = max(1, min(HW // 256 if HW >= 256 else 1, max(1, 2048 // B)))
        dev = logits.device
        partial = torch.empty((B * splits, 3 * n + 2), dtype=torch.float32, device=dev)
        coef = torch.empty((1 + 2 * B * n,), dtype=torch.float32, device=dev)
        loss = torch.empty((1,), dtype=torch.float32, device=dev)
        lib().mia_seg_loss_fwd(_p(logits), _p(target), B, HW, n, _p(cw), int(iou), float(smooth), float(ce_w),
                               float(ov_w), splits, _p(partial), _p(coef), _p(loss), _stream())
        ctx.save_for_backward(logits, target, cw, coef)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        logits, target, cw, coef = ctx.saved_tensors
        B, n, H, W = logits.shape
        dl = torch.empty_like(logits)
        up = g.detach().float().reshape(1).contiguous()
        lib().mia_seg_loss_bwd(_p(logits), _p(target), B, H * W, n, _p(cw), _p(coef), _p(up), _p(dl), _stream())
        return dl, None, None, None, None, None, None


def seg_loss(logits, target, *, class_weights=None, iou=False, smooth=1e-5, ce_weight=0.5, overlap_weight=0.5):
    return SegLossFn.apply(logits, target, class_weights, iou, smooth, ce_weight, overlap_weight)


class GeneralCEFn(torch.autograd.Function):
    """Softmax cross-entropy with class weights or focal modulation on [B,n] or [B,n,...] fp32 logits."""

    @staticmethod
    def forward(ctx, logits, target, class_weights, focal, alpha, gamma, reduction):
        logits = logits.float().contiguous()
        target = target.long().contiguous()
        B, n = logits.shape[0], logits.shape[1]
        P = logits[0, 0].numel() if logits.dim() > 2 else 1
        assert target.numel() == B * P, (target.shape, logits.shape)
        dev = logits.device
        cw = class_weights.detach().float().contiguous().to(dev) if class_weights is not None else None
        rows = max(1, min(1024, (B * P + 1023) // 1024))
        partial = torch.empty((rows, 2), dtype=torch.float32, device=dev)
        out = torch.empty((2,), dtype=torch.float32, device=dev)
        per = torch.empty(target.shape, dtype=torch.float32, device=dev) if reduction == "none" else None
        lib().mia_ce_general_fwd(_p(logits), n * P, P, 1, _p(target), B, P, n, _p(cw), int(focal), float(alpha),
                                 float(gamma), 1 if reduction == "mean" else 0, _p(per), _p(partial), rows, _p(out),
                                 _stream())
        ctx.meta = (B, P, n, int(focal), float(alpha), float(gamma), reduction)
        ctx.save_for_backward(logits, target, cw, out)
        return per if reduction == "none" else out[0]

    @staticmethod
    def backward(ctx, g):
        logits, target, cw, out = ctx.saved_tensors
        B, P, n, focal, alpha, gamma, reduction = ctx.meta
        dl = torch.empty_like(logits)
        per_elem = reduction == "none"
        up = g.detach().float().contiguous() if per_elem else g.detach().float().reshape(1).contiguous()
        lib().mia_ce_general_bwd(_p(logits), n * P, P, 1, _p(target), B, P, n, _p(cw), focal, alpha, gamma,
                                 None if per_elem else _p(out), _p(up), 1 if per_elem else 0, _p(dl), _stream())
        return dl, None, None, None, None, None, None


def general_ce(logits, target, *, class_weights=None, focal=False, alpha=1.0, gamma=2.0, reduction="mean"):
    if reduction not in ("mean", "sum", "none"):
        raise ValueError(f"unknown reduction {reduction!r}")
    return GeneralCEFn.apply(logits, target, class_weights, focal, alpha, gamma, reduction)

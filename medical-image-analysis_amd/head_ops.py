"""Head arithmetic on the gfx950 kernels: CBAM, multi-head self-attention, pooled LayerNorm/MLP classifier,
cross-entropy. Each autograd Function below is a forward kernel sequence plus its hand-written backward."""
import ctypes
import math

import torch
import torch.nn.functional as F

from . import ops
from ._lib import lib
from .ops import ACT_NONE, ACT_RELU, _DT, _p, _stream


def _seed() -> int:
    # host-side draw from torch's CPU generator: reproducible under torch.manual_seed, no device sync
    return int(torch.randint(0, 2 ** 62, (1,)).item())


def colsum(x2d):
    R, C = x2d.shape
    out = torch.empty((C,), dtype=torch.float32, device=x2d.device)
    chunks = max(1, min(128, R // 64))
    ws = torch.empty((chunks * C,), dtype=torch.float32, device=x2d.device) if chunks > 1 else None
    lib().mia_colsum(_DT[x2d.dtype], _p(x2d), R, C, x2d.stride(0), _p(out), _p(ws), chunks * C if chunks > 1 else 0,
                     _stream())
    return out


# ------------------------------------------------------------------------------------------------
class CBAMFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, W1, b1, W2, b2, w7, bn_g, bn_b, bn):
        x = x.contiguous()
        B, H, W, C = x.shape
        HW, HID, dev = H * W, W1.shape[0], x.device
        f32 = dict(dtype=torch.float32, device=dev)
        avg, mx, cscale = (torch.empty((B, C), **f32) for _ in range(3))
        amax = torch.empty((B, C), dtype=torch.int32, device=dev)
        ha, hm = torch.empty((B, HID), **f32), torch.empty((B, HID), **f32)
        nws = lib().mia_cbam_ws_floats(B, H, W, C, HID, 0)
        ws = torch.empty((nws,), **f32) if nws > 0 else None
        lib().mia_cbam_channel_fwd(_DT[x.dtype], _p(x), B, HW, C, HID, _p(W1), _p(b1), _p(W2), _p(b2), _p(avg), _p(mx),
                                   _p(amax), _p(ha), _p(hm), _p(cscale), _p(ws), _stream())
        comp = torch.empty((B, HW, 2), **f32)
        cmax = torch.empty((B, HW), dtype=torch.int32, device=dev)
        pre = torch.empty((B, HW), **f32)
        rows = max(1, min(1024, (B * HW + 255) // 256))
        part = torch.empty((rows, 2, 1), **f32)
        w7c = w7.detach().reshape(2, 49).contiguous()
        lib().mia_cbam_spatial_fwd(_DT[x.dtype], _p(x), _p(cscale), B, H, W, C, _p(w7c), _p(comp), _p(cmax), _p(pre),
                                   _p(part), rows, _stream())
        st = ops.bn_finalize_fwd(part if bn.training else None, B * HW, bn)  # [4,1]: scale, shift, mean, rstd
        out = torch.empty_like(x)
        sgate = torch.empty((B, HW), **f32)
        lib().mia_cbam_apply(_DT[x.dtype], _p(x), _p(cscale), _p(pre), _p(st), B, HW, C, _p(out), _p(sgate), _stream())
        ctx.training = bn.training
        ctx.save_for_backward(x, cscale, sgate, pre, st, bn_g, comp, cmax, amax, avg, mx, ha, hm, W1, W2, w7c)
        return out

    @staticmethod
    def backward(ctx, dx2):
        x, cscale, sgate, pre, st, bn_g, comp, cmax, amax, avg, mx, ha, hm, W1, W2, w7c = ctx.saved_tensors
        dx2 = dx2.contiguous()
        B, H, W, C = x.shape
        HID, dev = W1.shape[0], x.device
        f32 = dict(dtype=torch.float32, device=dev)
        nws = lib().mia_cbam_ws_floats(B, H, W, C, HID, 1)
        if nws < 0:
            raise ValueError("CBAM backward workspace does not fit: reduce the batch")
        ws = torch.empty((nws,), **f32)
        dx = torch.empty_like(x)
        dW1, db1 = torch.empty_like(W1), torch.empty((HID,), **f32)
        dW2, db2 = torch.empty_like(W2), torch.empty((C,), **f32)
        dw7 = torch.empty((1, 2, 7, 7), **f32)
        dg, db = torch.empty((1,), **f32), torch.empty((1,), **f32)
        lib().mia_cbam_bwd(_DT[x.dtype], _p(dx2), _p(x), B, H, W, C, HID, _p(cscale), _p(sgate), _p(pre), _p(st),
                           _p(bn_g), 1 if ctx.training else 0, _p(comp), _p(cmax), _p(amax), _p(avg), _p(mx), _p(ha),
                           _p(hm), _p(W1), _p(W2), _p(w7c), _p(ws), _p(dx), _p(dW1), _p(db1), _p(dW2), _p(db2),
                           _p(dw7), _p(dg), _p(db), _stream())
        return dx, dW1, db1, dW2, db2, dw7, dg, db, None


def cbam_forward(m, x):
    cg, sp = m.ChannelGate.mlp, m.SpatialGate.spatial
    return CBAMFn.apply(x, cg[1].weight, cg[1].bias, cg[3].weight, cg[3].bias, sp.conv.weight, sp.bn.weight,
                        sp.bn.bias, ops.BNState(sp.bn))


# ------------------------------------------------------------------------------------------------
class LinearFn(torch.autograd.Function):
    """y = act(x @ W^T + b) (+ res), x [M,K] in the compute dtype, W/b fp32 masters."""

    @staticmethod
    def forward(ctx, x, weight, bias, res, act):
        x = x.contiguous()
        w, _ = ops.cast_weight(weight.detach(), x.dtype)
        y, _ = ops.gemm_nt(x, w, bias=bias.detach() if bias is not None else None, epi_act=act,
                           add=res.contiguous() if res is not None else None)
        ctx.act, ctx.has_res, ctx.has_bias = act, res is not None, bias is not None
        ctx.save_for_backward(x, weight, y if act == ACT_RELU else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, y = ctx.saved_tensors
        dy = dy.contiguous()
        dres = dy if ctx.has_res else None
        if ctx.act == ACT_RELU:
            # ReLU mask (fp32 MLP only; a residual is never combined with ReLU in this model)
            dy = dy.float().clone()
            lib().mia_dropout_f32(_p(dy), _p(y.float()), dy.numel(), 0.0, 0, None, _stream())
            dy = dy.to(x.dtype)
        N, K = weight.shape
        vec = 8 if x.dtype == torch.bfloat16 else 4
        npad = (N + vec - 1) // vec * vec
        if npad != N:
            dy = F.pad(dy, (0, npad - N))
        _, wT = ops.cast_weight(weight.detach(), x.dtype, want=False, want_t=True) if npad == N else \
            _padded_transpose(weight.detach(), x.dtype, npad)
        dx, _ = ops.gemm_nt(dy, wT)
        dw = ops.gemm_tn(dy, x)[:N]
        db = colsum(dy)[:N] if ctx.has_bias else None
        return dx, dw, db, dres, None


def _padded_transpose(w, dtype, npad):
    N, K = w.shape
    dstT = torch.empty((K, npad), dtype=dtype, device=w.device)
    lib().mia_cast_pad_transpose(_DT[dtype], _p(w), N, K, None, K, _p(dstT), npad, _stream())
    return None, dstT


class AttentionFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, B, HW, C, heads, pdrop, seed):
        qkv = qkv.contiguous()
        P = torch.empty((B, heads, HW, HW), dtype=torch.float32, device=qkv.device)
        out = torch.empty((B * HW, C), dtype=qkv.dtype, device=qkv.device)
        lib().mia_attention_fwd(_DT[qkv.dtype], _p(qkv), B, HW, C, heads, float(pdrop), seed, ops.seed_offset(), _p(P),
                                _p(out), _stream())
        ctx.meta = (B, HW, C, heads, float(pdrop), seed, ops.seed_offset())
        ctx.save_for_backward(qkv, P, out)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, P, out = ctx.saved_tensors
        B, HW, C, heads, pdrop, seed, soff = ctx.meta
        dqkv = torch.empty_like(qkv)
        lib().mia_attention_bwd(_DT[qkv.dtype], _p(qkv), _p(P), _p(dout.contiguous()), _p(out), B, HW, C, heads, pdrop,
                                seed, soff, _p(dqkv), _stream())
        return dqkv, None, None, None, None, None, None


class PoolNormFn(torch.autograd.Function):
    """AdaptiveAvgPool2d(1) -> Flatten -> [LayerNorm] -> Dropout on an NHWC map; fp32 [B,C] out."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, pdrop, seed):
        x = x.contiguous()
        B, H, W, C = x.shape
        part = ops.img_colsum(x, None, None, None, ACT_NONE)
        f32 = dict(dtype=torch.float32, device=x.device)
        out, xhat, rstd = torch.empty((B, C), **f32), torch.empty((B, C), **f32), torch.empty((B,), **f32)
        lib().mia_pool_ln_dropout_fwd(_p(part), B, part.shape[1], H * W, C, _p(gamma), _p(beta), float(eps),
                                      float(pdrop), seed, ops.seed_offset(), _p(out), _p(xhat), _p(rstd), _stream())
        ctx.meta = (B, H, W, C, float(pdrop), seed, x.dtype, ops.seed_offset())
        ctx.save_for_backward(gamma, xhat, rstd)
        return out

    @staticmethod
    def backward(ctx, dout):
        gamma, xhat, rstd = ctx.saved_tensors
        B, H, W, C, pdrop, seed, dt, soff = ctx.meta
        f32 = dict(dtype=torch.float32, device=dout.device)
        dpooled = torch.empty((B, C), **f32)
        dg_rows = torch.empty((B, C), **f32) if gamma is not None else None
        db_rows = torch.empty((B, C), **f32) if gamma is not None else None
        lib().mia_pool_ln_dropout_bwd(_p(dout.contiguous()), B, C, _p(gamma), _p(xhat), _p(rstd), pdrop, seed, soff,
                                      _p(dpooled), _p(dg_rows), _p(db_rows), _stream())
        dx = torch.empty((B, H, W, C), dtype=dt, device=dout.device)
        lib().mia_broadcast_rows(_DT[dt], _p(dpooled), B, H * W, C, 1.0 / (H * W), None, _p(dx), _stream())
        if gamma is None:
            return dx, None, None, None, None, None
        return dx, colsum(dg_rows), colsum(db_rows), None, None, None


class DropoutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, pdrop, seed):
        y = x.clone()
        lib().mia_dropout_f32(_p(y), None, y.numel(), float(pdrop), seed, ops.seed_offset(), _stream())
        ctx.meta = (float(pdrop), seed, ops.seed_offset())
        return y

    @staticmethod
    def backward(ctx, dy):
        g = dy.clone()
        lib().mia_dropout_f32(_p(g), None, g.numel(), ctx.meta[0], ctx.meta[1], ctx.meta[2], _stream())
        return g, None, None


def _mlp(x32, fc1, fc2, pdrop, training):
    h = LinearFn.apply(x32, fc1.weight, fc1.bias, None, ACT_RELU)
    if training and pdrop > 0:
        h = DropoutFn.apply(h, pdrop, _seed())
    return LinearFn.apply(h, fc2.weight, fc2.bias, None, ACT_NONE)


def mha_head_forward(m, x):
    x2 = cbam_forward(m.cbam, x)
    B, H, W, C = x2.shape
    HW = H * W
    t = x2.view(B * HW, C)
    training = m.training
    wqkv = torch.cat((m.query.weight, m.key.weight, m.value.weight), 0)
    bqkv = torch.cat((m.query.bias, m.key.bias, m.value.bias), 0)
    qkv = LinearFn.apply(t, wqkv, bqkv, None, ACT_NONE)
    p_att = m.attention_dropout.p if training else 0.0
    a = AttentionFn.apply(qkv, B, HW, C, m.num_heads, p_att, _seed() if p_att > 0 else 0)
    x3 = LinearFn.apply(a, m.output_projection.weight, m.output_projection.bias, t, ACT_NONE).view(B, H, W, C)
    c = m.classifier
    p1 = c[3].p if training else 0.0
    z = PoolNormFn.apply(x3, c[2].weight, c[2].bias, c[2].eps, p1, _seed() if p1 > 0 else 0)
    return _mlp(z, c[4], c[7], c[6].p, training)


def simple_head_forward(m, x):
    training = m.training
    p1 = m[2].p if training else 0.0
    z = PoolNormFn.apply(x, None, None, 0.0, p1, _seed() if p1 > 0 else 0)
    return _mlp(z, m[3], m[6], m[5].p, training)


class CrossEntropyFn(torch.autograd.Function):
    """nn.CrossEntropyLoss() (mean reduction) on fp32 logits [B,n] and int64 labels."""

    @staticmethod
    def forward(ctx, logits, labels):
        logits = logits.float().contiguous()
        B, n = logits.shape
        loss = torch.empty((1,), dtype=torch.float32, device=logits.device)
        dl = torch.empty((B, n), dtype=torch.float32, device=logits.device)
        lib().mia_softmax_ce(_p(logits), n, _p(labels.contiguous()), B, n, _p(loss), _p(dl), n, _stream())
        ctx.save_for_backward(dl)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (dl,) = ctx.saved_tensors
        return dl * g, None


class CrossEntropyLoss(torch.nn.Module):
    """Drop-in for the classification loss get_loss_function('classification') returns (src/loss.py:143-147)."""

    def forward(self, logits, labels):
        return CrossEntropyFn.apply(logits, labels.long())

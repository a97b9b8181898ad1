"""Head arithmetic. INTERIM: composed from torch device ops until the dedicated gfx950 head kernels land."""
import math

import torch
import torch.nn.functional as F


def cbam_forward(m, x):  # x NHWC
    xf = x.float()
    mlp = m.ChannelGate.mlp
    att = mlp(xf.mean((1, 2))) + mlp(xf.amax((1, 2)))
    x1 = xf * torch.sigmoid(att)[:, None, None, :]
    comp = torch.stack((x1.amax(3), x1.mean(3)), dim=1)  # [B,2,H,W]
    sp = m.SpatialGate.spatial
    s = sp.bn(sp.conv(comp))  # [B,1,H,W]
    return x1 * torch.sigmoid(s)[:, 0, :, :, None]


def mha_head_forward(m, x):
    x2 = cbam_forward(m.cbam, x)
    B, H, W, C = x2.shape
    t = x2.reshape(B, H * W, C)
    sp = lambda y: y.reshape(B, H * W, m.num_heads, m.head_dim).transpose(1, 2)
    q, k, v = sp(m.query(t)), sp(m.key(t)), sp(m.value(t))
    w = F.softmax(q @ k.transpose(-2, -1) / math.sqrt(m.head_dim), dim=-1)
    w = m.attention_dropout(w)
    a = (w @ v).transpose(1, 2).reshape(B, H * W, C)
    x3 = t + m.output_projection(a)
    pooled = x3.mean(1)
    c = m.classifier
    h = c[3](c[2](pooled))
    h = c[6](c[5](c[4](h)))
    return c[7](h)


def simple_head_forward(m, x):
    pooled = x.float().mean((1, 2))
    h = m[5](m[4](m[3](m[2](pooled))))
    return m[6](h)

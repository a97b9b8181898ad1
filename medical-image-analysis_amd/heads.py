"""Classification heads of the reference on the gfx950 kernels, NHWC in.

State-dict keys equal the reference's (src/model.py:11-175, :308-316):
  cbam.ChannelGate.mlp.{1,3}.*, cbam.SpatialGate.spatial.{conv.weight,bn.*},
  query/key/value/output_projection.*, classifier.{2,4,7}.* (attention head);
  {3,6}.* (simple head, an nn.Sequential).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


class _BasicConv(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv = nn.Conv2d(2, 1, 7, 1, 3, bias=False)
        self.bn = nn.BatchNorm2d(1, eps=1e-5, momentum=0.01, affine=True)


class _ChannelGate(nn.Module):
    def __init__(self, c, r=16):
        super().__init__()
        self.mlp = nn.Sequential(nn.Identity(), nn.Linear(c, c // r), nn.ReLU(), nn.Linear(c // r, c))


class _SpatialGate(nn.Module):
    def __init__(self):
        super().__init__()
        self.spatial = _BasicConv()


class CBAM(nn.Module):
    """Channel gate then spatial gate (src/model.py:96-108); x: NHWC."""

    def __init__(self, c, r=16):
        super().__init__()
        self.ChannelGate = _ChannelGate(c, r)
        self.SpatialGate = _SpatialGate()

    def forward(self, x):
        from . import head_ops
        return head_ops.cbam_forward(self, x)


class MultiHeadAttentionClassifier(nn.Module):
    """CBAM -> 8-head self-attention over H*W tokens with residual -> avgpool -> LayerNorm -> MLP
    (src/model.py:112-175); x: NHWC feature map, returns fp32 logits [B, num_classes]."""

    def __init__(self, feature_dim, num_heads=8, num_classes=1, dropout=0.2):
        super().__init__()
        assert feature_dim % num_heads == 0, "feature_dim must be divisible by num_heads"
        self.feature_dim, self.num_heads, self.head_dim = feature_dim, num_heads, feature_dim // num_heads
        self.cbam = CBAM(feature_dim)
        self.query = nn.Linear(feature_dim, feature_dim)
        self.key = nn.Linear(feature_dim, feature_dim)
        self.value = nn.Linear(feature_dim, feature_dim)
        self.attention_dropout = nn.Dropout(dropout)
        self.output_projection = nn.Linear(feature_dim, feature_dim)
        self.classifier = nn.Sequential(
            nn.AdaptiveAvgPool2d(1), nn.Flatten(), nn.LayerNorm(feature_dim), nn.Dropout(dropout),
            nn.Linear(feature_dim, 512), nn.ReLU(), nn.Dropout(dropout), nn.Linear(512, num_classes))

    def forward(self, x):
        from . import head_ops
        return head_ops.mha_head_forward(self, x)


class SimpleHead(nn.Sequential):
    """AdaptiveAvgPool2d(1) -> Flatten -> Dropout -> Linear(C,512) -> ReLU -> Dropout -> Linear(512,n)
    (src/model.py:308-316); same child indices as the reference's nn.Sequential."""

    def __init__(self, feature_dim, num_classes, dropout=0.2):
        super().__init__(nn.AdaptiveAvgPool2d(1), nn.Flatten(), nn.Dropout(dropout), nn.Linear(feature_dim, 512),
                         nn.ReLU(), nn.Dropout(dropout), nn.Linear(512, num_classes))

    def forward(self, x):
        from . import head_ops
        return head_ops.simple_head_forward(self, x)

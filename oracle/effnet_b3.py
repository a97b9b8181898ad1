"""CPU oracle (TEST INFRASTRUCTURE ONLY) — plain-PyTorch fp32 restatement of the
EfficientNet-B3 features trunk and the classification heads of the reference.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this file. The shipped package never does.

What it restates, and from where:
  * trunk: ``timm.create_model('efficientnet_b3', features_only=True,
    out_indices=(0,1,2,3,4))`` as called at
    /root/reference/foundation_model/src/model.py:283-288. timm is a third-party
    dependency of the reference, NOT vendored and NOT version-pinned (no
    requirements file exists; training.log:2-3 implies timm >= 0.9). It is not
    installed here, so the trunk is restated from timm's published
    ``efficientnet_b3`` definition (channel multiplier 1.2, depth multiplier
    1.4, divisor 8, ceil depth rounding, SE ratio 0.25 of block input channels,
    SiLU, symmetric padding, BN eps 1e-5 / momentum 0.1).
    PARITY UNPINNED numerically for the trunk: the only pins are structural
    (parameter count 23,903,135 of the default model, training.log:5; dropped
    keys training.log:4; feature channel list).
  * CBAM: model.py:11-108; MultiHeadAttentionClassifier: model.py:112-175;
    simple head: model.py:308-316. These restatements ARE pinned: golden
    vectors in tests/golden/ were produced by the reference's own classes
    (tests/golden/make_golden.py).

State-dict key names follow timm (conv_stem, bn1, blocks.S.I.conv_pw ...), so a
state dict moves freely between this oracle and the HIP-backed module.
"""
import math
from typing import Dict, List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

# (type, repeats, kernel, stride, expand, out_channels) after B3 scaling.
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


class SqueezeExcite(nn.Module):
    def __init__(self, chs: int, rd: int):
        super().__init__()
        self.conv_reduce = nn.Conv2d(chs, rd, 1, bias=True)
        self.conv_expand = nn.Conv2d(rd, chs, 1, bias=True)

    def forward(self, x):
        s = x.mean((2, 3), keepdim=True)
        s = F.silu(self.conv_reduce(s))
        return x * torch.sigmoid(self.conv_expand(s))


class DSBlock(nn.Module):
    """conv_dw -> bn1+SiLU -> se -> conv_pw -> bn2 (+x)."""

    def __init__(self, cin, cout, k, s):
        super().__init__()
        self.has_skip = cin == cout and s == 1
        self.conv_dw = nn.Conv2d(cin, cin, k, s, (k - 1) // 2, groups=cin, bias=False)
        self.bn1 = nn.BatchNorm2d(cin)
        self.se = SqueezeExcite(cin, int(round(cin * 0.25)))
        self.conv_pw = nn.Conv2d(cin, cout, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)

    def forward(self, x):
        y = F.silu(self.bn1(self.conv_dw(x)))
        y = self.se(y)
        y = self.bn2(self.conv_pw(y))
        return y + x if self.has_skip else y


class IRBlock(nn.Module):
    """conv_pw -> bn1+SiLU -> conv_dw -> bn2+SiLU -> se -> conv_pwl -> bn3 (+x)."""

    def __init__(self, cin, cout, k, s, e):
        super().__init__()
        mid = cin * e
        self.has_skip = cin == cout and s == 1
        self.conv_pw = nn.Conv2d(cin, mid, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(mid)
        self.conv_dw = nn.Conv2d(mid, mid, k, s, (k - 1) // 2, groups=mid, bias=False)
        self.bn2 = nn.BatchNorm2d(mid)
        self.se = SqueezeExcite(mid, int(round(cin * 0.25)))
        self.conv_pwl = nn.Conv2d(mid, cout, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(cout)

    def forward(self, x):
        y = F.silu(self.bn1(self.conv_pw(x)))
        y = F.silu(self.bn2(self.conv_dw(y)))
        y = self.se(y)
        y = self.bn3(self.conv_pwl(y))
        return y + x if self.has_skip else y


class EfficientNetB3Features(nn.Module):
    """Returns the 5 feature maps [24,32,48,136,384] at strides 2/4/8/16/32."""

    def __init__(self):
        super().__init__()
        self.conv_stem = nn.Conv2d(3, B3_STEM, 3, 2, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(B3_STEM)
        stages = []
        cin = B3_STEM
        for (typ, reps, k, s, e, cout) in B3_STAGES:
            blocks = []
            for i in range(reps):
                stride = s if i == 0 else 1
                if typ == "ds":
                    blocks.append(DSBlock(cin, cout, k, stride))
                else:
                    blocks.append(IRBlock(cin, cout, k, stride, e))
                cin = cout
            stages.append(nn.Sequential(*blocks))
        self.blocks = nn.Sequential(*stages)
        self.reset_parameters()

    def reset_parameters(self):
        # timm efficientnet_init_weights: conv N(0, sqrt(2/fan_out)), BN 1/0, bias 0.
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                fan_out = m.kernel_size[0] * m.kernel_size[1] * m.out_channels // m.groups
                nn.init.normal_(m.weight, 0.0, math.sqrt(2.0 / fan_out))
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)

    def forward(self, x) -> List[torch.Tensor]:
        x = F.silu(self.bn1(self.conv_stem(x)))
        feats = []
        for si, stage in enumerate(self.blocks):
            x = stage(x)
            if si in B3_FEATURE_STAGES:
                feats.append(x)
        return feats


# ----------------------------------------------------------------------------
# Heads (restated from model.py:11-175, 308-316; pinned by tests/golden).
# ----------------------------------------------------------------------------
class _BasicConv(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv = nn.Conv2d(2, 1, 7, 1, 3, bias=False)
        self.bn = nn.BatchNorm2d(1, eps=1e-5, momentum=0.01, affine=True)

    def forward(self, x):
        return self.bn(self.conv(x))


class _Flatten(nn.Module):
    def forward(self, x):
        return x.reshape(x.size(0), -1)


class _ChannelGate(nn.Module):
    def __init__(self, c, r=16):
        super().__init__()
        self.mlp = nn.Sequential(_Flatten(), nn.Linear(c, c // r), nn.ReLU(), nn.Linear(c // r, c))

    def forward(self, x):
        # max_pool2d, not amax: on ties its backward routes the whole gradient to the first maximum (model.py:52)
        a = self.mlp(x.mean((2, 3))) + self.mlp(F.max_pool2d(x, (x.size(2), x.size(3))))
        return x * torch.sigmoid(a)[:, :, None, None]


class _SpatialGate(nn.Module):
    def __init__(self):
        super().__init__()
        self.spatial = _BasicConv()

    def forward(self, x):
        # torch.max(x, 1)[0] as at model.py:79: ties send the gradient to the first maximal channel
        comp = torch.cat((x.max(1, keepdim=True)[0], x.mean(1, keepdim=True)), dim=1)
        return x * torch.sigmoid(self.spatial(comp))


class CBAM(nn.Module):
    def __init__(self, c, r=16):
        super().__init__()
        self.ChannelGate = _ChannelGate(c, r)
        self.SpatialGate = _SpatialGate()

    def forward(self, x):
        return self.SpatialGate(self.ChannelGate(x))


class MultiHeadAttentionClassifier(nn.Module):
    def __init__(self, feature_dim, num_heads=8, num_classes=1, dropout=0.2):
        super().__init__()
        assert feature_dim % num_heads == 0
        self.num_heads, self.head_dim = num_heads, feature_dim // num_heads
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
        x = self.cbam(x)
        B, C, H, W = x.shape
        t = x.reshape(B, C, H * W).permute(0, 2, 1)
        sp = lambda y: y.reshape(B, H * W, self.num_heads, self.head_dim).transpose(1, 2)
        q, k, v = sp(self.query(t)), sp(self.key(t)), sp(self.value(t))
        w = F.softmax(q @ k.transpose(-2, -1) / math.sqrt(self.head_dim), dim=-1)
        w = self.attention_dropout(w)
        a = (w @ v).transpose(1, 2).reshape(B, H * W, C)
        a = self.output_projection(a).permute(0, 2, 1).reshape(B, C, H, W)
        return self.classifier(x + a)


def simple_head(feature_dim, num_classes, dropout=0.2):
    return nn.Sequential(nn.AdaptiveAvgPool2d(1), nn.Flatten(), nn.Dropout(dropout),
                         nn.Linear(feature_dim, 512), nn.ReLU(), nn.Dropout(dropout),
                         nn.Linear(512, num_classes))


class OracleFoundationModel(nn.Module):
    """FoundationModel (model.py:240-363): trunk + classification heads + segmentation decoders."""

    def __init__(self, classification_heads: Optional[Dict[str, int]] = None, dropout=0.2,
                 use_attention=True, num_attention_heads=8, segmentation_heads: Optional[Dict[str, int]] = None):
        super().__init__()
        self.backbone = EfficientNetB3Features()
        self.feature_dims = list(B3_FEATURE_DIMS)
        self.classification_layers = nn.ModuleDict()
        for name, n in (classification_heads or {}).items():
            if use_attention:
                self.classification_layers[name] = MultiHeadAttentionClassifier(
                    self.feature_dims[-1], num_attention_heads, n, dropout)
            else:
                self.classification_layers[name] = simple_head(self.feature_dims[-1], n, dropout)
        self.segmentation_layers = nn.ModuleDict()
        if segmentation_heads:
            from . import seg_ref
            cls = seg_ref.EnhancedUNetDecoder if use_attention else seg_ref.UNetDecoder
            for name, n in segmentation_heads.items():
                self.segmentation_layers[name] = cls(self.feature_dims, num_classes=n, dropout=dropout)

    def forward(self, x, task, dataset_name):
        feats = self.backbone(x)
        if task == "segmentation":
            if dataset_name not in self.segmentation_layers:
                raise ValueError(f"Segmentation head for {dataset_name} not found")
            return self.segmentation_layers[dataset_name](feats, x.shape[-2:])
        if task != "classification":
            raise ValueError(f"Unknown task: {task}")
        if dataset_name not in self.classification_layers:
            raise ValueError(f"Classification head for {dataset_name} not found")
        return self.classification_layers[dataset_name](feats[-1])

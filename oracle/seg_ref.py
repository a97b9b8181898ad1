"""TEST INFRASTRUCTURE ONLY — CPU restatement (plain PyTorch fp32) of the reference's segmentation path:
decoder blocks, the two UNet-style decoders and the Dice / IoU / combined losses. Only tests/, smoke() and the
CPU-baseline leg of bench.py may import this file; the product never does.

Pinned by golden vectors made from the reference's own classes (tests/golden/dec_*.npz, decblock_*.npz,
segloss_*.npz; generator tests/golden/make_golden.py), see tests/test_oracle.py.

Follows /root/reference/foundation_model/src/model.py:179-237 (EnhancedUNetDecoder), :417-468 (UNetDecoder),
:471-494 (DecoderBlock) and /root/reference/foundation_model/src/loss.py:27-80, :105-136.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .effnet_b3 import CBAM

DECODER_CHANNELS = (256, 128, 64, 32, 16)


class DecoderBlock(nn.Module):
    """(3x3 conv + bias -> BatchNorm -> ReLU) twice, channel dropout between them (model.py:471-494)."""

    def __init__(self, cin, cout, dropout=0.2):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.bn1 = nn.BatchNorm2d(cout)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.bn2 = nn.BatchNorm2d(cout)
        self.dropout = nn.Dropout2d(dropout)
        self.relu = nn.ReLU()

    def forward(self, x):
        x = self.dropout(F.relu(self.bn1(self.conv1(x))))
        return F.relu(self.bn2(self.conv2(x)))


class UNetDecoder(nn.Module):
    """Deepest feature first; every further level: bilinear upsample (align_corners=False) to the skip's size,
    concat [upsampled, skip], DecoderBlock; then upsample to the input size and a 1x1 conv (model.py:417-468).
    With `attention` each level is followed by CBAM (model.py:179-237)."""

    def __init__(self, encoder_channels, decoder_channels=DECODER_CHANNELS, num_classes=1, dropout=0.2,
                 attention=False):
        super().__init__()
        enc = list(encoder_channels)[::-1]
        self.decoder_blocks = nn.ModuleList()
        for i, (e, d) in enumerate(zip(enc, decoder_channels)):
            self.decoder_blocks.append(DecoderBlock(e + (decoder_channels[i - 1] if i else 0), d, dropout))
        if attention:
            self.cbam_modules = nn.ModuleList(CBAM(d) for d in decoder_channels[:len(enc)])
        self.final_conv = nn.Conv2d(decoder_channels[-1], num_classes, 1)

    def forward(self, features, input_size):
        feats = list(features)[::-1]
        x = None
        for i, blk in enumerate(self.decoder_blocks):
            if i == 0:
                x = blk(feats[0])
            else:
                x = F.interpolate(x, size=feats[i].shape[-2:], mode="bilinear", align_corners=False)
                x = blk(torch.cat((x, feats[i]), dim=1))
            if hasattr(self, "cbam_modules"):
                x = self.cbam_modules[i](x)
        x = F.interpolate(x, size=tuple(input_size), mode="bilinear", align_corners=False)
        return self.final_conv(x)


class EnhancedUNetDecoder(UNetDecoder):
    def __init__(self, encoder_channels, decoder_channels=DECODER_CHANNELS, num_classes=1, dropout=0.2):
        super().__init__(encoder_channels, decoder_channels, num_classes, dropout, attention=True)


def _overlap_score(logits, target, num_classes, smooth, iou):
    """mean over (image, class) of Dice = (2I+s)/(P+T+s) or IoU = (I+s)/(P+T-I+s) on softmax probabilities."""
    p = F.softmax(logits, dim=1)
    t = F.one_hot(target, num_classes).permute(0, 3, 1, 2).to(p.dtype)
    inter = (p * t).sum((2, 3))
    tot = p.sum((2, 3)) + t.sum((2, 3))
    score = (inter + smooth) / (tot - inter + smooth) if iou else (2.0 * inter + smooth) / (tot + smooth)
    return score.mean()


def dice_loss(logits, target, num_classes, smooth=1e-5):
    return 1.0 - _overlap_score(logits, target, num_classes, smooth, iou=False)


def iou_loss(logits, target, num_classes, smooth=1e-5):
    return 1.0 - _overlap_score(logits, target, num_classes, smooth, iou=True)


def combined_segmentation_loss(logits, target, num_classes, weights=None, dice_weight=0.5, ce_weight=0.5):
    return ce_weight * F.cross_entropy(logits, target, weight=weights) + dice_weight * dice_loss(logits, target,
                                                                                                 num_classes)

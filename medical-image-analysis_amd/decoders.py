"""Segmentation decoders of the reference on the gfx950 kernels, NHWC in, NCHW fp32 logits out.

State-dict keys equal the reference's (src/model.py:179-237, :417-494):
  decoder_blocks.{i}.{conv1,conv2}.{weight,bias}, decoder_blocks.{i}.{bn1,bn2}.*,
  cbam_modules.{i}.ChannelGate.mlp.{1,3}.*, cbam_modules.{i}.SpatialGate.spatial.{conv.weight,bn.*} (enhanced only),
  final_conv.{weight,bias}.
Parameter containers are ordinary nn.Conv2d / nn.BatchNorm2d objects (same shapes and default initialisation as
the reference's); their forward() is never called.
"""
import torch.nn as nn

from .heads import CBAM

DECODER_CHANNELS = (256, 128, 64, 32, 16)


class DecoderBlock(nn.Module):
    """conv3x3+bias -> BN -> ReLU -> Dropout2d -> conv3x3+bias -> BN -> ReLU (src/model.py:471-494); x: NHWC."""

    def __init__(self, in_channels, out_channels, dropout=0.2):
        super().__init__()
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, padding=1)
        self.bn1 = nn.BatchNorm2d(out_channels)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, padding=1)
        self.bn2 = nn.BatchNorm2d(out_channels)
        self.dropout = nn.Dropout2d(dropout)
        self.relu = nn.ReLU(inplace=True)

    def forward(self, x):
        from . import seg_ops
        return seg_ops.decoder_block_forward(self, x)


class UNetDecoder(nn.Module):
    """src/model.py:417-468. forward(features, input_size): features = the trunk's five NHWC maps, shallowest first."""

    def __init__(self, encoder_channels, decoder_channels=DECODER_CHANNELS, num_classes=1, dropout=0.2):
        super().__init__()
        enc = list(encoder_channels)[::-1]
        self.decoder_blocks = nn.ModuleList()
        for i, (e, d) in enumerate(zip(enc, decoder_channels)):
            self.decoder_blocks.append(DecoderBlock(e + (decoder_channels[i - 1] if i else 0), d, dropout=dropout))
        self._make_attention(decoder_channels[:len(enc)])
        self.final_conv = nn.Conv2d(decoder_channels[-1], num_classes, kernel_size=1)

    def _make_attention(self, channels):
        pass

    def forward(self, features, input_size):
        from . import seg_ops
        return seg_ops.decoder_forward(self, features, input_size)


class EnhancedUNetDecoder(UNetDecoder):
    """src/model.py:179-237: the same decoder with a CBAM after every block."""

    def _make_attention(self, channels):
        self.cbam_modules = nn.ModuleList(CBAM(c) for c in channels)

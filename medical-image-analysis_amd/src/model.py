"""Drop-in for the reference's `src.model` import surface
(/root/reference/foundation_model/src/model.py): `create_foundation_model`, `FoundationModel`,
`load_model_weights_safely` with the same signatures, attributes, state-dict keys and error
behaviour, so train.py / finetune.py / inference.py style drivers
(`from src.model import create_foundation_model, load_model_weights_safely`, train.py:18) run on
the MI355X kernels when this package directory is on sys.path in place of foundation_model/.

Differences that a caller can observe:
  * the trunk is this package's EfficientNetB3Features (timm is not needed); `pretrained=True`
    cannot download weights (no network) and raises unless MIA_ALLOW_RANDOM_INIT=1 is set;
  * segmentation heads are the reference's two decoders on this package's kernels (mia_amd.decoders);
  * new optional `compute_dtype` ('fp32' | 'bf16'; config key runtime.dtype), default fp32 as in the reference.
"""
import logging
import os
import sys
from typing import Dict, Optional

import torch
import torch.nn as nn

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if "mia_amd" not in sys.modules:
    import importlib.util
    _spec = importlib.util.spec_from_file_location("mia_amd", os.path.join(_PKG, "__init__.py"),
                                                   submodule_search_locations=[_PKG])
    _mod = importlib.util.module_from_spec(_spec)
    sys.modules["mia_amd"] = _mod
    _spec.loader.exec_module(_mod)

from mia_amd.decoders import EnhancedUNetDecoder, UNetDecoder  # noqa: E402
from mia_amd.heads import MultiHeadAttentionClassifier, SimpleHead  # noqa: E402
from mia_amd.trunk import EfficientNetB3Features  # noqa: E402

_DTYPES = {"fp32": torch.float32, "float32": torch.float32, "bf16": torch.bfloat16, "bfloat16": torch.bfloat16,
           None: torch.float32}
_BACKBONES = ("efficientnet_b3", "efficientnet-b3")


class FoundationModel(nn.Module):
    """Same constructor and forward contract as the reference's FoundationModel (model.py:240-363)."""

    def __init__(self, backbone: str = "efficientnet-b3", pretrained: bool = True, dropout: float = 0.2,
                 classification_heads: Optional[Dict[str, int]] = None,
                 segmentation_heads: Optional[Dict[str, int]] = None, use_attention: bool = True,
                 num_attention_heads: int = 8, compute_dtype="fp32"):
        super().__init__()
        if backbone not in _BACKBONES:
            raise ValueError(f"Unsupported backbone {backbone!r}: this build implements {_BACKBONES[0]} only")
        if pretrained and os.environ.get("MIA_ALLOW_RANDOM_INIT") != "1":
            raise RuntimeError("pretrained=True needs the timm/efficientnet_b3.ra2_in1k weights, which cannot be "
                               "downloaded here; load a checkpoint with load_model_weights_safely() and pass "
                               "pretrained=False (or set MIA_ALLOW_RANDOM_INIT=1)")
        self.backbone_name = backbone
        self.classification_heads = classification_heads or {}
        self.segmentation_heads = segmentation_heads or {}
        self.use_attention = use_attention
        self.num_attention_heads = num_attention_heads
        dt = compute_dtype if isinstance(compute_dtype, torch.dtype) else _DTYPES[compute_dtype]
        self.backbone = EfficientNetB3Features(compute_dtype=dt)
        self.feature_dims = list(self.backbone.feature_dims)
        self.classification_layers = nn.ModuleDict()
        for name, n in self.classification_heads.items():
            self.classification_layers[name] = self._make_cls_head(n, dropout)
        self.segmentation_layers = nn.ModuleDict()
        for name, n in self.segmentation_heads.items():
            self.segmentation_layers[name] = self._make_seg_head(n, dropout)

    def _make_cls_head(self, num_classes, dropout):
        if self.use_attention:
            return MultiHeadAttentionClassifier(self.feature_dims[-1], self.num_attention_heads, num_classes, dropout)
        return SimpleHead(self.feature_dims[-1], num_classes, dropout)

    def _make_seg_head(self, num_classes, dropout):
        cls = EnhancedUNetDecoder if self.use_attention else UNetDecoder
        return cls(encoder_channels=self.feature_dims, decoder_channels=(256, 128, 64, 32, 16),
                   num_classes=num_classes, dropout=dropout)

    def set_compute_dtype(self, dtype):
        self.backbone.compute_dtype = dtype if isinstance(dtype, torch.dtype) else _DTYPES[dtype]
        return self

    def refresh_inference_weights(self):
        """Eval-mode inference keeps BatchNorm-folded copies of the trunk's 1x1 convolution weights; they refresh
        themselves on every eager forward, and a HIP-graph replay (graphs.GraphedForward) calls this first."""
        self.backbone.refresh_inference_weights()

    def _check_device(self, x):
        p = next(self.parameters())
        if not x.is_cuda or not p.is_cuda:
            raise RuntimeError("mia_amd kernels need the model and its input on an MI355X device; there is no CPU "
                               f"path (input on {x.device}, parameters on {p.device})")

    def forward(self, x: torch.Tensor, task: str, dataset_name: str):
        if task == "classification":
            if dataset_name not in self.classification_layers:
                raise ValueError(f"Classification head for {dataset_name} not found")
            self._check_device(x)
            feats = self.backbone.forward_nhwc(x)
            return self.classification_layers[dataset_name](feats[-1])
        elif task == "segmentation":
            if dataset_name not in self.segmentation_layers:
                raise ValueError(f"Segmentation head for {dataset_name} not found")
            self._check_device(x)
            feats = self.backbone.forward_nhwc(x)
            return self.segmentation_layers[dataset_name](feats, x.shape[-2:])
        else:
            raise ValueError(f"Unknown task: {task}")

    def add_classification_head(self, dataset_name: str, num_classes: int, dropout: float = 0.2):
        """model.py:365-390: new head on the same device as the model."""
        self.classification_heads[dataset_name] = num_classes
        head = self._make_cls_head(num_classes, dropout)
        device = next(self.parameters()).device
        self.classification_layers[dataset_name] = head.to(device)

    def add_segmentation_head(self, dataset_name: str, num_classes: int, dropout: float = 0.2):
        """model.py:392-414: new decoder on the same device as the model."""
        self.segmentation_heads[dataset_name] = num_classes
        head = self._make_seg_head(num_classes, dropout)
        device = next(self.parameters()).device
        self.segmentation_layers[dataset_name] = head.to(device)


def create_foundation_model(config: dict = None, backbone: str = None, pretrained: bool = None, dropout: float = None,
                            classification_heads: dict = None, segmentation_heads: dict = None,
                            use_attention: bool = None, num_attention_heads: int = None,
                            compute_dtype=None) -> FoundationModel:
    """model.py:497-546. The config form reads config['model'][...] and the two dataset lists."""
    if config is not None:
        classification_heads = {}
        for dc in config.get("classification_datasets", []):
            classification_heads[dc["name"]] = len(dc["classes"])
        segmentation_heads = {}
        for dc in config.get("segmentation_datasets", []):
            segmentation_heads[dc["name"]] = dc["num_classes"]
        if compute_dtype is None:
            compute_dtype = (config.get("runtime") or {}).get("dtype")
        return FoundationModel(backbone=config["model"]["backbone"], pretrained=config["model"]["pretrained"],
                               dropout=config["model"]["dropout"], classification_heads=classification_heads,
                               segmentation_heads=segmentation_heads,
                               use_attention=config["model"].get("use_attention", True),
                               num_attention_heads=config["model"].get("num_attention_heads", 8),
                               compute_dtype=compute_dtype)
    return FoundationModel(backbone=backbone or "efficientnet-b3",
                           pretrained=pretrained if pretrained is not None else True, dropout=dropout or 0.2,
                           classification_heads=classification_heads or {},
                           segmentation_heads=segmentation_heads or {},
                           use_attention=use_attention if use_attention is not None else True,
                           num_attention_heads=num_attention_heads or 8, compute_dtype=compute_dtype)


def load_model_weights_safely(model, checkpoint, logger=None, suppress_warnings=False):
    """model.py:549-604: load_state_dict(strict=False) on a wrapped or bare state dict; returns
    (missing_keys, unexpected_keys); the head/bn2/conv_head keys of a full timm checkpoint are expected."""
    if logger is None:
        logger = logging.getLogger(__name__)
    if isinstance(checkpoint, dict) and "model_state_dict" in checkpoint:
        state_dict = checkpoint["model_state_dict"]
    else:
        state_dict = checkpoint
    missing_keys, unexpected_keys = model.load_state_dict(state_dict, strict=False)
    expected = ["classifier.weight", "classifier.bias", "conv_head.weight", "conv_head.bias", "bn2.weight", "bn2.bias",
                "bn2.running_mean", "bn2.running_var", "bn2.num_batches_tracked"]
    truly_unexpected = [k for k in unexpected_keys if not any(e in k for e in expected)]
    if not suppress_warnings:
        if missing_keys:
            logger.warning(f"Missing keys when loading model (these will be randomly initialized): {missing_keys}")
        if truly_unexpected:
            logger.warning(f"Truly unexpected keys found in checkpoint: {truly_unexpected}")
        if unexpected_keys and not truly_unexpected:
            logger.info("Found expected pretrained model keys that were replaced with task-specific layers "
                        "(this is normal)")
    else:
        if missing_keys:
            logger.debug(f"Missing keys when loading model: {missing_keys}")
        if truly_unexpected:
            logger.warning(f"Truly unexpected keys found in checkpoint: {truly_unexpected}")
    return missing_keys, unexpected_keys

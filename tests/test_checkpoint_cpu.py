"""Checkpoint compatibility (no GPU): files in the reference's two layouts (train.py:434-446 `config` form,
finetune.py:481-494 `model_config` form) written from this package's model load into the CPU oracle - whose module
tree restates the reference's - with strict key matching, and the other way round through
load_model_weights_safely; the inference-side rebuild from `model_config` (inference.py:84-108) reproduces the heads."""
import torch

import mia_amd  # noqa: F401
from mia_amd.src.model import FoundationModel, create_foundation_model, load_model_weights_safely
from oracle.effnet_b3 import OracleFoundationModel
from seeding import seeded_state_dict

CLS, SEG = {"brain": 4, "pancreas": 2}, {"btcv": 2}


def _mine(use_attention=True):
    return FoundationModel(backbone="efficientnet_b3", pretrained=False, dropout=0.2, classification_heads=dict(CLS),
                           segmentation_heads=dict(SEG), use_attention=use_attention)


def _oracle(use_attention=True):
    return OracleFoundationModel(dict(CLS), dropout=0.2, use_attention=use_attention, segmentation_heads=dict(SEG))


def test_state_dict_keys_and_shapes_equal_the_oracle_module_tree():
    for att in (True, False):
        a, b = _mine(att).state_dict(), _oracle(att).state_dict()
        assert list(a.keys()) == list(b.keys())
        assert all(a[k].shape == b[k].shape and a[k].dtype == b[k].dtype for k in a)


def test_reference_layout_checkpoints_round_trip(tmp_path):
    mine = _mine()
    mine.load_state_dict(seeded_state_dict(mine, 17))
    config = {"model": {"backbone": "efficientnet_b3", "pretrained": False, "dropout": 0.2, "use_attention": True},
              "classification_datasets": [{"name": k, "classes": list(range(v))} for k, v in CLS.items()],
              "segmentation_datasets": [{"name": k, "num_classes": v} for k, v in SEG.items()]}
    train_ckpt = {"epoch": 3, "model_state_dict": mine.state_dict(), "task": "classification", "dataset_name": "brain",
                  "val_loss": 0.5, "config": config}
    finetune_ckpt = {"model_state_dict": mine.state_dict(),
                     "model_config": {"backbone": mine.backbone_name, "classification_heads": mine.classification_heads,
                                      "segmentation_heads": mine.segmentation_heads, "dropout": 0.2}}
    for name, ck in (("train.pth", train_ckpt), ("finetune.pth", finetune_ckpt)):
        path = tmp_path / name
        torch.save(ck, path)
        loaded = torch.load(path, map_location="cpu", weights_only=True)
        # into the oracle (the reference's module tree): every key must be used, none may be missing
        ref = _oracle()
        res = ref.load_state_dict(loaded["model_state_dict"], strict=True)
        assert not res.missing_keys and not res.unexpected_keys
        # back into a fresh model of this package, built the way the drivers rebuild it
        if "config" in loaded:
            fresh = create_foundation_model(config=loaded["config"])
        else:
            mc = loaded["model_config"]
            fresh = create_foundation_model(backbone=mc["backbone"], pretrained=False, dropout=mc["dropout"],
                                            classification_heads=mc["classification_heads"],
                                            segmentation_heads=mc["segmentation_heads"])
        missing, unexpected = load_model_weights_safely(fresh, loaded, suppress_warnings=True)
        assert not missing and not unexpected
        assert fresh.classification_heads == CLS and fresh.segmentation_heads == SEG
        for k, v in fresh.state_dict().items():
            assert torch.equal(v, mine.state_dict()[k]), k


def test_oracle_checkpoint_loads_into_this_package():
    ref = _oracle(False)
    sd = seeded_state_dict(ref, 23)
    ref.load_state_dict(sd)
    mine = _mine(False)
    missing, unexpected = load_model_weights_safely(mine, {"model_state_dict": ref.state_dict()}, suppress_warnings=True)
    assert not missing and not unexpected
    # a checkpoint from a model with more heads: extra keys are reported, shared ones load
    small = FoundationModel(backbone="efficientnet_b3", pretrained=False, classification_heads={"brain": 4},
                            use_attention=False)
    missing, unexpected = load_model_weights_safely(small, ref.state_dict(), suppress_warnings=True)
    assert not missing and all(k.startswith(("classification_layers.pancreas", "segmentation_layers.btcv")) for k in unexpected)

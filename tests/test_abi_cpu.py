"""CPU-side checks (no GPU): the C-ABI library builds/loads and exports every symbol include/mia_hip.h declares,
the model surface mirrors the reference's (names, attributes, state-dict keys, errors, config handling), and the
product path refuses to run without a device instead of falling back."""
import ctypes
import os
import re
import subprocess

import pytest
import torch
import yaml

import mia_amd
from mia_amd import _lib
from mia_amd.src.model import FoundationModel, create_foundation_model, load_model_weights_safely

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_every_declared_symbol():
    protos = _lib.parse_header()
    assert len(protos) >= 30
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH], text=True)
    exported = set(re.findall(r" T (mia_\w+)", out))
    missing = sorted(set(protos) - exported)
    assert not missing, f"declared in include/mia_hip.h but not exported: {missing}"
    undeclared = sorted(exported - set(protos))
    assert not undeclared, f"exported but not declared in the header: {undeclared}"
    lib = mia_amd.lib()
    assert lib.mia_version() >= 100
    assert lib.mia_last_error() is not None


def test_fast_call_binding_matches_the_ctypes_binding():
    """The generated fast-call module is bound to the same loaded library and behaves like the ctypes prototypes:
    same return values, same exception type and message for a refused call, argument count checked."""
    lib = mia_amd.lib()
    if not lib.fastcall:       # a tree where only the kernels were built: generate and bind it now (gcc, a second)
        import importlib.util
        spec = importlib.util.spec_from_file_location("mia_build", os.path.join(ROOT, "medical-image-analysis_amd", "build.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        mod.build_fast()
        lib._bind_fast()
    assert lib.fastcall, "build.py's build_fast() output is missing or was generated from other prototypes"
    from mia_amd import _mia_fast, _proto
    assert _mia_fast.digest() == _proto.header_digest()
    assert lib.mia_version is _mia_fast.mia_version
    assert lib.mia_version() == lib._dll.mia_version()
    assert lib.mia_dwconv_bwd_data_lin_supported(1, 144, 3, 2, 24) == 1
    assert lib.mia_dwconv_bwd_data_lin_supported(0, 144, 3, 2, 24) == 0
    assert lib.mia_gemm_tn_ws_floats(1, 100000, 144, 24, 0) == lib._dll.mia_gemm_tn_ws_floats(1, 100000, 144, 24, 0)
    with pytest.raises(mia_amd.MiaError, match=r"mia_gemm_nt failed \(-1\): .*bad dtype"):
        lib.mia_gemm_nt(7, None, 0, None, 0, None, 0, None, 1, 1, 1, None, None, 0, None, 0, None, 0, None, 0, None)
    with pytest.raises(TypeError, match="takes 21 arguments"):
        lib.mia_gemm_nt(7, None)
    with pytest.raises(TypeError):
        lib.mia_gemm_nt("x", None, 0, None, 0, None, 0, None, 1, 1, 1, None, None, 0, None, 0, None, 0, None, 0, None)
    # addresses: ints, None, ctypes pointer objects and ctypes arrays
    arr = (ctypes.c_float * 3)(1.0, 2.0, 3.0)
    with pytest.raises(mia_amd.MiaError, match="channels must be 1 or 3"):
        lib.mia_resize_normalize_u8(16, 1, 4, 4, 2, ctypes.c_void_p(16), 4, 4, arr, ctypes.cast(arr, ctypes.c_void_p), 1, None)


def test_header_cites_reference_for_every_entry_point_group():
    text = open(os.path.join(ROOT, "include", "mia_hip.h")).read()
    for cite in ("src/model.py:283-288", "src/model.py:28-108", "model.py:153-163", "src/loss.py:147",
                 "src/dataset.py:247-251", "zi_remover.py:3-19"):
        assert cite in text, cite


def test_argument_validation_needs_no_gpu():
    lib = mia_amd.lib()
    with pytest.raises(mia_amd.MiaError, match="bad dtype"):
        lib.mia_gemm_nt(7, None, 0, None, 0, None, 0, None, 1, 1, 1, None, None, 0, None, 0, None, 0, None, 0, None)
    with pytest.raises(mia_amd.MiaError, match="null"):
        lib.mia_zero_slice_filter(None, 2, 0, 4, 16, None, None, None, None, None)
    with pytest.raises(mia_amd.MiaError, match="channels must be 1 or 3"):
        lib.mia_resize_normalize_u8(ctypes.c_void_p(16), 1, 4, 4, 2, ctypes.c_void_p(16), 4, 4, None, None, 1, None)


def test_no_cpu_fallback():
    from mia_amd import ops
    a = torch.zeros(8, 8)
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.gemm_nt(a, a)
    from mia_amd import preprocess
    with pytest.raises(RuntimeError):
        preprocess.resize_normalize(torch.zeros(1, 4, 4, 3, dtype=torch.uint8), (8, 8))
    with pytest.raises(RuntimeError):
        preprocess.zero_slice_filter(torch.zeros(2, 4, 4))


def test_factory_reads_reference_config_yaml(tmp_path):
    cfg = yaml.safe_load(open("/root/reference/foundation_model/config.yaml")) if os.path.exists(
        "/root/reference/foundation_model/config.yaml") else None
    if cfg is None:   # GPU box: no reference tree; same keys restated
        cfg = {"model": {"backbone": "efficientnet_b3", "pretrained": True, "dropout": 0.2, "use_attention": True,
                         "num_attention_heads": 8},
               "classification_datasets": [{"name": "brain_mri_nd5", "classes": ["a", "b", "c", "d"]},
                                           {"name": "hf_brain_tumor", "classes": ["a", "b", "c", "d"]},
                                           {"name": "pancreas", "classes": ["n", "p"]}],
               "segmentation_datasets": [{"name": "btcv", "num_classes": 2}]}
    with pytest.raises(RuntimeError, match="pretrained=True"):
        create_foundation_model(config=cfg)
    cfg["model"]["pretrained"] = False
    m = create_foundation_model(config=cfg)
    assert m.classification_heads == {"brain_mri_nd5": 4, "hf_brain_tumor": 4, "pancreas": 2}
    assert m.backbone_name == "efficientnet_b3" and m.use_attention and m.num_attention_heads == 8
    assert m.feature_dims == [24, 32, 48, 136, 384]
    # trunk + 3 attention heads + one CBAM-UNet decoder (2,274,021 parameters at 2 classes) per segmentation
    # dataset; with the reference config's 5 of them this is the logged 23,903,135 (training.log:5)
    n = sum(p.numel() for p in m.parameters())
    n_seg = len(cfg.get("segmentation_datasets", []))
    assert set(m.segmentation_layers.keys()) == {d["name"] for d in cfg.get("segmentation_datasets", [])}
    assert n == 10_103_336 + 2 * 810_240 + 809_214 + sum(
        2_274_021 + 17 * (d["num_classes"] - 2) for d in cfg.get("segmentation_datasets", []))
    if n_seg == 5 and all(d["num_classes"] == 2 for d in cfg["segmentation_datasets"]):
        assert n == 23_903_135
    cfg["runtime"] = {"dtype": "bf16"}
    assert create_foundation_model(config=cfg).backbone.compute_dtype == torch.bfloat16


def test_keyword_form_defaults_and_quirks():
    m = create_foundation_model(backbone="efficientnet_b3", pretrained=False, classification_heads={"x": 3},
                                use_attention=False, dropout=0.0)
    assert m.classification_layers["x"][2].p == 0.2          # `dropout or 0.2` (reference model.py:539)
    assert not m.use_attention and m.num_attention_heads == 8
    assert [type(c).__name__ for c in m.classification_layers["x"]] == [
        "AdaptiveAvgPool2d", "Flatten", "Dropout", "Linear", "ReLU", "Dropout", "Linear"]
    with pytest.raises(ValueError):
        FoundationModel(backbone="resnet50", pretrained=False)
    for name in ("efficientnet-b3", "efficientnet_b3"):
        FoundationModel(backbone=name, pretrained=False)


def test_state_dict_keys_match_reference_naming():
    m = FoundationModel(backbone="efficientnet_b3", pretrained=False, classification_heads={"ds": 4})
    keys = set(m.state_dict().keys())
    for k in ("backbone.conv_stem.weight", "backbone.bn1.running_mean", "backbone.blocks.0.0.conv_dw.weight",
              "backbone.blocks.0.0.se.conv_reduce.bias", "backbone.blocks.1.0.conv_pwl.weight",
              "backbone.blocks.6.1.bn3.num_batches_tracked",
              "classification_layers.ds.cbam.ChannelGate.mlp.1.weight", "classification_layers.ds.cbam.ChannelGate.mlp.3.bias",
              "classification_layers.ds.cbam.SpatialGate.spatial.conv.weight",
              "classification_layers.ds.cbam.SpatialGate.spatial.bn.running_var",
              "classification_layers.ds.query.weight", "classification_layers.ds.output_projection.bias",
              "classification_layers.ds.classifier.2.weight", "classification_layers.ds.classifier.4.weight",
              "classification_layers.ds.classifier.7.bias"):
        assert k in keys, k
    golden = open(os.path.join(ROOT, "tests", "golden", "param_counts.txt")).read()
    assert "MHA(384,8,4)=810240" in golden
    assert sum(p.numel() for p in m.classification_layers["ds"].parameters()) == 810240
    assert sum(p.numel() for p in m.classification_layers["ds"].cbam.parameters()) == 18940


def test_load_model_weights_safely_matches_reference_key_handling():
    # golden: what the reference's loader returned for the same synthetic state dict (tests/golden/make_golden.py)
    lines = dict(l.strip().split("=", 1) for l in open(os.path.join(ROOT, "tests", "golden", "load_weights_keys.txt")))

    class Toy(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.a = torch.nn.Linear(2, 2)
            self.b = torch.nn.Linear(2, 2)
    sd = {"a.weight": torch.ones(2, 2), "a.bias": torch.zeros(2), "classifier.weight": torch.ones(1),
          "bn2.running_mean": torch.ones(1), "conv_head.weight": torch.ones(1), "zzz.weight": torch.ones(1)}
    for ck in (sd, {"model_state_dict": sd}):
        missing, unexpected = load_model_weights_safely(Toy(), ck, suppress_warnings=True)
        assert ",".join(missing) == lines["missing"]
        assert ",".join(unexpected) == lines["unexpected"]


def test_forward_errors_before_touching_the_device():
    m = FoundationModel(backbone="efficientnet_b3", pretrained=False, classification_heads={"a": 2})
    x = torch.zeros(1, 3, 32, 32)
    with pytest.raises(ValueError, match="Classification head for nope not found"):
        m(x, "classification", "nope")
    with pytest.raises(ValueError, match="Unknown task"):
        m(x, "detect", "a")
    with pytest.raises(ValueError, match="Segmentation head for a not found"):
        m(x, "segmentation", "a")
    seg = FoundationModel(backbone="efficientnet_b3", pretrained=False, segmentation_heads={"s": 2})
    assert type(seg.segmentation_layers["s"]).__name__ == "EnhancedUNetDecoder"
    keys = list(seg.segmentation_layers["s"].state_dict().keys())
    assert keys[0] == "decoder_blocks.0.conv1.weight" and keys[-1] == "final_conv.bias"
    assert "cbam_modules.4.SpatialGate.spatial.bn.running_var" in keys
    plain = FoundationModel(backbone="efficientnet_b3", pretrained=False, segmentation_heads={"s": 2}, use_attention=False)
    assert type(plain.segmentation_layers["s"]).__name__ == "UNetDecoder"
    # host tensors are refused before anything is launched (a host address inside a kernel would fault the GPU)
    with pytest.raises(RuntimeError, match="no CPU path"):
        seg(x, "segmentation", "s")
    with pytest.raises(RuntimeError, match="no CPU path"):
        m(x, "classification", "a")
    from mia_amd import ops
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops._p(torch.zeros(4))
    assert ops._p(None) is None

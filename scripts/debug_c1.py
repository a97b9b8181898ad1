import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests", "golden"))
import torch, numpy as np
import mia_amd
from mia_amd.src.model import create_foundation_model, load_model_weights_safely
from oracle import effnet_b3 as O
from seeding import randn, seeded_state_dict

def relerr(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
R = int(sys.argv[2]) if len(sys.argv) > 2 else 300
heads = {"synthetic": 4}
ref = O.OracleFoundationModel(heads, dropout=0.0, use_attention=True)
sd = seeded_state_dict(ref, 3)
ref.load_state_dict(sd)
mine = create_foundation_model(backbone="efficientnet_b3", pretrained=False, dropout=0.0, classification_heads=heads, use_attention=True)
load_model_weights_safely(mine, sd)
mine.cuda()
x = randn(100, B, 3, R, R)
ref.train(); mine.train()
# stage by stage
with torch.no_grad():
    import torch.nn.functional as F
    hr = F.silu(ref.backbone.bn1(ref.backbone.conv_stem(x)))
    from mia_amd import ops
    from mia_amd.trunk import _Cfg
    cfg = _Cfg(); cfg.dtype = torch.float32; cfg.bn = ops.BNState(mine.backbone.bn1)
    hm = ops.StemFn.apply(x.cuda(), mine.backbone.conv_stem.weight, mine.backbone.bn1.weight, mine.backbone.bn1.bias, cfg)
    print("stem", relerr(hm.permute(0,3,1,2), hr), tuple(hr.shape))
    for si, (sr, sm) in enumerate(zip(ref.backbone.blocks, mine.backbone.blocks)):
        for bi, (br, bm) in enumerate(zip(sr, sm)):
            hr = br(hr)
            hm_self = bm(hm)                                  # mine fed by my own previous output
            hm = hm_self
            print(f"block {si}.{bi} chained {relerr(hm.permute(0,3,1,2), hr):.3e}  shape {tuple(hr.shape)} absmax {hr.abs().max().item():.3f}")
    fr = hr
    lr = ref.classification_layers["synthetic"](fr)
    lm = mine.classification_layers["synthetic"](hm)
    lm2 = mine.classification_layers["synthetic"](fr.permute(0,2,3,1).contiguous().cuda())
    print("logits chained", relerr(lm, lr), "head on oracle features", relerr(lm2, lr))
    print(lr[:2], lm[:2])

"""Where does a replayed training step leave the eager one: logits, loss, gradients, updated weights."""
import copy, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import mia_amd  # noqa: F401
from mia_amd.graphs import GraphedTrainStep
from mia_amd.head_ops import CrossEntropyLoss
from mia_amd.optim import AdamW
from mia_amd.src.model import FoundationModel

dtype = sys.argv[1] if len(sys.argv) > 1 else "fp32"
torch.manual_seed(0)
eager = FoundationModel(backbone="efficientnet_b3", pretrained=False, dropout=0.0, classification_heads={"d": 4},
                        use_attention=True, compute_dtype=dtype).cuda().train()
for m in eager.modules():
    if isinstance(m, (torch.nn.Dropout, torch.nn.Dropout2d)):
        m.p = 0.0
graphed = copy.deepcopy(eager)
crit = CrossEntropyLoss()
opt_e = AdamW(eager.parameters(), lr=1e-3, weight_decay=1e-4)
opt_g = AdamW(graphed.parameters(), lr=1e-3, weight_decay=1e-4)
x = torch.randn(4, 3, 96, 96, device="cuda")
y = torch.randint(0, 4, (4,), device="cuda")
step = GraphedTrainStep(graphed, "classification", "d", crit, opt_g, x, y)
for i in range(3):
    opt_e.zero_grad(set_to_none=True)
    oe = eager(x, "classification", "d")
    le = crit(oe, y)
    le.backward()
    opt_e.step()
    lg = step(x, y)
    print(i, "loss eager", float(le), "graph", float(lg))
    with torch.no_grad():
        fe, fg = eager(x, "classification", "d"), graphed(x, "classification", "d")
    print("  eager-mode forward of both after the step:", fe[0].tolist(), fg[0].tolist())
    bworst = sorted(((float((b1.float() - b2.float()).abs().max()), n1) for (n1, b1), (_, b2) in
                     zip(eager.named_buffers(), graphed.named_buffers())), reverse=True)[:4]
    print("  worst buffer diffs", bworst)
    nonfinite = [n for n, p in graphed.named_parameters() if not torch.isfinite(p).all()]
    print("  graphed non-finite params:", nonfinite[:6], len(nonfinite))
    nonfinite = [n for n, p in eager.named_parameters() if not torch.isfinite(p).all()]
    print("  eager non-finite params:", nonfinite[:6], len(nonfinite))
    st = opt_g.state[next(iter(graphed.parameters()))]
    print("  words", step.words[:8].tolist(), step.words[4:8].view(torch.float32).tolist(), "step", st["step"])
    print("  logits eager", oe[0].tolist(), "\n  logits graph", step.outputs[0].tolist())
    names = [n for n, _ in eager.named_parameters()]
    worst = sorted(((float((pe.grad - pg.grad).abs().max() / (pe.grad.abs().max() + 1e-12)), n)
                    for n, pe, pg in zip(names, eager.parameters(), graphed.parameters())
                    if pe.grad is not None and pg.grad is not None), reverse=True)[:5]
    print("  worst grad diffs (rel to max)", worst)
    worstp = sorted(((float((pe - pg).abs().max()), n) for n, pe, pg in zip(names, eager.parameters(),
                                                                            graphed.parameters())), reverse=True)[:5]
    print("  worst param diffs", worstp)

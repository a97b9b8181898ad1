"""cProfile of the host side of one training step (batch 8): where the enqueue time goes. The backward pass runs on
the calling thread here (autograd's device thread is switched off) so that the profile sees it."""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import mia_amd
from mia_amd.head_ops import CrossEntropyLoss
from mia_amd.optim import AdamW
from mia_amd.src.model import FoundationModel

torch.autograd.set_multithreading_enabled(False)
dev = torch.device("cuda", 0)
model = FoundationModel(backbone="efficientnet_b3", pretrained=False, dropout=0.2, classification_heads={"s": 4},
                        use_attention=True, compute_dtype="bf16").to(dev).train()
opt = AdamW(model.parameters(), lr=1e-3)
crit = CrossEntropyLoss()
x = torch.randn(8, 3, 300, 300, device=dev)
y = torch.randint(0, 4, (8,), device=dev)
def step():
    opt.zero_grad(set_to_none=True)
    crit(model(x, "classification", "s"), y).backward()
    opt.step()
for _ in range(5): step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(10): step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(30)
st.sort_stats("cumulative").print_stats(45)

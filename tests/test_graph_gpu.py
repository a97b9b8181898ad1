"""HIP-graph replay of the eval forward gives the eager results bit for bit, follows in-place weight updates, and
refuses shapes it was not captured for."""
import pytest
import torch

pytestmark = pytest.mark.gpu

import mia_amd  # noqa: E402,F401
from mia_amd.graphs import GraphedForward  # noqa: E402
from mia_amd.src.model import FoundationModel  # noqa: E402


@pytest.mark.parametrize("task,dtype", [("classification", "fp32"), ("classification", "bf16"), ("segmentation", "bf16")])
def test_graph_replay_equals_eager(task, dtype):
    torch.manual_seed(0)
    model = FoundationModel(backbone="efficientnet_b3", pretrained=False, dropout=0.2,
                            classification_heads={"d": 4} if task == "classification" else {},
                            segmentation_heads={"d": 2} if task == "segmentation" else {}, use_attention=True,
                            compute_dtype=dtype).cuda().eval()
    xs = [torch.randn(2, 3, 96, 96, device="cuda") for _ in range(3)]
    g = GraphedForward(model, task, "d", xs[0])
    with torch.no_grad():
        for x in xs:
            assert torch.equal(g(x), model(x, task, "d"))
        # in-place weight change (what load_state_dict / an optimizer step do): the replay must see it
        for p in model.parameters():
            p.mul_(1.01)
        assert torch.equal(g(xs[1]), model(xs[1], task, "d"))
    with pytest.raises(ValueError):
        g(torch.randn(3, 3, 96, 96, device="cuda"))
    model.train()
    with pytest.raises(RuntimeError):
        g(xs[0])


def test_graph_needs_eval_mode():
    model = FoundationModel(backbone="efficientnet_b3", pretrained=False, classification_heads={"d": 2}).cuda().train()
    with pytest.raises(RuntimeError, match="eval"):
        GraphedForward(model, "classification", "d", torch.randn(1, 3, 64, 64, device="cuda"))

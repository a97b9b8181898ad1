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


# ------------------------------------------------------------------------------------------------
# whole training step (forward, loss, backward, AdamW) as one replay
# ------------------------------------------------------------------------------------------------
import copy  # noqa: E402

from mia_amd.graphs import GraphedTrainStep  # noqa: E402
from mia_amd.head_ops import CrossEntropyLoss  # noqa: E402
from mia_amd.optim import AdamW  # noqa: E402


def _set_dropout(model, p):
    for m in model.modules():
        if isinstance(m, (torch.nn.Dropout, torch.nn.Dropout2d)):
            m.p = p


def _cls_model(dtype, dropout):
    torch.manual_seed(0)
    model = FoundationModel(backbone="efficientnet_b3", pretrained=False, dropout=dropout,
                            classification_heads={"d": 4}, use_attention=True, compute_dtype=dtype).cuda().train()
    _set_dropout(model, dropout)
    return model


@pytest.mark.parametrize("fork", [True, False])
@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_graphed_train_step_follows_eager(dtype, fork):
    """Without dropout a graphed run and an eager run from the same start are the same computation: every kernel of
    the step adds its partial results in a fixed order (no float atomics), the replayed AdamW reads the same
    step-dependent factors the eager launch forms, so losses, parameters, optimizer state and BatchNorm buffers agree
    BIT FOR BIT after 5 steps in either precision, including a learning-rate change made between steps. fork=True:
    the graph carries the weight-gradient branch on a second stream (fork and join captured), as the eager step does."""
    eager = _cls_model(dtype, 0.0)
    graphed = copy.deepcopy(eager)
    start = [p.detach().clone() for p in graphed.parameters()]
    crit = CrossEntropyLoss()
    opt_e = AdamW(eager.parameters(), lr=1e-3, weight_decay=1e-4)
    opt_g = AdamW(graphed.parameters(), lr=1e-3, weight_decay=1e-4)
    g = torch.Generator(device="cuda").manual_seed(1)
    batches = [(torch.randn(4, 3, 96, 96, device="cuda", generator=g),
                torch.randint(0, 4, (4,), device="cuda", generator=g)) for _ in range(5)]
    step = GraphedTrainStep(graphed, "classification", "d", crit, opt_g, *batches[0], fork=fork)
    assert step.fork == fork
    # construction leaves training where it was
    for p, s0 in zip(graphed.parameters(), start):
        assert torch.equal(p, s0)
    for (n1, b1), (n2, b2) in zip(eager.named_buffers(), graphed.named_buffers()):
        assert torch.equal(b1, b2), n1
    assert all(int(opt_g.state[p]["step"]) == 0 for p in graphed.parameters() if p in opt_g.state)

    for i, (x, y) in enumerate(batches):
        if i == 3:
            opt_e.param_groups[0]["lr"] = opt_g.param_groups[0]["lr"] = 2.5e-4
        opt_e.zero_grad(set_to_none=True)
        le = crit(eager(x, "classification", "d"), y)
        le.backward()
        opt_e.step()
        lg = step(x, y)
        assert float(le.detach()) == float(lg.detach()), (i, float(le.detach()), float(lg.detach()))
    moved = max(float((p.detach() - s0).abs().max()) for p, s0 in zip(eager.parameters(), start))
    assert moved > 1e-3
    for (n, pe), pg in zip(eager.named_parameters(), graphed.parameters()):
        assert torch.equal(pe, pg), n
        if pg in opt_g.state:
            assert int(opt_g.state[pg]["step"]) == 5
            assert torch.equal(opt_e.state[pe]["exp_avg_sq"], opt_g.state[pg]["exp_avg_sq"]), n
    for (n1, b1), (n2, b2) in zip(eager.named_buffers(), graphed.named_buffers()):
        assert torch.equal(b1, b2), n1


def test_graphed_train_step_draws_new_dropout_masks():
    """lr = 0 freezes the weights, so the loss of a repeated batch changes only through the dropout masks: replays
    must differ from each other, and the sequence must repeat under torch.manual_seed."""
    model = _cls_model("bf16", 0.5)
    crit = CrossEntropyLoss()
    opt = AdamW(model.parameters(), lr=0.0, weight_decay=0.0)
    x = torch.randn(4, 3, 96, 96, device="cuda")
    y = torch.tensor([0, 1, 2, 3], device="cuda")
    step = GraphedTrainStep(model, "classification", "d", crit, opt, x, y)
    torch.manual_seed(7)
    a = [float(step(x, y)) for _ in range(4)]
    torch.manual_seed(7)
    b = [float(step(x, y)) for _ in range(4)]
    assert len(set(a)) == 4, a
    assert a == b, (a, b)


def test_graphed_train_step_refusals():
    model = _cls_model("bf16", 0.1)
    crit = CrossEntropyLoss()
    x, y = torch.randn(2, 3, 64, 64, device="cuda"), torch.tensor([0, 1], device="cuda")
    with pytest.raises(TypeError):
        GraphedTrainStep(model, "classification", "d", crit, torch.optim.AdamW(model.parameters()), x, y)
    opt = AdamW(model.parameters(), lr=1e-3)
    with pytest.raises(RuntimeError, match="no CPU path"):
        GraphedTrainStep(model, "classification", "d", crit, opt, x.cpu(), y.cpu())
    model.eval()
    with pytest.raises(RuntimeError, match="train"):
        GraphedTrainStep(model, "classification", "d", crit, opt, x, y)
    model.train()
    step = GraphedTrainStep(model, "classification", "d", crit, opt, x, y)
    with pytest.raises(ValueError):
        step(torch.randn(3, 3, 64, 64, device="cuda"), torch.tensor([0, 1, 2], device="cuda"))

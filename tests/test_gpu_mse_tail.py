"""pfn_mpn_backward_mse (include/pfn_hip.h): `loss = MSELoss()(out, y); loss.backward()` of train_epoch's per-batch body
(/root/reference/utils/training.py:59-74) riding in the first launch of the model's backward pass -- the last layer's
graph-resident EdgeAggregation backward forms the output rows, the loss and its gradient itself, so the output Linear's launch
(lin_out4) and the loss launch (pfn_mse_loss) leave the step.  Held against the three-call path it replaces: `out` and every
gradient bit for bit, the loss to the rounding of another summation order; and against the CPU oracle."""
import pytest
import torch

from tests.util import assert_close

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _models(p=0.2, seed=6):
    from poweflownet_amd.networks.MPN import MaskEmbdMultiMPN
    torch.manual_seed(seed)
    m = MaskEmbdMultiMPN(4, 2, 4, 129, 4, 3, p).to(DEV)
    return m


def _run(m, d, loss_fn, attach, x_grad=False):
    from poweflownet_amd import _lib as L
    m.seed_dropout(78)
    m.zero_grad(set_to_none=True)
    d.x.grad = None
    d.x.requires_grad_(x_grad)
    L.profile_report(reset=True)
    L.profile_enable(True)
    if attach:
        loss_fn.attach(m, d.y)
    out = m(d)
    loss = loss_fn(out, d.y)
    loss.backward(loss_fn.unit_grad(loss))
    torch.cuda.synchronize()
    L.profile_enable(False)
    rep = L.profile_report(reset=True)
    launches = {k: v["count"] for k, v in rep.items() if not k.startswith("__")}
    return {"out": out.detach().clone(), "loss": loss.detach().clone(), "g": m.flat_grad().clone(),
            "gx": d.x.grad.clone() if x_grad else None, "launches": launches}


@pytest.mark.parametrize("case,B,train", [("118v2", 128, True), ("118v2", 16, False), ("14", 37, True), ("14", 300, True)])
def test_attached_mse_is_bit_identical_to_the_three_call_path(case, B, train):
    from poweflownet_amd.loss import MSELoss
    from poweflownet_amd.synth import make_batch
    m = _models()
    m.train(train)
    d = make_batch(case, B, seed=1).to(DEV)
    loss_fn = MSELoss()
    plain = _run(m, d, loss_fn, attach=False, x_grad=True)
    for rep in range(3):      # repeated passes: the arrival counter is re-armed by every launch
        fused = _run(m, d, loss_fn, attach=True, x_grad=True)
        lf = fused["launches"]
        assert lf.get("ea_seg_bwd+out+mse") == 1 and "lin_out4" not in lf, lf
        assert plain["launches"].get("lin_out4") == 1 and "ea_seg_bwd+out+mse" not in plain["launches"], plain["launches"]
        assert torch.isfinite(fused["out"]).all() and fused["out"].abs().max() > 0
        assert torch.equal(fused["out"], plain["out"]), (fused["out"] - plain["out"]).abs().max().item()
        assert torch.equal(fused["g"], plain["g"]), (fused["g"] - plain["g"]).abs().max().item()
        assert torch.equal(fused["gx"], plain["gx"])
        a, b = fused["loss"].item(), plain["loss"].item()
        assert abs(a - b) <= 2e-6 * abs(b), (a, b)


def test_attached_mse_against_the_cpu_oracle():
    """The same pair against oracle/ref_cpu.py (the reference dataflow on the CPU), eval mode: loss and every gradient."""
    from oracle import ref_cpu
    from poweflownet_amd.loss import MSELoss
    from poweflownet_amd.networks.MPN import MaskEmbdMultiMPN
    from poweflownet_amd.synth import make_batch
    torch.manual_seed(3)
    ref = ref_cpu.MaskEmbdMultiMPN(4, 2, 4, 129, 4, 3, 0.0).eval()
    m = MaskEmbdMultiMPN(4, 2, 4, 129, 4, 3, 0.0)
    m.load_state_dict(ref.state_dict())
    m = m.to(DEV).eval()
    data = make_batch("118v2", 8, seed=5)
    out_ref = ref(data)
    loss_ref = torch.nn.MSELoss()(out_ref, data.y)
    loss_ref.backward()
    d = data.to(DEV)
    loss_fn = MSELoss()
    loss_fn.attach(m, d.y)
    out = m(d)
    loss = loss_fn(out, d.y)
    loss.backward(loss_fn.unit_grad(loss))
    assert_close(out.cpu(), out_ref.detach(), what="attached MSELoss: out")
    assert_close(loss.cpu(), loss_ref.detach(), what="attached MSELoss: loss")
    for (name, p), (_, q) in zip(m.named_parameters(), ref.named_parameters()):
        assert_close(p.grad.cpu(), q.grad, rtol=2e-5, what=f"attached MSELoss: grad {name}")


def test_attach_is_transparent_where_the_tail_does_not_apply_and_loud_on_misuse():
    from poweflownet_amd.loss import MSELoss
    from poweflownet_amd.synth import make_batch
    m = _models(p=0.0).eval()
    loss_fn = MSELoss()
    # a batch beyond the graph-resident regime (2,048 graphs): attach is consumed, the plain path runs
    d = make_batch("118v2", 2048, seed=2).to(DEV)
    big = _run(m, d, loss_fn, attach=True)
    assert "ea_seg_bwd+out+mse" not in big["launches"], big["launches"]
    assert torch.isfinite(big["loss"])
    # a no_grad forward consumes the announcement and writes its output
    d = make_batch("118v2", 8, seed=2).to(DEV)
    loss_fn.attach(m, d.y)
    with torch.no_grad():
        o = m(d)
    assert getattr(o, "_pfn_mse_tail", None) is None and torch.isfinite(o).all() and m._mse_attach is None
    # the loss called with another target than the attached one: error, not garbage
    loss_fn.attach(m, d.y)
    o = m(d)
    with pytest.raises(RuntimeError, match="another target"):
        loss_fn(o, d.y.clone())
    # a scaled backward through the attached loss: error
    loss_fn.attach(m, d.y)
    o = m(d)
    loss = loss_fn(o, d.y)
    with pytest.raises(RuntimeError, match="unit_grad"):
        (loss * 2.0).backward()


def test_graphed_train_step_uses_the_tail_and_matches_the_eager_three_call_loop():
    """GraphedTrainStep (the replayed per-batch body) announces its MSELoss; three AdamW steps land on the parameters of the same
    three steps run call by call without the announcement, bit for bit."""
    from poweflownet_amd.loss import MSELoss
    from poweflownet_amd.optim import FlatAdamW
    from poweflownet_amd.synth import make_batch
    from poweflownet_amd.utils.training import GraphedTrainStep
    d = make_batch("118v2", 32, seed=4).to(DEV)
    finals, losses = [], []
    for graphed in (True, False):
        m = _models(seed=11).train()
        m.seed_dropout(5)
        opt = FlatAdamW(m, lr=1e-3)
        loss_fn = MSELoss()
        step = GraphedTrainStep(m, loss_fn, opt) if graphed else None
        ls = []
        for _ in range(3):
            if graphed:
                ls.append(step(d).item())
            else:
                opt.zero_grad()
                loss = loss_fn(m(d), d.y)
                loss.backward(loss_fn.unit_grad(loss))
                opt.step()
                ls.append(loss.item())
        if graphed:
            assert step.graph is not None and not step.disabled
        finals.append(opt.flat_param.detach().clone())
        losses.append(ls)
    assert torch.equal(finals[0], finals[1]), (finals[0] - finals[1]).abs().max().item()
    for a, b in zip(*losses):
        assert abs(a - b) <= 2e-6 * abs(b), (a, b)


def test_attached_pair_leaves_no_reference_cycle():
    """The arrangement hangs on the output tensor and in two autograd nodes: none of them may close a cycle (a cycle keeps the
    step's whole workspace alive until the garbage collector happens to run -- 432 MB per step at case118v2 x 128)."""
    import gc
    from poweflownet_amd.loss import MSELoss
    from poweflownet_amd.synth import make_batch
    m = _models().train()
    d = make_batch("118v2", 64, seed=9).to(DEV)
    loss_fn = MSELoss()
    gc.collect()
    gc.disable()
    try:
        seen = []
        for _ in range(6):
            m.zero_grad(set_to_none=True)
            loss_fn.attach(m, d.y)
            loss = loss_fn(m(d), d.y)
            loss.backward(loss_fn.unit_grad(loss))
            del loss
            torch.cuda.synchronize()
            seen.append(torch.cuda.memory_allocated())
    finally:
        gc.enable()
    assert seen[-1] <= seen[1], seen


# ------------------------------------------------------------------------------------------------ Masked_L2_loss
def _run_masked(m, d, loss_fn, attach):
    from poweflownet_amd import _lib as L
    m.seed_dropout(78)
    m.zero_grad(set_to_none=True)
    L.profile_report(reset=True)
    L.profile_enable(True)
    if attach:
        loss_fn.attach(m, d.y, d.pred_mask)
    out = m(d)
    loss = loss_fn(out, d.y, d.pred_mask)
    loss.backward(loss_fn.unit_grad(loss))
    torch.cuda.synchronize()
    L.profile_enable(False)
    rep = L.profile_report(reset=True)
    return {"out": out.detach().clone(), "loss": loss.detach().clone(), "g": m.flat_grad().clone(),
            "launches": {k: v["count"] for k, v in rep.items() if not k.startswith("__")}}


@pytest.mark.parametrize("case,B,train,float_mask,reg,coeff", [("118v2", 128, True, False, True, 1), ("118v2", 16, False, True, True, 0.5),
                                                                ("14", 37, True, False, False, 1), ("14", 300, True, True, True, 2.0)])
def test_attached_masked_l2_is_bit_identical_to_the_plain_path(case, B, train, float_mask, reg, coeff):
    """pfn_mpn_backward_masked_l2: the reference's default training loss (/root/reference/utils/custom_loss_functions.py:10-46)
    riding in the backward pass's first launch, the two means' denominators counted by the forward pass's first launch."""
    from poweflownet_amd.synth import make_batch
    from poweflownet_amd.utils.custom_loss_functions import Masked_L2_loss
    m = _models()
    m.train(train)
    d = make_batch(case, B, seed=1).to(DEV)
    if float_mask:
        d.pred_mask = d.pred_mask.float()
    loss_fn = Masked_L2_loss(regularize=reg, regcoeff=coeff)
    plain = _run_masked(m, d, loss_fn, attach=False)
    assert "ea_seg_bwd+out+masked_l2" not in plain["launches"] and plain["launches"].get("lin_out4") == 1
    for _ in range(3):
        fused = _run_masked(m, d, loss_fn, attach=True)
        assert fused["launches"].get("ea_seg_bwd+out+masked_l2") == 1 and "lin_out4" not in fused["launches"], fused["launches"]
        assert torch.isfinite(fused["out"]).all() and fused["g"].abs().max() > 0
        assert torch.equal(fused["out"], plain["out"])
        assert torch.equal(fused["g"], plain["g"]), (fused["g"] - plain["g"]).abs().max().item()
        a, b = fused["loss"].item(), plain["loss"].item()
        assert abs(a - b) <= 2e-6 * abs(b), (a, b)


def test_attached_masked_l2_against_torch_and_misuse():
    """The attached loss value against the reference's formula in plain torch on the stored outputs; a mask that is not the model's
    pred_mask falls back to the plain path; another mask at the loss call raises."""
    from poweflownet_amd.synth import make_batch
    from poweflownet_amd.utils.custom_loss_functions import Masked_L2_loss
    m = _models(p=0.0).eval()
    d = make_batch("118v2", 8, seed=7).to(DEV)
    loss_fn = Masked_L2_loss(regularize=True, regcoeff=0.25)
    r = _run_masked(m, d, loss_fn, attach=True)
    mk = d.pred_mask.bool()
    want = torch.nn.functional.mse_loss(r["out"][mk], d.y[mk]) + 0.25 * torch.nn.functional.mse_loss(r["out"][~mk], d.y[~mk])
    assert abs(r["loss"].item() - want.item()) <= 5e-6 * abs(want.item())
    # an all-zero mask: the first mean is over nothing -> NaN, as torch's
    d0 = make_batch("118v2", 8, seed=7).to(DEV)
    d0.pred_mask = torch.zeros_like(d0.pred_mask)
    r0 = _run_masked(m, d0, Masked_L2_loss(), attach=True)
    assert torch.isnan(r0["loss"])
    other = d.pred_mask.clone()
    loss_fn.attach(m, d.y, other)                  # not the tensor the model reads: consumed, plain path
    out = m(d)
    assert getattr(out, "_pfn_mse_tail", None) is None
    loss_fn.attach(m, d.y, d.pred_mask)
    out = m(d)
    with pytest.raises(RuntimeError, match="another target"):
        loss_fn(out, d.y, other)


def test_graphed_train_step_with_masked_l2_matches_the_eager_loop():
    from poweflownet_amd.optim import FlatAdamW
    from poweflownet_amd.synth import make_batch
    from poweflownet_amd.utils.custom_loss_functions import Masked_L2_loss
    from poweflownet_amd.utils.training import GraphedTrainStep
    d = make_batch("118v2", 32, seed=4).to(DEV)
    finals = []
    for graphed in (True, False):
        m = _models(seed=11).train()
        m.seed_dropout(5)
        opt = FlatAdamW(m, lr=1e-3)
        loss_fn = Masked_L2_loss()
        step = GraphedTrainStep(m, loss_fn, opt) if graphed else None
        for _ in range(3):
            if graphed:
                step(d)
            else:
                opt.zero_grad()
                loss = loss_fn(m(d), d.y, d.pred_mask)
                loss.backward(loss_fn.unit_grad(loss))
                opt.step()
        finals.append(opt.flat_param.detach().clone())
    assert torch.equal(finals[0], finals[1]), (finals[0] - finals[1]).abs().max().item()

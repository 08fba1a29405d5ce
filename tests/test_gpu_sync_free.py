"""The proposal stage and everything behind it WITHOUT a host read of its sizes (include/gpn.h section DEV, network/model.py
``sync_free_proposals``): buffers at their bounds, row counts as device counters, the previous step's counts as the plan that
only sizes grids.  Reference behaviour = the blocking path (every stage boundary of model.py:228-462 reads a size back), which
the golden-pipeline tests pin to the reference model; here the two paths of THIS repo are compared on the GPU:
  * plan = the exact counts  -> the same kernel variants run: loss and every gradient BIT-equal;
  * stale plans (far too small / far too large) -> results do not depend on the plan (grid-stride walks);
  * a step without any proposal, and a step whose plan said "no proposals" but which has some."""
import copy

import pytest
import torch

from gapartnet_amd.smoke import make_batch, make_model

pytestmark = pytest.mark.gpu
JITTER = (torch.tensor([0.3, 0.6, 0.1]), torch.tensor([0.5, 0.2, 0.9]))


def _step(model, batch, cuda):
    model.revoxelize_jitter = tuple(j.to(cuda) for j in JITTER)
    loss = model.training_step(batch, 0)
    loss.backward()
    torch.cuda.synchronize()
    grads = {k: (p.grad.clone() if p.grad is not None else None) for k, p in model.named_parameters()}
    buffers = {k: b.clone() for k, b in model.named_buffers()}
    return loss.detach().clone(), grads, buffers


def _counts_of(base, batch, cuda):
    probe = copy.deepcopy(base)
    probe.sync_free_proposals = True
    _step(probe, batch, cuda)  # first step of a model: no plan yet -> blocking read, which leaves the plan behind
    assert probe._prop_plan is not None
    return list(probe._prop_plan)


@pytest.mark.parametrize("n_scenes,n_points", [(2, 5000), (8, 20000), (4, 50000)])  # (the last two: BASELINE configs 3 and 4 as the bench runs them)
def test_sync_free_step_with_the_exact_plan_is_bit_equal(cuda, n_scenes, n_points):
    batch = [pc.to(cuda) for pc in make_batch(n_scenes, n_points, seed0=640)]
    base = make_model((0, 0), channels=[16, 32, 48] if n_points < 10000 else None).to(cuda)
    ref = copy.deepcopy(base)
    ref.sync_free_proposals = False
    want = _step(ref, batch, cuda)
    plan = _counts_of(base, batch, cuda)
    assert plan[1] > 0 and plan[2] > 0, "the case needs proposals"
    model = copy.deepcopy(base)
    model.sync_free_proposals = True
    model._prop_plan = list(plan)
    got = _step(model, batch, cuda)
    assert model._prop_pending, "the step took the path without a read (its counts are on their way to pinned memory)"
    assert torch.equal(got[0], want[0]), (float(got[0]), float(want[0]))
    for k in want[1]:
        assert (got[1][k] is None) == (want[1][k] is None), k
        if want[1][k] is not None:
            assert torch.equal(got[1][k], want[1][k]), f"gradient of {k}"
    for k in want[2]:
        assert torch.equal(got[2][k], want[2][k]), f"buffer {k}"
    model._take_over_proposal_counts(wait=True)
    assert model._prop_plan[:4] == plan[:4] and model._prop_plan[6] == plan[6]


@pytest.mark.parametrize("scale", [0.02, 40.0])
def test_sync_free_results_do_not_depend_on_the_plan(cuda, scale):
    batch = [pc.to(cuda) for pc in make_batch(2, 5000, seed0=640)]
    base = make_model((0, 0), channels=[16, 32, 48]).to(cuda)
    ref = copy.deepcopy(base)
    ref.sync_free_proposals = False
    want = _step(ref, batch, cuda)
    plan = _counts_of(base, batch, cuda)
    model = copy.deepcopy(base)
    model.sync_free_proposals = True
    model._prop_plan = [max(int(c * scale), 0) for c in plan]
    got = _step(model, batch, cuda)
    assert abs(float(got[0]) - float(want[0])) <= 1e-5 * max(1.0, abs(float(want[0])))
    for k in want[1]:
        if want[1][k] is None:
            continue
        a, b = got[1][k], want[1][k]
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-5 * float(b.abs().max()) + 1e-9), f"gradient of {k}: {float((a - b).abs().max())}"


def test_sync_free_step_without_any_proposal(cuda):
    """every point predicted as background: the device counters are zero, every launch behind the proposal stage is a no-op -
    finite loss equal to the blocking path's, backbone / point-head gradients equal, proposal networks receive ZERO gradients
    (the blocking path: none - it never runs them), BatchNorm statistics of the proposal networks untouched"""
    base = make_model((0, 0), channels=[16, 32, 48]).to(cuda)
    with torch.no_grad():
        base.sem_seg_head.weight.zero_()
        base.sem_seg_head.bias.fill_(-10.0)
        base.sem_seg_head.bias[0] = 10.0
    batch = [pc.to(cuda) for pc in make_batch(2, 4000)]
    ref = copy.deepcopy(base)
    ref.sync_free_proposals = False
    want = _step(ref, batch, cuda)
    model = copy.deepcopy(base)
    model.sync_free_proposals = True
    model._prop_plan = [0] * 7
    got = _step(model, batch, cuda)
    assert torch.isfinite(got[0]) and torch.equal(got[0], want[0])
    for k, g in want[1].items():
        if g is not None:
            assert torch.equal(got[1][k], g), k
        elif k.startswith(("score_unet", "npcs_unet", "score_head", "npcs_head")):
            assert got[1][k] is None or float(got[1][k].abs().max()) == 0.0, k
    for k, b in want[2].items():
        assert torch.equal(got[2][k], b), f"buffer {k} (running statistics / batch counters of a network that had no rows)"


def test_sync_free_plan_of_an_empty_step_then_a_step_with_proposals(cuda):
    """the plan says "nothing" (previous step had no proposals), this step has hundreds: grids sized for nothing, grid-stride
    walks do the rest - same results as the blocking path"""
    batch = [pc.to(cuda) for pc in make_batch(2, 5000, seed0=640)]
    base = make_model((0, 0), channels=[16, 32, 48]).to(cuda)
    ref = copy.deepcopy(base)
    ref.sync_free_proposals = False
    want = _step(ref, batch, cuda)
    model = copy.deepcopy(base)
    model.sync_free_proposals = True
    model._prop_plan = [0] * 7
    got = _step(model, batch, cuda)
    assert abs(float(got[0]) - float(want[0])) <= 1e-5 * max(1.0, abs(float(want[0])))
    for k in want[1]:
        if want[1][k] is not None:
            a, b = got[1][k], want[1][k]
            assert torch.allclose(a, b, rtol=1e-4, atol=1e-5 * float(b.abs().max()) + 1e-9), k


def test_three_sync_free_steps_train_like_three_blocking_steps(cuda):
    """a short training run (optimizer steps in between, plans taken over from earlier steps whenever their counts have arrived):
    same losses as the run with the blocking read in every step, to fp tolerance (kernel variants may differ with the plan)"""
    scenes = [[pc.to(cuda) for pc in make_batch(2, 5000, seed0=700 + 10 * j)] for j in range(4)]
    base = make_model((0, 0), channels=[16, 32, 48]).to(cuda)
    runs = []
    for sync_free in (False, True):
        model = copy.deepcopy(base)
        model.sync_free_proposals = sync_free
        model.revoxelize_jitter = tuple(j.to(cuda) for j in JITTER)
        opt = model.configure_optimizers()
        losses = []
        for i, batch in enumerate(scenes):
            opt.zero_grad(set_to_none=True)
            loss = model.training_step(batch, i)
            loss.backward()
            opt.step()
            losses.append(loss.detach())
        torch.cuda.synchronize()
        runs.append(torch.stack(losses).cpu())
        if sync_free:
            assert model._prop_plan is not None
    assert torch.allclose(runs[0], runs[1], rtol=1e-4, atol=1e-5), (runs[0], runs[1])


def test_a_step_without_proposals_leaves_the_proposal_networks_to_the_optimizer_untouched(cuda):
    """ the reference never runs ScoreNet / NPCS-Net on a batch without proposals: their gradients stay None and
    torch.optim.Adam skips them - value, moments and step count unchanged.  The device-counted step runs them over zero rows (zero
    gradients); FusedAdam's gate (gpn_adam_step_gated on the step's own proposal counter) must leave them exactly as the
    blocking path does - also AFTER a momentum has built up - and the step that follows must use the un-advanced step number."""
    def background(model, on):
        with torch.no_grad():
            if on:
                model._saved_head = (model.sem_seg_head.weight.clone(), model.sem_seg_head.bias.clone())
                model.sem_seg_head.weight.zero_()
                model.sem_seg_head.bias.fill_(-10.0)
                model.sem_seg_head.bias[0] = 10.0
            else:
                model.sem_seg_head.weight.copy_(model._saved_head[0])
                model.sem_seg_head.bias.copy_(model._saved_head[1])

    scenes = [[pc.to(cuda) for pc in make_batch(2, 5000, seed0=900 + 10 * j)] for j in range(4)]
    base = make_model((0, 0), channels=[16, 32, 48]).to(cuda)
    finals = []
    for sync_free in (False, True):
        model = copy.deepcopy(base)
        model.sync_free_proposals = sync_free
        model.revoxelize_jitter = tuple(j.to(cuda) for j in JITTER)
        opt = model.configure_optimizers()
        snapshots = []
        for i, batch in enumerate(scenes):
            empty = i == 2  # steps 0, 1 build up moments; step 2 has no proposal; step 3 has some again
            if empty:
                background(model, True)
            opt.zero_grad(set_to_none=True)
            model.training_step(batch, i).backward()
            if empty:  # (the point head itself is put back before the update so that both runs update the same values)
                for p in (model.sem_seg_head.weight, model.sem_seg_head.bias):
                    p.grad = None
            opt.step()
            if empty:
                background(model, False)
            torch.cuda.synchronize()
            snapshots.append({k: p.detach().clone() for k, p in model.named_parameters()
                              if k.startswith(("score_unet", "npcs_unet", "score_head", "npcs_head"))})
        if sync_free:
            assert model._prop_gate is not None, "the device-counted step ran (otherwise this test compares nothing)"
        sd = opt.state_dict()
        index_of = {p: i for i, p in enumerate(p for g in opt.param_groups for p in g["params"])}
        finals.append((snapshots, {k: float(sd["state"][index_of[p]]["step"]) for k, p in model.named_parameters()
                                   if index_of[p] in sd["state"]}))
    (blocking, sd_b), (gated, sd_g) = finals
    for k in blocking[2]:
        assert torch.equal(blocking[2][k], blocking[1][k]), f"reference behaviour: {k} untouched by the empty step"
        assert torch.equal(gated[2][k], gated[1][k]), f"{k} moved in a step without proposals"
    for k in blocking[3]:  # the step after: same update as the blocking run (same bias corrections = same step number)
        assert torch.allclose(gated[3][k], blocking[3][k], rtol=1e-4, atol=1e-6), k
    differ = {k: (sd_b[k], sd_g.get(k)) for k in sd_b if sd_b[k] != sd_g.get(k)}
    assert not differ and len(sd_b) == len(sd_g), f"state_dict step counts (the gated tensors skipped one step): {differ}"


def test_sync_free_is_not_used_when_a_proposal_network_runs_module_by_module(cuda):
    """ only the native executor reads a tensor's device row count: with use_native_executor = False on a proposal
    U-Net the model keeps the blocking read (same results as the blocking path over two steps), and handing the per-layer path
    a device-counted tensor raises instead of normalising over unwritten rows"""
    from gapartnet_amd.hip_ops import DevCount
    from gapartnet_amd.spconv import pytorch as spconv
    scenes = [[pc.to(cuda) for pc in make_batch(2, 5000, seed0=940 + 10 * j)] for j in range(2)]
    base = make_model((0, 0), channels=[16, 32, 48]).to(cuda)
    base.score_unet.use_native_executor = False
    runs = []
    for sync_free in (False, True):
        model = copy.deepcopy(base)
        model.score_unet.use_native_executor = False  # (a class attribute shadowed on the instance: deepcopy keeps it)
        model.sync_free_proposals = sync_free
        model.revoxelize_jitter = tuple(j.to(cuda) for j in JITTER)
        opt = model.configure_optimizers()
        losses = []
        for i, batch in enumerate(scenes):
            opt.zero_grad(set_to_none=True)
            loss = model.training_step(batch, i)
            loss.backward()
            opt.step()
            losses.append(loss.detach())
        assert model._prop_gate is None, "no device-counted step may have run"
        runs.append(torch.stack(losses).cpu())
    assert torch.equal(runs[0], runs[1]), (runs[0], runs[1])
    # the guard itself
    model = copy.deepcopy(base)
    x = spconv.SparseConvTensor(torch.zeros((64, 16), device=cuda), torch.zeros((64, 4), dtype=torch.int32, device=cuda),
                                spatial_shape=[28, 28, 28], batch_size=1)
    x.rows_dev = DevCount(torch.tensor([3], dtype=torch.int64, device=cuda), 3)
    with pytest.raises(RuntimeError, match="device counter"):
        model.score_unet(x)

"""The check behind ``__graft_entry__.smoke()``: one small full-pipeline train step on the GPU (forward + backward through the
HIP library), compared with the same step run through the CPU oracle (``oracle.torch_ops``).  Test infrastructure: the oracle
is the checker here, never the product path - which is why this file lives under ``tests/`` and not in the package."""
import copy

import torch

from gapartnet_amd import backend
from gapartnet_amd.smoke import make_batch, make_model


def run_smoke(device: torch.device, n_scenes: int = 2, n_points: int = 4000, tol: float = 2e-3) -> dict:
    from oracle import torch_ops as oracle_ops  # checker only

    model = make_model((0, 0), channels=[16, 32, 48, 64])
    jitter = (torch.tensor([0.3, 0.6, 0.1]), torch.tensor([0.5, 0.2, 0.9]))
    batch = make_batch(n_scenes, n_points)

    ref_model = copy.deepcopy(model)
    ref_model.revoxelize_jitter = jitter
    with backend.using(oracle_ops):
        ref_loss = ref_model.training_step(batch, 0)
        ref_loss.backward()

    model = model.to(device)
    model.revoxelize_jitter = tuple(j.to(device) for j in jitter)
    loss = model.training_step([pc.to(device) for pc in batch], 0)
    loss.backward()
    torch.cuda.synchronize(device)

    got, want = float(loss), float(ref_loss)
    assert abs(got - want) <= tol * max(1.0, abs(want)), f"smoke: loss {got} vs oracle {want}"
    worst, worst_name = compare_gradients(model, ref_model, rel=5e-4)
    print(f"smoke ok: loss {got:.6f} (oracle {want:.6f}), worst gradient error {worst:.2e} x max|g| at {worst_name}")
    return dict(loss=got, oracle_loss=want, worst_grad_rel_err=worst)


def compare_gradients(model, ref_model, rel: float = 5e-4):
    """per parameter tensor: max|g - g_ref| <= rel * max|g_ref| (north_star asks 1e-4 on features; gradients pass through
    ~200 BatchNorm layers in training mode, whose batch statistics on the tiny deep levels amplify fp32 summation-order
    noise, hence 5e-4 on gradients - measured: 5e-6).  A tensor whose reference gradient is structurally zero - a bias in front of a
    BatchNorm: the mean subtraction removes it - is not divided by its own noise: it must be ~zero on both sides,
    measured against the largest gradient entry of the whole model.  -> (worst ratio, its parameter name)"""
    ref = {n: q.grad for n, q in ref_model.named_parameters()}
    top = max(float(g.abs().max()) for g in ref.values() if g is not None)
    worst, worst_name = 0.0, ""
    for name, p in model.named_parameters():
        g_ref = ref[name]
        if g_ref is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, name
            continue
        assert p.grad is not None, f"{name}: no gradient on the HIP path"
        g = p.grad.detach().cpu()
        scale = float(g_ref.abs().max())
        if scale <= 1e-6 * top:
            assert float(g.abs().max()) <= 1e-5 * top, f"{name}: structurally zero gradient is {float(g.abs().max()):.3e}"
            continue
        ratio = float((g - g_ref).abs().max()) / scale
        if ratio > worst:
            worst, worst_name = ratio, name
    assert worst <= rel, f"gradient mismatch vs oracle: {worst:.3e} x max|g| at {worst_name} (bound {rel})"
    return worst, worst_name

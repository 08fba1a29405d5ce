"""PointNet++ SA / FP modules (gapartnet_amd/pointnet2/pointnet2_modules.py).

CPU: through the oracle's point operators (the line-by-line restatement of the vendored kernels) against an independent
plain-torch evaluation written here (cdist ball query with first-nsample-by-index and first-hit padding, 3-NN by topk),
plus the reference's state_dict key names.  GPU: the HIP kernels against the oracle path, values and gradients."""
import pytest
import torch
import torch.nn.functional as F

from gapartnet_amd import backend
from gapartnet_amd.pointnet2.pointnet2_modules import PointnetFPModule, PointnetSAModule, PointnetSAModuleMSG
from oracle import torch_ops


def _cloud(b=2, n=96, c=5, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(b, n, 3, generator=g), torch.randn(b, c, n, generator=g)


def _ball_groups(xyz, centres, radius, nsample):
    """(B, M, nsample) indices: first nsample points by index with d2 < r2, padded with the first hit (ball_query_gpu.cu:9-45)"""
    d2 = ((centres[:, :, None, :] - xyz[:, None, :, :]) ** 2).sum(-1)
    out = torch.zeros(xyz.shape[0], centres.shape[1], nsample, dtype=torch.long)
    for b in range(xyz.shape[0]):
        for m in range(centres.shape[1]):
            hits = torch.nonzero(d2[b, m] < radius * radius).squeeze(1)[:nsample]
            if hits.numel():
                out[b, m, :] = hits[0]
                out[b, m, :hits.numel()] = hits
    return out


def test_sa_module_matches_a_plain_torch_evaluation():
    xyz, feats = _cloud()
    torch.manual_seed(1)
    sa = PointnetSAModule(mlp=[5, 8, 12], npoint=16, radius=0.35, nsample=8).eval()
    with backend.using(torch_ops):
        new_xyz, out = sa(xyz, feats)
    assert new_xyz.shape == (2, 16, 3) and out.shape == (2, 12, 16)
    idx = _ball_groups(xyz, new_xyz, 0.35, 8)
    gathered_xyz = torch.stack([xyz[b][idx[b]] for b in range(2)]) - new_xyz[:, :, None, :]       # (B, M, S, 3)
    gathered_f = torch.stack([feats[b].t()[idx[b]] for b in range(2)])                             # (B, M, S, C)
    stacked = torch.cat([gathered_xyz, gathered_f], dim=-1).permute(0, 3, 1, 2)                     # (B, 3+C, M, S)
    want = sa.mlps[0](stacked).amax(dim=3)
    assert torch.allclose(out, want, atol=1e-5)


def test_msg_and_group_all_shapes_and_keys():
    xyz, feats = _cloud()
    spec = [[5, 8], [5, 6, 10]]
    msg = PointnetSAModuleMSG(npoint=12, radii=[0.2, 0.4], nsamples=[4, 8], mlps=spec, pool_method="avg_pool")
    assert spec == [[5, 8], [5, 6, 10]], "the caller's spec must not be widened in place"
    keys = set(msg.state_dict())
    assert {"mlps.0.layer0.conv.weight", "mlps.0.layer0.bn.bn.weight", "mlps.0.layer0.bn.bn.running_mean",
            "mlps.1.layer1.conv.weight", "mlps.1.layer1.bn.bn.num_batches_tracked"} <= keys
    assert not any(k.endswith("conv.bias") for k in keys), "no conv bias in front of a BatchNorm (pytorch_utils.py:57)"
    assert msg.mlps[0].layer0.conv.weight.shape == (8, 8, 1, 1)
    with backend.using(torch_ops):
        new_xyz, out = msg(xyz, feats)
        whole = PointnetSAModule(mlp=[5, 7], bn=False)(xyz, feats)
    assert out.shape == (2, 18, 12)
    assert whole[0] is None and whole[1].shape == (2, 7, 1)
    assert "mlps.0.layer0.conv.bias" in PointnetSAModule(mlp=[5, 7], bn=False).state_dict()


def test_fp_module_matches_a_plain_torch_evaluation():
    unknown, skip = _cloud(n=40, c=4, seed=3)
    known, kf = _cloud(n=12, c=6, seed=4)
    torch.manual_seed(2)
    fp = PointnetFPModule(mlp=[10, 9]).eval()
    with backend.using(torch_ops):
        out = fp(unknown, known, skip, kf)
        only_known = fp.__class__(mlp=[6, 3]).eval()(unknown, None, None, kf[:, :, :1])
    d = torch.cdist(unknown, known)
    dist, idx = d.topk(3, dim=2, largest=False)
    w = 1.0 / (dist + 1e-8)
    w = w / w.sum(2, keepdim=True)
    interp = torch.stack([(kf[b][:, idx[b]] * w[b][None]).sum(-1) for b in range(2)])                # (B, C2, n)
    want = fp.mlp(torch.cat([interp, skip], dim=1).unsqueeze(-1)).squeeze(-1)
    assert out.shape == (2, 9, 40) and torch.allclose(out, want, atol=1e-4)
    assert only_known.shape == (2, 3, 40)


@pytest.mark.gpu
def test_modules_on_the_gpu_match_the_oracle_path(cuda):
    xyz, feats = _cloud(b=2, n=512, c=6, seed=7)
    torch.manual_seed(5)
    sa = PointnetSAModuleMSG(npoint=64, radii=[0.15, 0.3], nsamples=[8, 16], mlps=[[6, 16], [6, 16, 24]])
    fp = PointnetFPModule(mlp=[40 + 6, 32])
    results = []
    for dev, ops in (("cpu", torch_ops), (cuda, None)):
        sa_d, fp_d = sa.to(dev).train(), fp.to(dev).train()
        x, f = xyz.to(dev), feats.detach().clone().to(dev).requires_grad_(True)
        ctx = backend.using(ops) if ops is not None else backend.using(backend.raw())
        with ctx:
            centres, coarse = sa_d(x, f)
            dense = fp_d(x, centres, f, coarse)
            dense.square().mean().backward()
        results.append((centres.detach().cpu(), coarse.detach().cpu(), dense.detach().cpu(), f.grad.detach().cpu()))
        sa_d.zero_grad(); fp_d.zero_grad()
    (c0, k0, d0, g0), (c1, k1, d1, g1) = results
    assert torch.equal(c0, c1), "furthest point sampling picks the same centres"
    assert torch.allclose(k0, k1, atol=1e-4) and torch.allclose(d0, d1, atol=1e-4) and torch.allclose(g0, g1, atol=1e-4)

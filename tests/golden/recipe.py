"""Recipes shared by the fixture generator (make_golden_pipeline.py, build container only) and the tests that consume
the fixtures: how the deterministic model weights and the input scenes of ``glue_step.npz`` / ``loader.npz`` are made.
No reference code involved — this file is pure data-generation logic, so it travels to the GPU box."""
import zlib

import numpy as np
import torch

PIPELINE_SCENES = ((4101, "Box"), (4102, "StorageFurniture"))   # (seed, category prefix of the pc_id)
PIPELINE_POINTS = 4000
PIPELINE_TORCH_SEED = 77           # torch.manual_seed right before each reference step: fixes the two torch.rand(3) jitters
LOADER_SEEDS = (11, 12, 13, 14, 15, 16)
LOADER_POINTS = 600
AUG = dict(pos_jitter=0.1, color_jitter=0.3, flip_prob=0.3, rotate_prob=0.3)   # gapartnet.yaml data.init_args


def name_keyed_state(module: torch.nn.Module) -> dict:
    """Deterministic state_dict for ``module``: every tensor is drawn from a generator seeded by the CRC32 of its
    state_dict key, so two module trees with the same keys/shapes get the same values regardless of construction order
    or of how many random numbers their constructors consumed.  Conv / linear weights ~ N(0, 1/fan_in)-ish, BatchNorm
    affine and running statistics perturbed away from (1, 0, 0, 1) so eval-mode BatchNorm is not an identity."""
    out = {}
    for key, ref in module.state_dict().items():
        g = torch.Generator().manual_seed(zlib.crc32(key.encode()))
        if key.endswith("num_batches_tracked"):
            out[key] = torch.zeros_like(ref)
        elif key.endswith("running_var"):
            out[key] = 0.5 + torch.rand(ref.shape, generator=g)
        elif key.endswith("running_mean"):
            out[key] = 0.1 * torch.randn(ref.shape, generator=g)
        elif ref.dim() == 1 and key.endswith("weight"):                       # BatchNorm gamma
            out[key] = 1.0 + 0.1 * torch.randn(ref.shape, generator=g)
        elif ref.dim() == 1:                                                  # biases (BatchNorm beta, linear bias)
            out[key] = 0.1 * torch.randn(ref.shape, generator=g)
        else:
            fan_in = ref.numel() // ref.shape[0]       # linear [out, in]; sparse conv [Cout, k, k, k, Cin]
            out[key] = torch.randn(ref.shape, generator=g) * (1.5 / np.sqrt(fan_in))
        if key.startswith("offset_head.3."):
            out[key] = out[key] * 0.02    # predicted centre offsets of ~1 cm: the shifted point set still forms clusters
        out[key] = out[key].to(ref.dtype)
    return out


def scene_arrays(seed: int, n_points: int):
    """the reference's ``.pth`` 6-tuple for one synthetic scene, with the instance ids made NON-contiguous
    (id -> 3 id + 2) so that compact_instance_labels has something to do."""
    from gapartnet_amd.dataset import synthetic
    xyz, rgb, sem, ins, npcs, pix = synthetic.make_scene_arrays(seed, n_points, parts_range=(8, 12))
    ins = np.where(ins >= 0, ins * 3 + 2, ins).astype(np.int32)
    # colours that vary smoothly over the surface (the generator's are white noise): a randomly initialised network then
    # predicts spatially coherent classes, so ball query + CCL find real multi-point clusters at this small scene size
    rgb = (0.5 + 0.5 * np.sin(xyz @ np.array([[5.0, 1.0, -2.0], [-1.5, 4.0, 2.5], [2.0, -3.0, 4.5]]) + seed)).astype(np.float32)
    return xyz, rgb, sem, ins, npcs, pix

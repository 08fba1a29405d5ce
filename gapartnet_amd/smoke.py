"""Model / batch factories of the default configuration (gapartnet.yaml's values): what ``bench.py``, the tests and the tools
build their synthetic workloads from.  (The smoke CHECK of ``__graft_entry__.smoke()`` - this step compared with the same step
through the CPU checker - lives in ``tests/smoke_check.py``: nothing in this package imports the checker.)"""
import copy

import torch

DEFAULT_CFG = dict(
    in_channels=6, num_part_classes=10, backbone_type="SparseUNet",
    backbone_cfg=dict(channels=[16, 32, 48, 64, 80, 96, 112], block_repeat=2),
    instance_seg_cfg=dict(ball_query_radius=0.04, max_num_points_per_query=50, min_num_points_per_proposal=5,
                          max_num_points_per_query_shift=300, score_fullscale=28, score_scale=50),
    learning_rate=1e-3, ignore_sem_label=-100, use_sem_focal_loss=True, use_sem_dice_loss=True,
    training_schedule=[5, 10], val_nms_iou_threshold=0.3, val_ap_iou_threshold=0.5,
    symmetry_indices=[0, 1, 3, 3, 2, 0, 3, 2, 4, 1], visualize_cfg=dict(visualize=False), debug=True)


def make_model(training_schedule=(0, 0), channels=None, seed: int = 0):
    from .network.model import GAPartNet
    cfg = copy.deepcopy(DEFAULT_CFG)
    cfg["training_schedule"] = list(training_schedule)
    if channels is not None:
        cfg["backbone_cfg"]["channels"] = list(channels)
    torch.manual_seed(seed)
    return GAPartNet(**cfg)


def make_batch(n_scenes: int, n_points: int, seed0: int = 1000):
    from .dataset.gapartnet import SyntheticGAPartNetDataset
    ds = SyntheticGAPartNetDataset(n_scenes, n_points=n_points, seed0=seed0)
    return [ds[i] for i in range(n_scenes)]

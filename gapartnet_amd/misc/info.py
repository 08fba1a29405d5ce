"""Category / part tables and the NPCS symmetry groups (reference: gapartnet/misc/info.py).

The id tables are the dataset's label contract (misc/info.py:6-101).  The five symmetry groups used by the NPCS
loss (misc/info.py:104-346, selected per part class by ``symmetry_indices`` in gapartnet.yaml:34) are generated
here from their definition instead of being spelled out:
  group 0: {I, I}            group 1: {I, Rz(180deg)}          group 2: {I, Ry(180deg)}
  group 3: the 12 rotations about z by k*30deg
  group 4: group 3 followed by the 12 improper maps [[sin t, cos t, 0], [cos t, -sin t, 0], [0, 0, -1]], t = k*30deg, k=1..12
tests/test_golden.py pins the generated tensors against values captured from the reference module.
"""
import math
from typing import List, Tuple

import torch

_OBJECTS = ["Box", "Remote", "Microwave", "Camera", "Dishwasher", "WashingMachine", "CoffeeMachine", "Toaster",
            "StorageFurniture", "AKBBucket", "AKBBox", "AKBDrawer", "AKBTrashCan", "Bucket", "Keyboard", "Printer",
            "Toilet", "KitchenPot", "Safe", "Oven", "Phone", "Refrigerator", "Table", "TrashCan", "Door", "Laptop",
            "Suitcase"]
OBJECT_NAME2ID = {name: i for i, name in enumerate(_OBJECTS)}

TARGET_PARTS = ["others", "line_fixed_handle", "round_fixed_handle", "slider_button", "hinge_door", "slider_drawer",
                "slider_lid", "hinge_lid", "hinge_knob", "revolute_handle"]
PART_NAME2ID = {name: i for i, name in enumerate(TARGET_PARTS)}
PART_ID2NAME = {i: name for i, name in enumerate(TARGET_PARTS)}
TARGET_IDX = list(range(len(TARGET_PARTS)))
PI = math.pi


def _rot_z(t: float) -> List[List[float]]:
    return [[math.cos(t), math.sin(t), 0.0], [-math.sin(t), math.cos(t), 0.0], [0.0, 0.0, 1.0]]


def _flip(t: float) -> List[List[float]]:
    return [[math.sin(t), math.cos(t), 0.0], [math.cos(t), -math.sin(t), 0.0], [0.0, 0.0, -1.0]]


_I = [[1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, 1.0]]
SYMMETRY_MATRIX = [
    [_I, _I],
    [_I, [[-1.0, 0.0, 0.0], [0.0, -1.0, 0.0], [0.0, 0.0, 1.0]]],
    [_I, [[-1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, -1.0]]],
    [_rot_z(PI * k / 6) for k in range(12)],
    [_rot_z(PI * k / 6) for k in range(12)] + [_flip(PI * k / 6) for k in range(1, 13)],
]


def get_symmetry_matrix() -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """-> ([3,2,3,3] groups 0-2, [1,12,3,3] group 3, [1,24,3,3] group 4), float32 (misc/info.py:338-346)."""
    first = torch.as_tensor(SYMMETRY_MATRIX[:3], dtype=torch.float32)
    second = torch.as_tensor(SYMMETRY_MATRIX[3:4], dtype=torch.float32)
    third = torch.as_tensor(SYMMETRY_MATRIX[4:5], dtype=torch.float32)
    return first, second, third

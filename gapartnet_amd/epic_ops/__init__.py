"""Mirror of the ``epic_ops`` functions GAPartNet imports (network/grouping_utils.py:4-8, network/model.py:11-12,
dataset/gapartnet.py:11), implemented over libgpn_hip.so.  Same module paths, names, argument order and return
tuples, so reference-style glue code runs unchanged against this package."""
from . import ball_query, ccl, iou, nms, reduce, voxelize  # noqa: F401

"""gapartnet_amd — MI355X-native implementation of the GAPartNet sparse-conv perception hot path.

Layout:
  csrc/ + libgpn_hip.so   hand-written HIP kernels behind the C-ABI of include/gpn.h
  hip_ops                 raw torch-tensor front-end of that C-ABI (no autograd)
  functional              autograd wrappers (sparse conv, row gather, segmented max-pool)
  spconv.pytorch          mirror of the spconv classes the reference model uses
  epic_ops.*              mirror of the seven epic_ops functions the reference model uses
  pointnet2               mirror of the vendored pointnet2_utils wrappers
  structure / dataset / network / misc   host glue with the reference's names and signatures
"""
import os as _os

# HIP maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), round-robin; streams that share a queue
# execute in submission order.  This path uses three streams of its own (training, batch preparation, weight gradients);
# an RCCL communicator adds its own, and with the default the batch-preparation stream then shares a queue with the
# training stream - its host reads wait behind the whole backbone instead of overlapping it (+2.2 ms on a 12.6 ms step,
# tools/exchange_phase_probe.py).  Read by the HIP runtime at its first call, so setting it at import is early enough.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

__version__ = "0.1.0"

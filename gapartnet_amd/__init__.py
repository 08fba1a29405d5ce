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
__version__ = "0.1.0"

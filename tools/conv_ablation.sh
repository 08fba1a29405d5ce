#!/bin/bash
# What bounds the direct conv kernel?  Builds libgpn_hip.so variants with parts of spconv_fwd_direct_kernel compiled out
# (GPN_ABL in csrc/spconv_fwd.hip) into _variants/ and times the bench's level shapes with each (tools/conv_probe.py).
#   here (no GPU):   tools/conv_ablation.sh build
#   on the GPU box:  tools/conv_ablation.sh run
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result -Wno-unused-value -mllvm -amdgpu-mfma-vgpr-form=1"
if [ "${1:-build}" = build ]; then
  mkdir -p $R/_variants
  make -s -C $R/gapartnet_amd/csrc -j8
  for A in 1 2 3 4; do
    /opt/rocm/bin/hipcc $FLAGS -DGPN_ABL=$A -c $R/gapartnet_amd/csrc/spconv_fwd.hip -o /tmp/spconv_fwd_abl$A.o 2> /dev/null &&
      (cd $R/gapartnet_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls _build/*.o | grep -v spconv_fwd.o) /tmp/spconv_fwd_abl$A.o -o $R/_variants/libgpn_abl$A.so) &
  done
  wait
  ls -la $R/_variants
else
  cp $R/gapartnet_amd/libgpn_hip.so /tmp/libgpn_full.so
  echo "== full kernel"; python $R/tools/conv_probe.py
  for A in 1 2 3 4; do
    echo "== GPN_ABL=$A"; cp $R/_variants/libgpn_abl$A.so $R/gapartnet_amd/libgpn_hip.so; python $R/tools/conv_probe.py
  done
  cp /tmp/libgpn_full.so $R/gapartnet_amd/libgpn_hip.so
fi

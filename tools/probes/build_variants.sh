#!/bin/bash
# Builds measurement variants of the library into tools/probes/_build/libgpn_<tag>.so (git-ignored; travels to the GPU box):
#   build_variants.sh "<tag> <file.hip>[,<file.hip>...] <flags>" ...
# e.g. build_variants.sh "trace spconv_fwd.hip,spconv_msplit.hip -DGPN_SPLIT_TRACE=1 -DGPN_MSPLIT_TRACE=1" "regs96 spconv_msplit.hip -DGPN_MSPLIT_OPERAND_REGS=96"
set -eu
R=$(cd "$(dirname "$0")/../.." && pwd)
C=$R/gapartnet_amd/csrc
O=$R/tools/probes/_build
mkdir -p "$O"
make -C "$C" -s -j 8
for spec in "$@"; do
  set -- $spec; tag=$1; files=$2; shift 2
  ( objs=""; skip=""
    for f in ${files//,/ }; do
      b=${f%.hip}
      vf=""; case $b in spconv_fwd|spconv_tiles|spconv_msplit) vf="-mllvm -amdgpu-mfma-vgpr-form=1";; esac  # (as csrc/Makefile)
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result -Wno-unused-value \
        $vf "$@" -c "$C/$f" -o "$O/${b}_$tag.o" 2>/dev/null
      objs="$objs $O/${b}_$tag.o"; skip="$skip -e /$b.o"
    done
    rest=$(ls "$C"/_build/*.o | grep -v $skip)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $rest $objs -o "$O/libgpn_$tag.so"; rm -f $objs; echo "built $tag" ) &
done
wait

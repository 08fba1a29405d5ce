#!/bin/bash
# Builds measurement variants of the masked-tile conv kernel (compile-time switches GPN_TILES_ABL / GPN_TILES_OPERAND_REGS of
# csrc/spconv_tiles.hip) into tools/probes/_build/libgpn_<tag>.so (git-ignored; travels to the GPU box), to be run with
#   GPN_PROBE_SO=tools/probes/_build/libgpn_<tag>.so python tools/conv_tiles_bench.py
# usage: tiles_ablation.sh "<tag> <flags>" ...     e.g.  tiles_ablation.sh "r64 -DGPN_TILES_OPERAND_REGS=64" "r64_noB -DGPN_TILES_ABL=2"
set -eu
R=$(cd "$(dirname "$0")/../.." && pwd)
C=$R/gapartnet_amd/csrc
O=$R/tools/probes/_build
mkdir -p "$O"
make -C "$C" -s -j 8
for spec in "$@"; do
  set -- $spec; tag=$1; shift
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result -Wno-unused-value \
      -mllvm -amdgpu-mfma-vgpr-form=1 "$@" -c "$C/spconv_tiles.hip" -o "$O/spconv_tiles_$tag.o" 2>/dev/null
    objs=$(ls "$C"/_build/*.o | grep -v spconv_tiles.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs "$O/spconv_tiles_$tag.o" -o "$O/libgpn_$tag.so"; rm -f "$O/spconv_tiles_$tag.o"; echo "built $tag" ) &
done
wait

#!/bin/bash
# A/B of two builds of the library on ONE GPU box, interleaved (box-to-box spread is larger than most single changes):
#   tools/ab_same_box.sh _ab/libgpn_hip_base.so _ab/libgpn_hip_new.so [rounds]
# prints the bench's ms per step for each build and round.
R=${GRAFT_REPO_ROOT:-/root/repo}
A=$R/$1; B=$R/$2; N=${3:-3}
cp "$R/gapartnet_amd/libgpn_hip.so" /tmp/libgpn_keep.so
for i in $(seq $N); do
  for v in A B; do
    if [ $v = A ]; then cp "$A" "$R/gapartnet_amd/libgpn_hip.so"; else cp "$B" "$R/gapartnet_amd/libgpn_hip.so"; fi
    timeout 200 python "$R/bench.py" --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['ms_per_step'],3), 'ms', round(d['value'],1), 'pc/s')"
  done
done
cp /tmp/libgpn_keep.so "$R/gapartnet_amd/libgpn_hip.so"

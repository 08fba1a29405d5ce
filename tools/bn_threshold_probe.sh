#!/bin/bash
# BatchNorm category of the bench under rocprofv3 for libgpn variants in _variants/ (A/B of kernel thresholds)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cp $R/gapartnet_amd/libgpn_hip.so /tmp/libgpn_keep.so
for v in "$@"; do
  cp $R/_variants/libgpn_$v.so $R/gapartnet_amd/libgpn_hip.so
  rm -rf /tmp/pb_$v
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb_$v -o t -- python $R/bench.py --steps 16 --warmup 4 --no-cpu-baseline > /tmp/pb_$v.log 2>&1
  f=$(find /tmp/pb_$v -name "*kernel_stats.csv" | head -1)
  echo "== $v: $(tail -1 /tmp/pb_$v.log | cut -c60-130)"
  python $R/tools/gpu_categories.py "$f" 21 | grep -E "batchnorm|total"
done
cp /tmp/libgpn_keep.so $R/gapartnet_amd/libgpn_hip.so

#!/bin/bash
# temporary job: stretch wgrad, best variant, step A/B
cd /root/repo
mkdir -p gpurun_out/r05w
touch gapartnet_amd/csrc/spconv.hip
make -C gapartnet_amd/csrc -s -j 16 EXTRA="-DGPN_STRETCH_WAVES=8 -DGPN_STRETCH_SHRINK=2" > /dev/null 2>&1
: > gpurun_out/r05w/ab2.txt
for i in 1 2 3; do
  for v in 0 1; do
    echo "stretch=$v" >> gpurun_out/r05w/ab2.txt
    GPN_WGRAD_STRETCH=$v timeout 300 python bench.py --steps 60 --warmup 15 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])" >> gpurun_out/r05w/ab2.txt
  done
done
cat gpurun_out/r05w/ab2.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof -o st -- python /root/repo/bench.py --steps 16 --warmup 4 --no-cpu-baseline > /dev/null 2>&1
cd /root/repo
f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1)
python tools/kernel_table.py $f 21 wgrad > gpurun_out/r05w/ktable_stretch.txt 2>&1
cat gpurun_out/r05w/ktable_stretch.txt

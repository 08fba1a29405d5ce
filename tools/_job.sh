#!/bin/bash
# temporary job: proposal sort A/B
R=/root/repo
O=$R/gpurun_out/r05p
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_proposals.py tests/test_gpu_sync_free.py tests/test_golden_pipeline.py -x -q -m gpu > $O/pytest.txt 2>&1
tail -2 $O/pytest.txt
cp gapartnet_amd/libgpn_hip.so /tmp/lib_onesweep.so
touch gapartnet_amd/csrc/proposals.hip
make -C gapartnet_amd/csrc -s -j 16 EXTRA="-DGPN_PROP_SORT_MERGE_LIMIT=1048576" > /dev/null 2>&1
cp gapartnet_amd/libgpn_hip.so /tmp/lib_merge.so
: > $O/ab.txt
for i in 1 2 3; do
  for v in merge onesweep; do
    cp /tmp/lib_$v.so gapartnet_amd/libgpn_hip.so
    echo "$v $(timeout 300 python bench.py --steps 60 --warmup 15 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])")" >> $O/ab.txt
  done
done
cat $O/ab.txt

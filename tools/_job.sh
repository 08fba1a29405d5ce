#!/bin/bash
cd /root/repo
O=gpurun_out/r05b2; mkdir -p $O
timeout 900 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
cp gapartnet_amd/libgpn_hip.so /tmp/new.so
cp tools/_old.so /tmp/old.so
: > $O/eval_ab.txt
for i in 1 2 3; do for v in old new; do cp /tmp/$v.so gapartnet_amd/libgpn_hip.so; timeout 200 python tools/eval_bench.py 2>/dev/null | tail -1 | python -c "import sys,json;d=json.loads(sys.stdin.read());print('eval $v', round(d['value'],1), round(d['ms_per_validation_step'],2))" >> $O/eval_ab.txt; done; done
cat $O/eval_ab.txt

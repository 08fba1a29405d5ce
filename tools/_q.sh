cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/pq; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/pq -o t -- python $R/bench.py --steps 16 --warmup 4 --no-cpu-baseline > /tmp/pq.log 2>&1
f=$(find /tmp/pq -name "*kernel_trace.csv" | head -1)
python $R/tools/queue_breakdown.py $f 21 > $R/gpurun_out/queue_breakdown_full.txt
sed -n 1,70p $R/gpurun_out/queue_breakdown_full.txt | cut -c1-110

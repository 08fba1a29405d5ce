cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
GPN_BENCH_FORCE_GRAD_SYNC=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf -o t -- python $R/bench.py --steps 16 --warmup 4 --no-cpu-baseline > /tmp/fs.log 2>&1
tail -1 /tmp/fs.log | cut -c1-200
f=$(find /tmp/pf -name "*kernel_stats.csv" | head -1)
python - "$f" <<'P'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n=r['Name']
    if any(w in n.lower() for w in ('nccl','rccl','allreduce','all_reduce','reduce_scatter','cat','copy')):
        print(n[:90], r['Calls'], r['TotalDurationNs'], r['AverageNs'])
P
python $R/tools/gpu_categories.py "$f" 21 | tail -16

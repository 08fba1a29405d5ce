cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/xcd; mkdir -p $O
timeout 200 python $R/bench.py --no-cpu-baseline 2>/dev/null | cut -c1-200
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --output-format csv -d /tmp/p_$c -o t -- python "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-validation > /dev/null 2>&1 < /dev/null
  f=$(find /tmp/p_$c -name "*counter_collection.csv" 2>/dev/null | head -1)
  if [ -n "$f" ]; then python "$R/tools/pmc_summary.py" "$f" spconv > "$O/pmc_$c.txt" 2>&1 < /dev/null; fi
done
(cd "$R" && python tools/pmc_traffic.py "$O/pmc_FETCH_SIZE.txt" "$O/pmc_WRITE_SIZE.txt" "$O/traffic.json" spconv_fwd); cat $O/traffic.json

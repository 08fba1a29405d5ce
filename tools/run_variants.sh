R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/variants; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for v in "" "--freeze-convs" "--no-bn-fuse" "--freeze-convs --no-bn-fuse"; do
  tag=$(echo "default$v" | tr -d ' ' ); rm -rf /tmp/pv
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/pv -o t -- python $R/tools/step_loop.py $v > $O/$tag.log 2>&1
  f=$(find /tmp/pv -name "*kernel_trace.csv" | head -1)
  python $R/tools/fwd_bwd_split.py $f > $O/$tag.txt 2>&1; tail -1 $O/$tag.txt; grep "last 6" $O/$tag.log
  python $R/tools/step_loop.py $v 2>/dev/null | tail -1
done

cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_msplit.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r6b_msplit_tests.txt
tail -3 gpurun_out/r6b_msplit_tests.txt
for v in "" regs96 regs144; do
  for c in "" --cold; do
    echo "### variant '${v:-default}' $c"
    GPN_PROBE_SO=${v:+tools/probes/_build/libgpn_$v.so} python tools/conv_msplit_sweep.py $c 2>&1 | grep -v amdgpu.ids
  done
done > gpurun_out/r6b_sweep.txt 2>&1
GPN_PROBE_SO=tools/probes/_build/libgpn_trace.so python tools/probes/msplit_trace.py > gpurun_out/r6b_trace_warm.txt 2>&1
GPN_PROBE_SO=tools/probes/_build/libgpn_trace.so python tools/probes/msplit_trace.py --cold > gpurun_out/r6b_trace_cold.txt 2>&1
for i in 1 2 3; do for m in 0 1; do echo "msplit=$m"; GPN_CONV_MSPLIT=$m python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), round(d['value'],1), d['roofline'].get('frac_raw_events'))"; done; done 2>&1 | tee gpurun_out/r6b_bench_ab.txt
python -m pytest tests -m gpu -q 2>&1 | tail -12 | tee gpurun_out/r6b_pytest_gpu.txt

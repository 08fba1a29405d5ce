cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_msplit.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r6a_msplit_tests.txt
tail -3 gpurun_out/r6a_msplit_tests.txt
python tools/conv_msplit_sweep.py > gpurun_out/r6a_sweep.txt 2>&1
cat gpurun_out/r6a_sweep.txt
python tools/conv_msplit_sweep.py --cold > gpurun_out/r6a_sweep_cold.txt 2>&1
for i in 1 2; do for m in 0 1; do echo "msplit=$m"; GPN_CONV_MSPLIT=$m python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), round(d['value'],1), d['roofline'].get('frac_raw_events'))"; done; done 2>&1 | tee gpurun_out/r6a_bench_ab.txt
python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee gpurun_out/r6a_pytest_gpu.txt

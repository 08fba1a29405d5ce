cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6c
O=gpurun_out/r6c
python -m pytest tests/test_gpu_proposals.py tests/test_gpu_sync_free.py tests/test_gpu_msplit.py -m gpu -q 2>&1 | tail -15 > $O/tests_a.txt; tail -3 $O/tests_a.txt
python -m pytest tests/test_gpu_model.py -m gpu -q -k "adam or two_rank" 2>&1 | tail -15 > $O/tests_b.txt; tail -3 $O/tests_b.txt
for a in "--steps 20 --warmup 5" "--steps 30 --warmup 10" "--steps 20 --warmup 5 --moving" "--steps 30 --warmup 10 --moving" "--steps 20 --warmup 5" "--steps 30 --warmup 10"; do
  echo "bench $a"; python bench.py --no-cpu-baseline $a 2>/dev/null | tail -1 > $O/bench_last.json
  python -c "import sys,json; d=json.loads(open('$O/bench_last.json').read()); print(round(d['ms_per_step'],3), round(d['value'],1), d['roofline'].get('frac_raw_events'), (d.get('proposal_stage') or {}).get('points_min_max'), (d.get('proposal_stage') or {}).get('proposals_min_max'))"
done 2>&1 | tee $O/bench_stationary.txt
cp $O/bench_last.json $O/bench_default_stationary.json
bash tools/mini_measure.sh > $O/mini.log 2>&1; tail -5 $O/mini.log
cp -r gpurun_out/mini $O/

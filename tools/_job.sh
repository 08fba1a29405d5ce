cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05c
O=gpurun_out/r05c
python -m pytest tests -m gpu -q 2>&1 | tail -40 > $O/pytest.txt; tail -12 $O/pytest.txt
timeout 300 python tools/eval_bench.py 2>$O/eval.err | tail -1 | tee $O/eval.json
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline 2>$O/bench_$i.err | tail -1 > $O/bench_$i.json; python -c "import sys,json; d=json.loads(open('$O/bench_$i.json').read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), round(d['value'],1))"; done

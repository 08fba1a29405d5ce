cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05b
O=gpurun_out/r05b
python -m pytest tests -m gpu -q -x 2>&1 | tail -30 > $O/pytest.txt; tail -5 $O/pytest.txt
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline 2>$O/bench_$i.err | tail -1 > $O/bench_$i.json; python -c "import sys,json; d=json.loads(open('$O/bench_$i.json').read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), round(d['value'],1)); print({k:(round(v['frac'],3) if 'frac' in v else None) for k,v in d['roofline'].get('all_kernels',{}).items()})"; done

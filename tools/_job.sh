cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05e
O=gpurun_out/r05e
python -m pytest tests -m gpu -q 2>&1 | tail -8 > $O/pytest.txt; tail -4 $O/pytest.txt
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), round(d['value'],1))"; done

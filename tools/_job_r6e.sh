cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6e
O=gpurun_out/r6e
python -m pytest tests/test_gpu_model.py -m gpu -q -k "inference or forced" 2>&1 | tail -15 > $O/tests_a.txt; tail -3 $O/tests_a.txt
for i in 1 2; do python tools/eval_bench.py --ab-bn-fusion --epochs 4 2>/dev/null | tail -1 | cut -c1-400; done | tee $O/eval_ab.txt
python tools/eval_bench.py 2>/dev/null | tail -1 | tee $O/eval_default.json | cut -c1-300
python bench.py --no-cpu-baseline --lib-knobs msplit=0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('msplit=0', round(d['ms_per_step'],3))"
python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default', round(d['ms_per_step'],3))"
python -m pytest tests -m gpu -q 2>&1 | tail -12 | tee $O/pytest_gpu.txt

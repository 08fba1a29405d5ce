cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q -k "prefetch or two_rank or training_mode_backbone" -s 2>&1 | grep -E "passed|failed|worst" | head
run() { env "$@" timeout 200 python bench.py --steps 60 --warmup 15 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', round(d['ms_per_step'],3))"; }
for r in 1 2 3 4; do run GPN_PREFETCH_DEFER=0; run GPN_PREFETCH_DEFER=1; done

cd $GRAFT_REPO_ROOT
GPN_DIRECT_MIN_TILES=1 python -m pytest tests -m gpu -x -q 2>&1 | tail -2 | cut -c1-250
b() { env "$@" python bench.py --steps 60 --warmup 15 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', round(d['ms_per_step'],3))"; }
for r in 1 2 3 4; do b GPN_DIRECT_MIN_TILES=16; b GPN_DIRECT_MIN_TILES=1; done

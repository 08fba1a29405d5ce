cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q -k "prefetch or model or golden or sync_free or loader" 2>&1 | tail -2 | cut -c1-300
b() { env "$@" python bench.py --steps 60 --warmup 15 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', round(d['ms_per_step'],3), d['host'])"; }
for r in 1 2 3 4; do b GPN_PREFETCH_THREAD=0; b GPN_PREFETCH_THREAD=1; done

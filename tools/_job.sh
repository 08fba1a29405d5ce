cd $GRAFT_REPO_ROOT
run() { env "$@" timeout 200 python bench.py --steps 60 --warmup 15 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', round(d['ms_per_step'],3))"; }
for r in 1 2 3; do run GPN_X=0; run GPN_TILE_ORDER_MIN_ROWS=16384; run GPN_TILE_ORDER_MIN_ROWS=4096; run GPN_WGRAD_PARTIAL_MB=2; run GPN_WGRAD_TARGET_WGS=2048; done

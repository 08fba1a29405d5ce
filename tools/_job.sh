cd $GRAFT_REPO_ROOT
b() { env "$@" python bench.py --steps 60 --warmup 15 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', round(d['ms_per_step'],3))"; }
for r in 1 2 3; do b GPN_WGRAD_PARTIAL_MB=4; b GPN_WGRAD_PARTIAL_MB=8; b GPN_WGRAD_PARTIAL_MB=16; b GPN_WGRAD_PARTIAL_MB=8 GPN_WGRAD_TARGET_WGS=8192; b GPN_WGRAD_PARTIAL_MB=16 GPN_WGRAD_TARGET_WGS=8192 GPN_WGRAD_ROWS_PER_SLICE=64; done

cd $GRAFT_REPO_ROOT
GPN_WGRAD_ATOMIC=1 python -m pytest tests -m gpu -x -q -k "wgrad or weight_gradient or golden or model or conv" 2>&1 | tail -3 | cut -c1-300
b() { env "$@" python bench.py --steps 60 --warmup 15 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', round(d['ms_per_step'],3))"; }
for r in 1 2 3; do b GPN_WGRAD_ATOMIC=0; b GPN_WGRAD_ATOMIC=1; b GPN_WGRAD_ATOMIC=1 GPN_WGRAD_TARGET_WGS=8192; b GPN_WGRAD_ATOMIC=1 GPN_WGRAD_PARTIAL_MB=2; done

R=$GRAFT_REPO_ROOT
cd $R
run() { env "$@" python bench.py --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', round(d['ms_per_step'],3))"; }
for r in 1 2 3; do run GPN_WGRAD_BATCH_MB=96; run GPN_WGRAD_BATCH_MB=32; run GPN_WGRAD_BATCH_MB=12; run GPN_WGRAD_TARGET_WGS=2048; run GPN_WGRAD_TARGET_WGS=8192; run GPN_WGRAD_ROWS_PER_SLICE=256; done

cd $GRAFT_REPO_ROOT
run() { env "$@" timeout 200 python bench.py --steps 60 --warmup 15 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', round(d['ms_per_step'],3))"; }
for r in 1 2 3; do run GPN_X=0; run GPN_BENCH_STREAM=1; run GPN_BENCH_STREAM=1 GPN_WGRAD_CU_MASK=mod8:8; run GPN_BENCH_STREAM=1 GPN_WGRAD_CU_MASK=mod8:4; run GPN_BENCH_STREAM=1 GPN_WGRAD_CU_MASK=mod8:2; run GPN_BENCH_STREAM=1 GPN_WGRAD_CU_MASK=first:2; run GPN_BENCH_STREAM=1 GPN_WGRAD_CU_MASK=first:4; done

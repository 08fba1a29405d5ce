cd $GRAFT_REPO_ROOT
b() { (cd $1 && python bench.py --steps 60 --warmup 15 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$2', round(d['ms_per_step'],3), round(d['value'],1))"); }
for r in 1 2 3 4; do b _ab/r3 round3_final_4c9128f; b . round4_final; done

cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q -k "golden or model or sync_free or configs" 2>&1 | tail -2 | cut -c1-300
b() { env "$@" python bench.py --steps 60 --warmup 15 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', round(d['ms_per_step'],3), round(d['host']['cpu_ms_per_step'],2))"; }
for r in 1 2 3; do b A=1; done
python tools/aten_sites.py 2>/dev/null | grep -E "thread|per step" | head -40

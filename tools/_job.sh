cd $GRAFT_REPO_ROOT
b() { env "$@" python $SO bench.py --steps 60 --warmup 15 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$SO', round(d['ms_per_step'],3))"; }
for r in 1 2 3 4; do
SO=""; b A=1
SO="tools/probes/with_so.py tools/probes/_build/libgpn_old.so"; b A=1
done

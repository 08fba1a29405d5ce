cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6d
O=gpurun_out/r6d
python -m pytest tests/test_gpu_model.py -m gpu -q -k "inference or adam or native_executor or paired" 2>&1 | tail -15 > $O/tests_a.txt; tail -3 $O/tests_a.txt
for v in "" nohints; do echo "### variant '${v:-default (hints)}'"; GPN_PROBE_SO=${v:+tools/probes/_build/libgpn_$v.so} python tools/conv_msplit_sweep.py 2>&1 | grep "L2 25190 rows" | cut -c1-200; done | tee $O/sweep_hints.txt
cat > /tmp/with_so_bench.py <<'PY'
import os, sys, runpy
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from gapartnet_amd import _C
if os.environ.get("GPN_PROBE_SO"): _C.SO_PATH = os.path.abspath(os.environ["GPN_PROBE_SO"])
sys.argv = ["bench.py", "--no-cpu-baseline"]
runpy.run_path(os.path.join(os.environ["GRAFT_REPO_ROOT"], "bench.py"), run_name="__main__")
PY
for i in 1 2 3; do for v in "" nohints; do echo "variant '${v:-hints}'"; GPN_PROBE_SO=${v:+tools/probes/_build/libgpn_$v.so} python /tmp/with_so_bench.py 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), round(d['value'],1), d['roofline'].get('frac_raw_events'))"; done; done 2>&1 | tee $O/bench_hints_ab.txt
for i in 1 2 3; do for f in 1 0; do echo "eval GPN_BN_FUSE=$f"; GPN_BN_FUSE=$f python tools/eval_bench.py 2>/dev/null | tail -1 | cut -c1-200; done; done 2>&1 | tee $O/eval_ab.txt
python -m pytest tests -m gpu -q 2>&1 | tail -12 | tee $O/pytest_gpu.txt

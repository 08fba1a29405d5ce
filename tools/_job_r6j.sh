cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6j
O=gpurun_out/r6j
for v in "" strided; do echo "### variant '${v:-contiguous tap ranges (product)}'"; GPN_PROBE_SO=${v:+tools/probes/_build/libgpn_$v.so} python tools/conv_msplit_sweep.py 2>&1 | grep -E "rows|down|up" | sed 's/nt1\/sp9.*//' | cut -c1-200; done | tee $O/strided_sweep.txt
cat > /tmp/with_so.py <<'PY'
import os, sys, runpy
root = os.environ["GRAFT_REPO_ROOT"]
sys.path.insert(0, root)
from gapartnet_amd import _C
if os.environ.get("GPN_PROBE_SO"): _C.SO_PATH = os.path.abspath(os.environ["GPN_PROBE_SO"])
script = sys.argv[1]; sys.argv = sys.argv[1:]
runpy.run_path(os.path.join(root, script), run_name="__main__")
PY
for i in 1 2 3; do for v in "" strided; do echo "variant '${v:-contiguous}'"; GPN_PROBE_SO=${v:+tools/probes/_build/libgpn_$v.so} python /tmp/with_so.py bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), round(d['value'],1), d['roofline'].get('frac_raw_events'))"; done; done 2>&1 | tee $O/bench_strided_ab.txt
GPN_PROBE_SO=tools/probes/_build/libgpn_strided.so python /tmp/with_so.py /usr/bin/env true 2>/dev/null
cd $GRAFT_REPO_ROOT && GPN_PROBE_SO=tools/probes/_build/libgpn_strided.so python - <<'PY' 2>&1 | tail -5 | tee $O/strided_tests.txt
import os, sys
sys.path.insert(0, os.getcwd())
from gapartnet_amd import _C
_C.SO_PATH = os.path.abspath(os.environ["GPN_PROBE_SO"])
import pytest
sys.exit(pytest.main(["tests/test_gpu_msplit.py", "tests/test_gpu_model.py", "tests/test_golden_pipeline.py", "-m", "gpu", "-q", "-x", "--deselect", "tests/test_gpu_msplit.py::test_masked_split_kernel_fwd_dgrad_every_cut"]))
PY

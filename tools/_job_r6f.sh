cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6f
O=gpurun_out/r6f
python -m pytest tests/test_gpu_model.py tests/test_golden_pipeline.py tests/test_gpu_proposals.py -m gpu -q -k "inference or golden or validation or post_processing" 2>&1 | tail -15 > $O/tests_a.txt; tail -3 $O/tests_a.txt
for i in 1 2; do python tools/eval_bench.py --ab-bn-fusion --epochs 5 2>/dev/null | tail -1 | cut -c1-500; done | tee $O/eval_ab.txt
cat > /tmp/with_so.py <<'PY'
import os, sys, runpy
root = os.environ["GRAFT_REPO_ROOT"]
sys.path.insert(0, root)
from gapartnet_amd import _C
if os.environ.get("GPN_PROBE_SO"): _C.SO_PATH = os.path.abspath(os.environ["GPN_PROBE_SO"])
script = sys.argv[1]; sys.argv = sys.argv[1:]
runpy.run_path(os.path.join(root, script), run_name="__main__")
PY
for v in "" wgrad1; do echo "### wgrad variant '${v:-two tiles ahead (default)}'"; GPN_PROBE_SO=${v:+tools/probes/_build/libgpn_$v.so} python /tmp/with_so.py tools/conv_bench.py 2>&1 | grep -v amdgpu; done | tee $O/wgrad_micro.txt
for i in 1 2 3; do for v in "" wgrad1; do echo "variant '${v:-ahead2}'"; GPN_PROBE_SO=${v:+tools/probes/_build/libgpn_$v.so} python /tmp/with_so.py bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), round(d['value'],1), d['roofline']['all_kernels']['spconv_wgrad_kernel'].get('frac_raw_events'))"; done; done 2>&1 | tee $O/bench_wgrad_ab.txt
python -m pytest tests -m gpu -q -x 2>&1 | tail -6 | tee $O/pytest_gpu.txt

cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05c
O=gpurun_out/r05c
python -m pytest tests/test_gpu_proposals.py tests/test_golden_pipeline.py tests/test_eval_ap.py -m gpu -q 2>&1 | tail -30 > $O/pytest.txt; tail -8 $O/pytest.txt
for i in 1 2; do timeout 300 python tools/eval_bench.py 2>$O/eval.err | tail -1 | tee $O/eval_$i.json; done

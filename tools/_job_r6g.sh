cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6g
O=gpurun_out/r6g
python -m pytest tests/test_gpu_model.py -m gpu -q -k "inference" 2>&1 | tail -15 > $O/tests_a.txt; tail -3 $O/tests_a.txt
python -m pytest tests -m gpu -q 2>&1 | tail -6 | tee $O/pytest_gpu.txt

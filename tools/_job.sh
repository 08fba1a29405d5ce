cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -2 | cut -c1-300
timeout 300 python tools/wgrad_bench.py 2>&1 | grep -E "^L| level" | cut -c1-110 > gpurun_out/wgrad_bench.txt; cat gpurun_out/wgrad_bench.txt | head -30

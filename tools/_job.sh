cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q -k "validation or nms or config2 or eval" 2>&1 | tail -2
python tools/eval_bench.py 2>&1 | grep -v amdgpu | tee gpurun_out/r04_bench_eval_batch4.json
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pe && timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pe -o t -- python $GRAFT_REPO_ROOT/tools/eval_bench.py --epochs 1 --steps 6 > /dev/null 2>&1
f=$(find /tmp/pe -name "*kernel_stats.csv" | head -1); cd $GRAFT_REPO_ROOT; python tools/gpu_categories.py "$f" 36 > gpurun_out/r04_eval_gpu_time_by_category.txt 2>&1; cp "$f" gpurun_out/r04_eval_kernel_stats.csv; cat gpurun_out/r04_eval_gpu_time_by_category.txt

cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_configs.py -m gpu -x -q -s -k training_mode 2>&1 | grep -E "worst|Error|assert|features" | head -10
python -m pytest tests/test_gpu_sync_free.py -x -q 2>&1 | tail -3
run() { env "$@" timeout 200 python bench.py --steps 60 --warmup 15 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', round(d['ms_per_step'],3))"; }
for r in 1 2 3 4; do run GPN_PROPOSALS_SYNC=1; run GPN_PROPOSALS_SYNC=0; done > gpurun_out/r04_sync_ab.txt 2>&1
python tools/sync_sites.py > gpurun_out/r04_sync_sites.txt 2>&1
python tools/host_wait.py > gpurun_out/r04_host_wait.txt 2>&1
cat gpurun_out/r04_sync_ab.txt; tail -8 gpurun_out/r04_sync_sites.txt; tail -15 gpurun_out/r04_host_wait.txt

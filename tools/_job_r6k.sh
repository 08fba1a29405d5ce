cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6k
O=gpurun_out/r6k
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $O/smoke.txt
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_driver_like.json; cut -c1-330 $O/bench_driver_like.json
GPN_DIST_SHARE_DEVICE=1 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-300 | tee $O/bench_two_ranks_shared.txt
python tools/soak.py 2>&1 | tail -12 | tee $O/soak.txt

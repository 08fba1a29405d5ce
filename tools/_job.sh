cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05d
O=gpurun_out/r05d
timeout 600 python tools/pth_loader_bench.py 2>$O/pth.err | tail -1 | tee $O/pth_loader.json
tail -3 $O/pth.err

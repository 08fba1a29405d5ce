cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6l
python tools/eval_cprofile.py 2>&1 | grep -v "^$" | cut -c1-170 > gpurun_out/r6l/eval_cprofile.txt
head -90 gpurun_out/r6l/eval_cprofile.txt

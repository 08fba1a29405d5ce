cd $GRAFT_REPO_ROOT
bash tools/final_measure_r04.sh

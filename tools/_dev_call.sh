R=$GRAFT_REPO_ROOT
cd $R
bash tools/ab_same_box.sh _ab/libgpn_hip_base.so _ab/libgpn_hip_new.so 4

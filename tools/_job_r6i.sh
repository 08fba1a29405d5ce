cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6i
GPN_PROBE_SO=tools/probes/_build/libgpn_walk.so python tools/conv_msplit_sweep.py --walk 2>&1 | grep -E "L2 25190 rows|L3 6915 rows|down 25190|up 25190" | sed 's/nt1\/sp4.*nt[0-9]\/sp9[^w]*//' | cut -c1-260 | tee gpurun_out/r6i/walk.txt

cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6h
O=gpurun_out/r6h
for v in "" ablB ablA ablAB; do echo "### variant '${v:-product}'"; GPN_PROBE_SO=${v:+tools/probes/_build/libgpn_$v.so} python tools/conv_msplit_sweep.py 2>&1 | grep -E "L2 25190 rows|L3 6915 rows|L4 1831 rows|L5 487 rows" | cut -c1-175; done | tee $O/ablation.txt

#!/bin/bash
# Round-6 measurement run (one box): GPU tests, the default bench line (stationary workload) four times, the moving workload of
# rounds 1-5 for comparison, other BASELINE configurations, rocprofv3 stats + PMC passes behind profiles/profile_summary.json /
# traffic.json (fingerprint of the kernel sources), queue breakdown / step sequence, the evaluation path (with its in-process
# BatchNorm A/B), the disk-fed loop, per-kernel rooflines, host-side tools.  Every step is bounded by `timeout`.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/final
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
timeout 900 python -m pytest "$R/tests" -m gpu -q > "$O/pytest_gpu.txt" 2>&1 < /dev/null; tail -2 "$O/pytest_gpu.txt"
rm -rf /tmp/p1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -o t -- python "$R/bench.py" --steps 16 --warmup 4 --no-cpu-baseline --no-validation > "$O/bench_under_rocprof.log" 2>&1 < /dev/null
f=$(find /tmp/p1 -name "*kernel_stats.csv" 2>/dev/null | head -1)
if [ -n "$f" ]; then cp "$f" "$O/kernel_stats.csv"; python "$R/tools/gpu_categories.py" "$f" 21 > "$O/gpu_time_by_category.txt" 2>&1 < /dev/null; (cd "$R" && python tools/profile_summary.py "$f" 21 "$O/profile_summary.json" > /dev/null 2>&1 < /dev/null); fi
f=$(find /tmp/p1 -name "*kernel_trace.csv" 2>/dev/null | head -1)
if [ -n "$f" ]; then python "$R/tools/step_sequence.py" "$f" > "$O/step_sequence.txt" 2>&1 < /dev/null; python "$R/tools/fwd_bwd_split.py" "$f" > "$O/fwd_bwd_split.txt" 2>&1 < /dev/null; python "$R/tools/summarize_trace.py" "$f" "$O/conv_kernels_by_grid.csv" "spconv,wgrad,bn_,ccl_,linear" > /dev/null 2>&1 < /dev/null; python "$R/tools/queue_breakdown.py" "$f" 21 > "$O/queue_breakdown.txt" 2>&1 < /dev/null; fi
cp "$O/profile_summary.json" "$R/profiles/profile_summary.json" 2>/dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/p_$c
  timeout 300 rocprofv3 --pmc $c --output-format csv -d /tmp/p_$c -o t -- python "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-validation > /dev/null 2>&1 < /dev/null
  f=$(find /tmp/p_$c -name "*counter_collection.csv" 2>/dev/null | head -1)
  if [ -n "$f" ]; then python "$R/tools/pmc_summary.py" "$f" spconv > "$O/pmc_$c.txt" 2>&1 < /dev/null; fi
done
f1=$O/pmc_FETCH_SIZE.txt; f2=$O/pmc_WRITE_SIZE.txt
if [ -s "$f1" ] && [ -s "$f2" ]; then (cd "$R" && python tools/pmc_traffic.py "$f1" "$f2" "$O/traffic.json" spconv_fwd,spconv_tiles,spconv_msplit > /dev/null 2>&1); cp "$O/traffic.json" "$R/profiles/traffic.json"; fi
rm -rf /tmp/p_sq
timeout 200 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d /tmp/p_sq -o t -- python "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-validation > /dev/null 2>&1 < /dev/null
f=$(find /tmp/p_sq -name "*counter_collection.csv" 2>/dev/null | head -1)
if [ -n "$f" ]; then python "$R/tools/pmc_summary.py" "$f" spconv > "$O/pmc_sq.txt" 2>&1 < /dev/null; fi
# the bench line: with profiles/profile_summary.json + traffic.json of THESE sources in place (profile_frac / traffic in the line)
timeout 300 python "$R/bench.py" > "$O/bench_default.json" 2> "$O/bench_default.err" < /dev/null; cut -c1-300 "$O/bench_default.json"
for i in 1 2 3; do timeout 240 python "$R/bench.py" --no-cpu-baseline > "$O/bench_default_run$i.json" 2> /dev/null < /dev/null; done
timeout 240 python "$R/bench.py" --steps 20 --warmup 5 --no-cpu-baseline > "$O/bench_steps20_warmup5.json" 2> /dev/null < /dev/null
for i in 1 2; do timeout 240 python "$R/bench.py" --moving --no-cpu-baseline > "$O/bench_moving_run$i.json" 2> /dev/null < /dev/null; done
timeout 240 python "$R/bench.py" --lib-knobs msplit=0 --no-cpu-baseline > "$O/bench_msplit_off.json" 2> /dev/null < /dev/null
GPN_BENCH_FORCE_GRAD_SYNC=1 timeout 240 python "$R/bench.py" --no-cpu-baseline > "$O/bench_forced_grad_sync.json" 2> /dev/null < /dev/null
for cfg in "--schedule 5,10" "--points 50000 --batch 4" "--batch 32" "--batch 4" "--batch 16"; do
  tag=$(echo "$cfg" | tr -d ' -' | tr ',' '_')
  timeout 200 python "$R/bench.py" $cfg --no-cpu-baseline > "$O/bench_$tag.json" 2> /dev/null < /dev/null
done
cd "$R"
for i in 1 2; do timeout 200 python tools/eval_bench.py 2>/dev/null | tail -1 > "$O/bench_eval_batch4_run$i.json"; done
timeout 300 python tools/eval_bench.py --ab-bn-fusion --epochs 5 2>/dev/null | tail -1 > "$O/eval_bn_ab.json"
timeout 600 python tools/pth_loader_bench.py 2> "$O/pth_loader.err" | tail -1 > "$O/pth_loader.json"
timeout 200 python tools/kernel_rooflines.py > "$O/kernel_rooflines.txt" 2>&1 < /dev/null
timeout 200 python tools/sync_sites.py > "$O/sync_sites.txt" 2>&1 < /dev/null
timeout 200 python tools/host_cprofile.py > "$O/host_cprofile.txt" 2>&1 < /dev/null
timeout 200 python tools/critical_path.py > "$O/critical_path.txt" 2>&1 < /dev/null
timeout 200 python tools/step_times.py 300 > "$O/step_times.txt" 2>&1 < /dev/null
cd /tmp; rm -rf /tmp/pe
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pe -o t -- python "$R/tools/eval_bench.py" --epochs 3 > /dev/null 2>&1 < /dev/null
f=$(find /tmp/pe -name "*kernel_stats.csv" 2>/dev/null | head -1)
if [ -n "$f" ]; then cp "$f" "$O/eval_kernel_stats.csv"; python "$R/tools/gpu_categories.py" "$f" 144 > "$O/eval_gpu_time_by_category.txt" 2>&1 < /dev/null; fi
cd "$R"; tail -2 "$O/pytest_gpu.txt"; cut -c1-400 "$O/bench_default.json"; head -5 "$O/gpu_time_by_category.txt"; cat "$O/eval_bn_ab.json" | cut -c1-200

#!/bin/bash
# First measurement of GPN_WGRAD_GROUP > 1 (consecutive same-shape layers in one weight-gradient contraction launch): the knob
# was unreadable through round 3 (profiles/r03_findings.md), so the grouped path has never run on a GPU.
#   gpurun --timeout 600 -- 'bash tools/wgrad_group_ab.sh [rounds]'
# 1. correctness of the grouped path (executor vs per-layer path, paired passes, golden training step) at GROUP=4;
# 2. launches per step at GROUP=4 (expect 66 contraction launches on the weight-gradient queue instead of 84);
# 3. interleaved bench runs at GROUP = 1 / 2 / 4 and GROUP=4 with the row threshold lifted.
R=${GRAFT_REPO_ROOT:-/root/repo}
N=${1:-4}
cd "$R"
GPN_WGRAD_GROUP=4 timeout 600 python -m pytest tests -m gpu -q -x -k "executor or pair or golden or train_step or determin or wgrad" 2>&1 < /dev/null | tail -2
export TMPDIR=/tmp
(cd /tmp && rm -rf /tmp/pg && GPN_WGRAD_GROUP=4 timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/pg -o t -- python "$R/bench.py" --steps 16 --warmup 4 --no-cpu-baseline > /dev/null 2>&1 < /dev/null)
f=$(find /tmp/pg -name "*kernel_trace.csv" 2>/dev/null | head -1)
if [ -n "$f" ]; then python tools/queue_breakdown.py "$f" 21 2>/dev/null | grep "^queue"; fi
run() { env "$@" timeout 200 python bench.py --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', round(d['ms_per_step'],3))"; }
for r in $(seq $N); do run GPN_WGRAD_GROUP=1; run GPN_WGRAD_GROUP=2; run GPN_WGRAD_GROUP=4; run GPN_WGRAD_GROUP=4 GPN_WGRAD_GROUP_ROWS=1000000000; done

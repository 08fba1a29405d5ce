#!/usr/bin/env python
"""Which kernels surround the synchronous host<->device copies of a step: from a rocprofv3 run with --kernel-trace
--hip-runtime-trace (csv), prints for every hipMemcpyWithStream / hipMemcpyAsync of the last steps the previous and the
next kernel launched by the same thread."""
import csv, glob, sys, collections
d = sys.argv[1]
api = list(csv.DictReader(open(glob.glob(d + '/**/*hip_api_trace.csv', recursive=True)[0])))
ker = {r['Correlation_Id']: r['Kernel_Name'] for r in csv.DictReader(open(glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0]))}
api.sort(key=lambda r: int(r['Start_Timestamp']))
by_thread = collections.defaultdict(list)
for r in api:
    by_thread[r['Thread_Id']].append(r)
ctx = collections.Counter()
for tid, rows in by_thread.items():
    last_k = None
    pend = []
    for r in rows:
        f = r['Function']
        if f in ('hipLaunchKernel', 'hipModuleLaunchKernel', 'hipExtModuleLaunchKernel', 'hipLaunchCooperativeKernel'):
            name = ker.get(r['Correlation_Id'], '?')[:60]
            for p in pend:
                ctx[(p[0], p[1], name)] += 1
            pend = []
            last_k = name
        elif f in ('hipMemcpyWithStream', 'hipMemcpyAsync', 'hipMemcpy'):
            pend.append((f, last_k))
n = sum(ctx.values())
print(n, "copies")
for (f, a, b), c in ctx.most_common(60):
    print(f"{c:5d}  {f:20s} after [{a}]  before [{b}]")

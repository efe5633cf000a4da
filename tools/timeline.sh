#!/bin/bash
# kernel trace of the default bench (or "$@"); prints the timeline of the LAST step: every kernel
# with its start offset, duration and the idle gap before it (run on the GPU box)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/timeline; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/trace -o t -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extras "$@" > $O/bench.log 2>&1 || { tail -5 $O/bench.log; exit 1; }
python3 - <<PY
import csv,glob
rows=[]
for f in glob.glob("$O/trace/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:90]))
for f in glob.glob("$O/trace/*memory_copy_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY "+r.get("Direction","")))
rows.sort()
# steps: split at the first kernel of a step (small_bucket or prep is not always there): use gaps > 150 us
steps=[[]]
for i,r in enumerate(rows):
    if i and r[0]-rows[i-1][1] > 150000: steps.append([])
    steps[-1].append(r)
# the last group may be the host-buffer call or the tail; print the last 2 groups that have >= 10 kernels
big=[s for s in steps if len(s)>=6]
for s in big[-2:]:
    t0=s[0][0]; prev=s[0][0]
    print("---- %d launches, %.1f us" % (len(s),(s[-1][1]-t0)/1e3))
    for a,b,n in s:
        print("%9.1f  dur %8.1f  gap %7.1f  %s" % ((a-t0)/1e3,(b-a)/1e3,(a-prev)/1e3,n))
        prev=max(prev,b)
PY
tail -1 $O/bench.log | cut -c1-400

#!/bin/bash
# kernel trace + stats of the default bench (run on the GPU box); top kernels to stdout
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/quick; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline "$@" > $O/bench.log 2>&1 || exit 1
python3 - <<PY
import csv,glob
f=glob.glob("$O/trace/*kernel_stats.csv")[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:12]:
    print("%-70s calls %5s avg %10.1f us  %5s%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
tail -1 $O/bench.log | cut -c1-330

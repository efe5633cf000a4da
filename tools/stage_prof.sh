#!/bin/bash
# per-kernel split of the device staging at config 2 and config 3 scale (on the GPU box)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/stage_prof; rm -rf $O; mkdir -p $O
for c in 2 3; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/c$c -o t -- python3 $R/tools/stage_bench.py --config $c --calls 5 > $O/c$c.log 2>&1 || { tail -3 $O/c$c.log; exit 1; }
  grep "reads/s" $O/c$c.log
  python3 - <<PY
import csv,glob
f=glob.glob("$O/c$c/*kernel_stats.csv")[0]
tot=0
for r in list(csv.DictReader(open(f)))[:14]:
    n=r["Name"].replace("umihip::(anonymous namespace)::","").replace("void ","").split("(")[0][:50]
    per=int(r["Calls"])/6.0; us=float(r["AverageNs"])/1e3; tot+=per*us
    print("  %-52s x%6.2f avg %8.1f us -> %8.1f us/call" % (n, per, us, per*us))
print("  sum %.1f us/call" % tot)
PY
  find $O/c$c -name '*kernel_trace.csv' -size +2M -delete
done

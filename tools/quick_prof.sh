#!/bin/bash
# kernel trace + stats of the default bench (run on the GPU box); per-step kernel times to stdout
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/quick; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python3 $R/bench.py --steps 5 --warmup 1 --no-extras "$@" > $O/bench.log 2>&1 || { tail -5 $O/bench.log; exit 1; }
python3 - <<PY
import csv,glob
f=glob.glob("$O/trace/*kernel_stats.csv")[0]
rows=list(csv.DictReader(open(f)))
tot=0
for r in rows[:24]:
    n=r["Name"].replace("umihip::(anonymous namespace)::","").replace("void ","").split("(")[0][:60]
    per=int(r["Calls"])/6.0
    us=float(r["AverageNs"])/1e3
    tot+=per*us
    print("%-62s x%5.2f avg %8.1f us  -> %7.1f us/step" % (n, per, us, per*us))
print("sum of the listed kernels per step: %.1f us" % tot)
PY
python3 -c "
import json;d=json.loads([l for l in open('$O/bench.log') if l.startswith('{')][-1]);print(d['ms_per_step'],d['phases_ms'],d['counters'])"

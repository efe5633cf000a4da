#!/bin/bash
# bench.py at several bucket sizes under option settings: bash tools/sweep_reads.sh "<reads...>" "<opt set>" ...
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
reads="$1"; shift
for n in $reads; do for o in "$@"; do
  args=""; for kv in $o; do [ "$kv" = "-" ] || args="$args --opt $kv"; done
  python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --reads $n $args > $R/gpurun_out/sweep.tmp 2>&1 || { tail -5 $R/gpurun_out/sweep.tmp; exit 1; }
  python3 - "$n $o" <<PY
import json,sys
d=json.loads(open("$R/gpurun_out/sweep.tmp").read().strip().splitlines()[-1])
print(sys.argv[1], "step %.3f ms" % d["ms_per_step"], {k: round(v,3) for k,v in d["phases_ms"].items()}, d["counters"]["pairs_evaluated"])
PY
done; done

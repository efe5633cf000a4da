#!/bin/bash
# bench.py under a list of ctx option settings (run on the GPU box): one line per setting
# usage: bash tools/sweep_opts.sh "bs_tab_chunk=1024" "bs_tab_chunk=2048" ...
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
for o in "$@"; do
  args=""; for kv in $o; do args="$args --opt $kv"; done
  python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline $args > $R/gpurun_out/sweep.tmp 2>&1 || { tail -5 $R/gpurun_out/sweep.tmp; exit 1; }
  python3 - "$o" <<PY
import json,sys
d=json.loads(open("$R/gpurun_out/sweep.tmp").read().strip().splitlines()[-1])
print(sys.argv[1], "step %.3f ms" % d["ms_per_step"], d.get("phases_ms"))
PY
done

#!/bin/bash
# one SQ counter pass (instruction counts) over the config-2 bench; prints the pair kernels' rows
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_insts; rm -rf $O; mkdir -p $O
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH --kernel-trace --output-format csv -d $O -o p -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $O/log 2>&1 || exit 1
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for r in csv.DictReader(open(glob.glob("$O/*counter_collection.csv")[0])):
    if any(k in r["Kernel_Name"] for k in ("bs_run","bs_tab","tab_scan")):
        acc[(r["Kernel_Name"][30:50], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k,v in sorted(acc.items()): print(k, "%.4g" % (sum(v)/len(v)))
PY

#!/bin/bash
# SQ counter passes over the default bench (or "$@"): per kernel averages of the counters below.
# usage (GPU box): bash tools/pmc_kernels.sh <kernel name substring> [bench args]
cd /tmp && export TMPDIR=/tmp
K=$1; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_k; rm -rf $O; mkdir -p $O
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -o p$i -- python3 $R/bench.py --steps 2 --warmup 1 --no-extras "$@" > $O/p$i.log 2>&1 || { tail -3 $O/p$i.log; exit 1; }
done
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob("$O/p*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "$K" in r["Kernel_Name"]:
            acc[(r["Kernel_Name"].split("(")[0][-40:], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k,v in sorted(acc.items()): print("%-42s %-24s %.4g  (n=%d)" % (k[0], k[1], sum(v)/len(v), len(v)))
PY

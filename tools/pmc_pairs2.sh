#!/bin/bash
# second set of SQ counter passes (instruction fetch, issue levels) over the config-2 bench
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/pmc_pu2
i=0
for set in "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC" "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_WAVES" "SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc_pu2/p$i -o p$i -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_pu2/p$i.log 2>&1 || exit 1
done
echo ok

#!/bin/bash
# kernel trace + SQ instruction counters of the many-small-positions step (bench.py --config 3)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_c3; rm -rf $O; mkdir -p $O
B="python3 $R/bench.py --config 3 --steps 5 --warmup 2 --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- $B > $O/bench_under_rocprof.log 2>&1 || { tail -3 $O/bench_under_rocprof.log; exit 1; }
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --kernel-trace --output-format csv -d $O/insts -o i -- $B > $O/insts.log 2>&1 || { tail -3 $O/insts.log; exit 1; }
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o f -- $B > $O/fetch.log 2>&1 || { tail -3 $O/fetch.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o w -- $B > $O/write.log 2>&1 || { tail -3 $O/write.log; exit 1; }
echo ok

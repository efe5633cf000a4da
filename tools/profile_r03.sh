#!/bin/bash
# rocprofv3 passes behind profiles/r03_counters.json, for one bench config (default 2):
#     bash tools/profile_r03.sh [config]          (on the GPU box; then, here: python3 tools/summarize_r03.py)
#   1. kernel trace + stats of `bench.py --config C --no-extras`    (durations, launches per step)
#   2. SQ instruction counters, one pass                            (VALU / SALU / LDS / branch / VMEM)
#   3. SQ cycle counters, one pass                                  (wave cycles, waits)
#   4. FETCH_SIZE and 5. WRITE_SIZE, a pass each                    (fabric-side bytes)
# Counters are collected with --kernel-trace only (no other trace domain beside --pmc).
C=${1:-2}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_r03/c$C; rm -rf $O; mkdir -p $O
B="python3 $R/bench.py --config $C --steps 5 --warmup 2 --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- $B > $O/bench_under_rocprof.log 2>&1 || { tail -3 $O/bench_under_rocprof.log; exit 1; }
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM --kernel-trace --output-format csv -d $O/insts -o i -- $B > $O/insts.log 2>&1 || { tail -3 $O/insts.log; exit 1; }
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES --kernel-trace --output-format csv -d $O/cycles -o c -- $B > $O/cycles.log 2>&1 || { tail -3 $O/cycles.log; exit 1; }
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o f -- $B > $O/fetch.log 2>&1 || { tail -3 $O/fetch.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o w -- $B > $O/write.log 2>&1 || { tail -3 $O/write.log; exit 1; }
python3 $R/bench.py --config $C --steps 20 --warmup 5 --no-extras > $O/bench_plain.log 2>&1 || exit 1
sha256sum $R/umi_collapse_rs_amd/csrc/* > $O/sources.sha256
# keep what travels back small: the per-dispatch traces are not needed once the stats exist
find $O -name '*kernel_trace.csv' -size +2M -delete
echo ok config $C

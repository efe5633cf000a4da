#!/bin/bash
# rocprofv3 passes behind the numbers in DESIGN.md / bench.py (run on the GPU box):
#   kernel trace + stats of the default bench, FETCH_SIZE and WRITE_SIZE in separate PMC passes,
#   and one plain bench run.  Results under gpurun_out/prof_config2/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_config2; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_under_rocprof.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o f -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o w -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/write.log 2>&1 || exit 1
python3 $R/bench.py > $O/bench_plain.log 2>&1 || exit 1
echo ok

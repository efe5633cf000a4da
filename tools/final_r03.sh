#!/bin/bash
# end-of-round bench lines on the GPU box (after tools/profile_r03.sh + summarize_r03.py, so that the
# lines carry the counters of these very sources): gpurun_out/final_r03/*.json, collected into
# profiles/ by `python3 tools/summarize_r03.py --collect-final`
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final_r03; rm -rf $O; mkdir -p $O
cd $R
python3 bench.py > $O/bench_line_default.log 2>&1 || { tail -3 $O/bench_line_default.log; exit 1; }
for c in 3 4 5 2m wide24; do
  python3 bench.py --config $c --steps 20 --warmup 3 --no-extras > $O/bench_line_config$c.log 2>&1 || { tail -3 $O/bench_line_config$c.log; exit 1; }
done
python3 tools/e2e_probe.py > $O/end_to_end_split.txt 2>&1
timeout -k 10 600 python3 -m pytest tests/test_gpu_fullscale.py -m gpu -x -q -s > $O/fullscale.txt 2>&1
for f in $O/bench_line_*.log; do grep -h -o "\"ms_per_step\": [0-9.]*" $f | head -1 | sed "s|^|$(basename $f) |"; done
tail -3 $O/fullscale.txt

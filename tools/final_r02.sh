#!/bin/bash
# End-of-round measurements (GPU box): the other configs' bench lines, then the rocprofv3 passes of config 2
# (tools/profile_r02.sh).  Outputs under gpurun_out/final/.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final; mkdir -p $O
cd $R
for c in 2m 3 4 5; do
  python3 bench.py --config $c --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_config$c.log 2>&1 || { tail -3 $O/bench_config$c.log; exit 1; }
  echo "config $c done"
done
python3 bench.py > $O/bench_config2.log 2>&1 || { tail -3 $O/bench_config2.log; exit 1; }
echo "config 2 done"
bash tools/profile_r02.sh || exit 1
bash tools/profile_config3.sh || exit 1
# the N > 1 launch rehearsed with two ranks on this one GPU (gloo; the RCCL run is the driver's)
cd $R && BENCH_BACKEND=gloo timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 10 --warmup 2 > $O/bench_2ranks_gloo.log 2>&1 || { tail -5 $O/bench_2ranks_gloo.log; exit 1; }
echo "2 ranks (gloo) done"

#!/bin/bash
# round 3, call 34: full GPU suite on the final tree + the train lines
mkdir -p gpurun_out/prof3; cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -x -q -m gpu > gpurun_out/r3c34_pytest.log 2>&1; tail -3 gpurun_out/r3c34_pytest.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof3
timeout 900 python $R/bench.py --config train --steps 20 --warmup 5 > $OUT/bench_train.json 2> $OUT/bench_train.err
SIPMASK_FORCE_DIST=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29519 $R/bench.py --gpus 1 --config train --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_train_rccl1.json 2> $OUT/bench_train_rccl1.err
timeout -k 5 600 rocprofv3 --kernel-trace -d $OUT/train -o train -- python $R/bench.py --config train --steps 10 --warmup 3 --no-cpu-baseline > $OUT/train.log 2>&1
python $R/tools/prof_stats.py $OUT/train $OUT/kernel_stats_train_step.csv 8
rm -rf $OUT/train
cut -c1-200 $OUT/bench_train.json; tail -1 $OUT/bench_train_rccl1.json | cut -c1-200

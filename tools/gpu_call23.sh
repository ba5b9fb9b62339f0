#!/bin/bash
# the collective path on one GPU: one rank under torch.distributed.run, RCCL process group of size 1, SIPMASK_FORCE_DIST=1
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/call23
rm -rf $OUT; mkdir -p $OUT
cd $R
export SIPMASK_FORCE_DIST=1
L="python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517"
NCCL_DEBUG=VERSION timeout 300 $L bench.py --gpus 1 --config train --steps 3 --warmup 1 > $OUT/train_rccl1.json 2>$OUT/train_rccl1.err
tail -1 $OUT/train_rccl1.json | cut -c1-900; grep -i "rccl\|nccl version" $OUT/train_rccl1.err | head -3
timeout 300 $L bench.py --gpus 1 --no-cpu-baseline --steps 20 --warmup 5 > $OUT/r50_rccl1.json 2>$OUT/r50_rccl1.err
tail -1 $OUT/r50_rccl1.json | cut -c1-300
timeout 300 $L bench.py --gpus 1 --config vis > $OUT/vis_rccl1.json 2>$OUT/vis_rccl1.err
tail -1 $OUT/vis_rccl1.json | cut -c1-300
tail -3 $OUT/*.err | cut -c1-300

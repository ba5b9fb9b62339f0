#!/bin/bash
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
SIPMASK_FORCE_DIST=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 $R/bench.py --gpus 1 --steps 30 --warmup 5 --no-cpu-baseline 2>/tmp/err.txt | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('torchrun + RCCL barrier, default config:', d['value'], d['ms_per_step'], d['n_gpus'], d['config']['steps_in_flight'], d['config']['detections_per_image'])"
tail -2 /tmp/err.txt | cut -c1-200

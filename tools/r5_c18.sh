# the driver's multi-rank launch shape on one GPU: bench.py under torch.distributed.run, collective path forced on
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
SECONDS=0
SIPMASK_FORCE_DIST=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5c18_dist.json 2> gpurun_out/r5c18_dist.err; echo "dist rc $? wall ${SECONDS}s"
tail -n 3 gpurun_out/r5c18_dist.err | cut -c1-300
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r5c18_dist.json').read().strip().splitlines() if l.startswith('{')][-1])
print(d['value'], d['n_gpus'], d['steady_state']['value'], d['with_results']['value'], d.get('parity_pairs') is not None, list(d.get('other_configs',{}).keys()))
PY
SECONDS=0
timeout 600 python bench.py --config train --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r5c18_train.json 2> gpurun_out/r5c18_train.err; echo "train rc $? wall ${SECONDS}s"; cut -c1-200 gpurun_out/r5c18_train.json
timeout 600 python bench.py --batch 1 --in-flight 1 --steps 30 --warmup 5 --no-cpu-baseline --no-extras | cut -c1-250
timeout 600 python bench.py --batch 2 --steps 30 --warmup 5 --no-cpu-baseline --no-extras | cut -c1-250

set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for i in 1 2 3; do timeout 300 python bench.py --config eval_shapes 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('eval_shapes K=64:', d['value'], d['ms_per_step'], d['config']['host_ms_per_batch_plan_lookup'], [v['build_s'] for v in d['config']['first_touch'].values()])"; done
for i in 1 2; do timeout 300 python bench.py --config eval_shapes --steps 512 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('eval_shapes K=512:', d['value'], d['ms_per_step'])"; done
nproc; python -c "import os; print(os.cpu_count(), os.sched_getaffinity(0).__len__())"

#!/bin/bash
# round 3, GPU call 6: geometry-hoisted deformable f32 loader, VIS forward_train, every BASELINE config's bench line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_f32_plan.py tests/test_gpu_x3.py tests/test_gpu_vis.py tests/test_gpu_kernels.py -m gpu -q --maxfail=10 > gpurun_out/r3c6_pytest.log 2>&1
echo "pytest rc $?" >> gpurun_out/r3c6_pytest.log
tail -15 gpurun_out/r3c6_pytest.log
show() { python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r3c6_bench_$1.json").read().strip().splitlines()[-1])
    r=d.get("roofline") or {}
    print("$1", d["value"], d["unit"], d["ms_per_step"], "roofline", r.get("achieved"), r.get("frac"), r.get("ms_per_launch"), "cpu", (d.get("cpu_baseline") or {}).get("value"))
except Exception as e:
    print("$1 failed", e); print(open("gpurun_out/r3c6_bench_$1.err").read()[-1500:])
PY
}
timeout 600 python bench.py --steps 20 --warmup 5 --precision head_x3 --breakdown gpurun_out/r3c6_breakdown_x3.txt > gpurun_out/r3c6_bench_x3.json 2> gpurun_out/r3c6_bench_x3.err; show x3
timeout 600 python bench.py --config r101 > gpurun_out/r3c6_bench_r101.json 2> gpurun_out/r3c6_bench_r101.err; show r101
timeout 900 python bench.py --config train > gpurun_out/r3c6_bench_train.json 2> gpurun_out/r3c6_bench_train.err; show train
timeout 600 python bench.py --config vis > gpurun_out/r3c6_bench_vis.json 2> gpurun_out/r3c6_bench_vis.err; show vis
timeout 600 python bench.py --config vis --no-graph --no-cpu-baseline > gpurun_out/r3c6_bench_vis_eager.json 2> gpurun_out/r3c6_bench_vis_eager.err; show vis_eager
SIPMASK_FORCE_DIST=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --config train --no-cpu-baseline > gpurun_out/r3c6_bench_train_rccl1.json 2> gpurun_out/r3c6_bench_train_rccl1.err; show train_rccl1

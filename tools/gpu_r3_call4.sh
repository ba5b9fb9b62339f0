#!/bin/bash
# round 3, GPU call 4: full suite (per-image metas, x3 tolerances, bucket test), x3 launch-structure A/B
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --maxfail=25 --deselect tests/test_gpu_baseline_shape.py > gpurun_out/r3c4_pytest.log 2>&1
echo "pytest rc $?" >> gpurun_out/r3c4_pytest.log
tail -40 gpurun_out/r3c4_pytest.log
for L in 1 0; do
  timeout 600 python bench.py --steps 20 --warmup 5 --precision head_x3 --lanes $L --no-cpu-baseline > gpurun_out/r3c4_bench_x3_lanes$L.json 2> gpurun_out/r3c4_bench_x3_lanes$L.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r3c4_bench_x3_lanes$L.json").read().strip().splitlines()[-1])
print("x3 lanes=$L", d["value"], d["ms_per_step"], d["config"]["launch"])
PY
done
timeout 600 python bench.py --steps 20 --warmup 5 --precision head_x3 --batch 8 --no-cpu-baseline > gpurun_out/r3c4_bench_x3_b8.json 2> gpurun_out/r3c4_bench_x3_b8.err
tail -c 400 gpurun_out/r3c4_bench_x3_b8.json | head -c 400; python - <<PY
import json
d=json.loads(open("gpurun_out/r3c4_bench_x3_b8.json").read().strip().splitlines()[-1])
print("x3 batch 8", d["value"], d["ms_per_step"], d["config"]["launch"])
PY

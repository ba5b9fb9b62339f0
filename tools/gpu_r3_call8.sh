#!/bin/bash
# round 3, GPU call 8: wave-parallel clip tracker, fused split producers, MFMA clock probe
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_vis.py tests/test_gpu_x3.py tests/test_gpu_baseline_shape.py::test_head_x3_plan_at_baseline_shape -m gpu -q --maxfail=10 > gpurun_out/r3c8_pytest.log 2>&1
echo "pytest rc $?" >> gpurun_out/r3c8_pytest.log
tail -15 gpurun_out/r3c8_pytest.log
show() { python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r3c8_bench_$1.json").read().strip().splitlines()[-1])
    r=d.get("roofline") or {}
    print("$1", d["value"], d["unit"], d["ms_per_step"], "roofline", r.get("achieved"), r.get("frac"), r.get("ms_per_launch"))
except Exception as e:
    print("$1 failed", e); print(open("gpurun_out/r3c8_bench_$1.err").read()[-1500:])
PY
}
timeout 600 python bench.py --config vis --no-cpu-baseline > gpurun_out/r3c8_bench_vis.json 2> gpurun_out/r3c8_bench_vis.err; show vis
timeout 600 python bench.py --precision head_x3 --no-cpu-baseline --breakdown gpurun_out/r3c8_breakdown_x3.txt > gpurun_out/r3c8_bench_x3.json 2> gpurun_out/r3c8_bench_x3.err; show x3
( for a in "8 1 4000 0" "8 1 4000 1" "8 1 40000 0" "8 1 40000 1" "4 1 40000 0" "4 2 40000 0" "8 1 400000 0"; do tools/bin/mfma_peak $a | tail -1; done ) > gpurun_out/r3c8_mfma_peak.txt 2>&1
cat gpurun_out/r3c8_mfma_peak.txt | cut -c1-400

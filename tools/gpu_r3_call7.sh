#!/bin/bash
# round 3, GPU call 7: device clip tracker, deformable-kernel auto choice, full suite, VIS / bf16 bench lines
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --maxfail=25 > gpurun_out/r3c7_pytest.log 2>&1
echo "pytest rc $?" >> gpurun_out/r3c7_pytest.log
tail -30 gpurun_out/r3c7_pytest.log
show() { python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r3c7_bench_$1.json").read().strip().splitlines()[-1])
    r=d.get("roofline") or {}
    print("$1", d["value"], d["unit"], d["ms_per_step"], "roofline", r.get("achieved"), r.get("frac"), r.get("ms_per_launch"), d["config"].get("deform_kernel"))
except Exception as e:
    print("$1 failed", e); print(open("gpurun_out/r3c7_bench_$1.err").read()[-1500:])
PY
}
timeout 600 python bench.py --config vis --no-cpu-baseline > gpurun_out/r3c7_bench_vis.json 2> gpurun_out/r3c7_bench_vis.err; show vis
timeout 600 python bench.py --no-cpu-baseline --breakdown gpurun_out/r3c7_breakdown.txt > gpurun_out/r3c7_bench_bf16.json 2> gpurun_out/r3c7_bench_bf16.err; show bf16
SIPMASK_DEFORM_GATHER=0 timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r3c7_bench_bf16_window.json 2> gpurun_out/r3c7_bench_bf16_window.err; show bf16_window

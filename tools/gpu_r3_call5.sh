#!/bin/bash
# round 3, GPU call 5: x3 FeatureAlign on the split-precision f32-input kernel, x3 parity at the BASELINE shape, launch A/Bs
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_x3.py tests/test_gpu_baseline_shape.py "tests/test_gpu_api.py::test_gradients_living_in_allreduce_buckets_match_plain_training" -m gpu -q --maxfail=10 > gpurun_out/r3c5_pytest.log 2>&1
echo "pytest rc $?" >> gpurun_out/r3c5_pytest.log
tail -25 gpurun_out/r3c5_pytest.log
run() { # name, env, args
  env $2 timeout 600 python bench.py --steps 20 --warmup 5 --precision head_x3 --no-cpu-baseline $3 > gpurun_out/r3c5_bench_$1.json 2> gpurun_out/r3c5_bench_$1.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r3c5_bench_$1.json").read().strip().splitlines()[-1])
    print("$1", d["value"], d["ms_per_step"], d["config"]["launch"])
except Exception as e:
    print("$1 failed", e); print(open("gpurun_out/r3c5_bench_$1.err").read()[-800:])
PY
}
run x3_default "A=1" "--breakdown gpurun_out/r3c5_breakdown_x3.txt"
run x3_fa_f32 "SIPMASK_X3_FEAT_ALIGN=f32" ""
run x3_subgraphs1 "A=1" "--sub-graphs 1"
run x3_subgraphs2 "A=1" "--sub-graphs 2"
timeout 600 python tools/parity_baseline.py --plan subbatch --precision head_x3 --out gpurun_out/r3c5_parity_subbatch_x3.json > gpurun_out/r3c5_parity_x3.log 2>&1
grep -n "features  mask_logits\|features parity\|^parity" gpurun_out/r3c5_parity_x3.log

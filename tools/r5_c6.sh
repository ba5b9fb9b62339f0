set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_kernels.py tests/test_gpu_api.py tests/test_gpu_baseline_shape.py -x -q -m gpu > gpurun_out/r5c6_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r5c6_pytest.log
tail -n 12 gpurun_out/r5c6_pytest.log
timeout 600 python bench.py --config eval_shapes > gpurun_out/r5c6_eval.json 2> gpurun_out/r5c6_eval.err; echo "eval rc $?"
tail -n 3 gpurun_out/r5c6_eval.err; cut -c1-3000 gpurun_out/r5c6_eval.json
timeout 600 python bench.py --no-cpu-baseline --extras-budget 60 > gpurun_out/r5c6_bench.json 2> gpurun_out/r5c6_bench.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r5c6_bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "steady", d.get("steady_state",{}).get("value"))
print("post", json.dumps(d.get("post_processing"))[:900])
print("worst", json.dumps(d.get("mask_assemble_worst_case"))[:300])
print("with_results", d.get("with_results",{}).get("value"))
PY

set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r5c5_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r5c5_pytest.log
tail -n 15 gpurun_out/r5c5_pytest.log
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r5c5_bench.json 2> gpurun_out/r5c5_bench.err; echo "bench rc $?"
tail -n 5 gpurun_out/r5c5_bench.err
python - <<'PY'
import json
try:
    d=json.loads(open("gpurun_out/r5c5_bench.json").read().strip().splitlines()[-1])
    print("value", d["value"], "ms", d["ms_per_step"], "steady", d.get("steady_state",{}).get("value"))
    print("with_results", json.dumps(d.get("with_results"))[:900])
    print("post", json.dumps(d.get("post_processing"))[:1200])
    print("pairs", json.dumps(d.get("parity_pairs"))[:900])
    pp=d.get("parity_plan",{}); print("parity_plan", pp.get("value"), pp.get("ms_per_step"), pp.get("mask_logit_max_abs"), pp.get("same_order"))
    print("other", json.dumps(d.get("other_configs"))[:2500])
    print("worst", json.dumps(d.get("mask_assemble_worst_case"))[:300])
except Exception as e:
    print("parse failed", e)
PY
timeout 600 python bench.py --config eval_shapes > gpurun_out/r5c5_eval.json 2> gpurun_out/r5c5_eval.err; echo "eval rc $?"
tail -n 3 gpurun_out/r5c5_eval.err; cut -c1-2500 gpurun_out/r5c5_eval.json

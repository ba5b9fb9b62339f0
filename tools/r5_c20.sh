set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_x3.py tests/test_gpu_baseline_shape.py -q -m gpu > gpurun_out/r5c20_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r5c20_pytest.log
grep -v "^  File\|^$" gpurun_out/r5c20_pytest.log | tail -n 8
timeout 300 python bench.py --precision head_x3 --no-cpu-baseline --extras-budget 20 --breakdown gpurun_out/r5c20_x3_breakdown.txt > gpurun_out/r5c20_x3.json 2> gpurun_out/r5c20_x3.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5c20_x3.json').read().strip().splitlines()[-1])
print(d['value'], d['steady_state']['value'], d['roofline']['frac'], d['roofline']['gflop_per_launch'], d['roofline']['ms_per_launch'])
PY
timeout 600 python tools/parity_baseline.py --plan pipelined --precision head_x3 --out gpurun_out/r5c20_parity_x3.json > gpurun_out/r5c20_parity_x3.log 2>&1; tail -n 5 gpurun_out/r5c20_parity_x3.log | cut -c1-300

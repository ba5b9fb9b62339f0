set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_deform_patch.py tests/test_gpu_engine.py tests/test_gpu_kernels.py -x -q -m gpu > gpurun_out/r5c9_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r5c9_pytest.log
grep -v "^  File\|^$" gpurun_out/r5c9_pytest.log | tail -n 6
timeout 300 python tools/deform_fwd_bench.py 4 0.0,1.0 2>&1 | grep "B=4" | head -4
timeout 600 python bench.py --no-cpu-baseline --extras-budget 30 --breakdown gpurun_out/r5c9_bd.txt > gpurun_out/r5c9_bench.json 2> gpurun_out/r5c9_bench.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r5c9_bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "steady", d.get("steady_state",{}).get("value"))
PY
grep -n "feat_align\|mask_assemble\|det_boxes\|# sum" gpurun_out/r5c9_bd.txt

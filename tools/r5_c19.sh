# x3 plan: first tower launch on two terms ([hi | hi] of the bf16 FPN outputs)
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_x3.py tests/test_gpu_baseline_shape.py -x -q -m gpu > gpurun_out/r5c19_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r5c19_pytest.log
grep -v "^  File\|^$" gpurun_out/r5c19_pytest.log | tail -n 8
for pass in 1 2; do
  for two in True False; do
    timeout 300 python tools/bench_with.py _X3_TOWER0_TWO_TERMS=$two -- --precision head_x3 --steps 200 --warmup 20 --no-cpu-baseline --no-extras > gpurun_out/r5c19_two${two}_$pass.json 2> gpurun_out/r5c19_two${two}_$pass.err
    echo "two_terms=$two pass $pass: $(python -c "import json,sys; d=json.loads(open('gpurun_out/r5c19_two${two}_$pass.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['ms_per_launch'], d['roofline']['frac'])")"
  done
done
timeout 300 python bench.py --precision head_x3 --no-cpu-baseline --extras-budget 20 --breakdown gpurun_out/r5c19_x3_breakdown.txt > gpurun_out/r5c19_x3.json 2> gpurun_out/r5c19_x3.err
grep 'tower\|split:pyr' gpurun_out/r5c19_x3_breakdown.txt

# EXPERIMENT: persistent blocks in the patch kernel (one block per CU walking its tiles) vs one block per tile
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_patch_conv.py -x -q -m gpu > gpurun_out/r5c22_pytest0.log 2>&1; echo "pytest(default grid) rc $?"; tail -n 2 gpurun_out/r5c22_pytest0.log
SIPMASK_EXP_PATCH_PERSIST=256 timeout 600 python -m pytest tests/test_gpu_patch_conv.py -x -q -m gpu > gpurun_out/r5c22_pytest1.log 2>&1; echo "pytest(persist) rc $?"; tail -n 2 gpurun_out/r5c22_pytest1.log
for pass in 1 2; do
  for p in 0 256 512; do
    echo "persist=$p: $(SIPMASK_EXP_PATCH_PERSIST=$p timeout 300 python bench.py --tower-only 50 2>/dev/null | cut -c1-200)"
  done
done
for pass in 1 2; do
  for p in 0 256; do
    SIPMASK_EXP_PATCH_PERSIST=$p timeout 300 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-extras > gpurun_out/r5c22_p${p}_$pass.json 2> gpurun_out/r5c22_p${p}_$pass.err
    echo "persist=$p pass $pass: $(python -c "import json,sys; d=json.loads(open('gpurun_out/r5c22_p${p}_$pass.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'])")"
  done
done

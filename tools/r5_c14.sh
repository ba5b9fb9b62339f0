# experiments build: mask assembly with tile-row work units (+ batched stage-1 loads, two-row stage 3) against the committed kernel
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_engine.py tests/test_gpu_api.py -x -q -m gpu -k "mask or mixed_size or keep_ratio or rle" > gpurun_out/r5c14_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r5c14_pytest.log
grep -v "^  File\|^$" gpurun_out/r5c14_pytest.log | tail -n 6
for rep in 1 2; do
echo "old kernel"; timeout 200 python tools/mask_bench.py --lib sipmask_amd/libsipmask_hip_old.so 2>&1 | grep -v amdgpu.ids
echo "new, 4 blocks per CU"; SIPMASK_EXP_MASK_OCC=4 timeout 200 python tools/mask_bench.py 2>&1 | grep -v amdgpu.ids
echo "new, 5 blocks per CU"; SIPMASK_EXP_MASK_OCC=5 timeout 200 python tools/mask_bench.py 2>&1 | grep -v amdgpu.ids
done
for pass in 1 2; do
  for o in 4 5; do
    SIPMASK_EXP_MASK_OCC=$o timeout 300 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-extras > gpurun_out/r5c14_o${o}_$pass.json 2> gpurun_out/r5c14_o${o}_$pass.err
    echo "occ $o pass $pass: $(python -c "import json,sys; d=json.loads(open('gpurun_out/r5c14_o${o}_$pass.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")"
  done
done

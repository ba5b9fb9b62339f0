# experiments build: mask assembly with batched stage-1 loads + two-row stage 3; 256 vs 512 threads per tile
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_engine.py tests/test_gpu_api.py -x -q -m gpu -k "mask or mixed_size or keep_ratio or rle" > gpurun_out/r5c13_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r5c13_pytest.log
grep -v "^  File\|^$" gpurun_out/r5c13_pytest.log | tail -n 6
for t in 256 512 256 512; do echo "threads $t"; SIPMASK_EXP_MASK_THREADS=$t timeout 200 python tools/mask_bench.py 2>&1 | grep -v amdgpu.ids; done
for pass in 1 2; do
  for t in 256 512; do
    SIPMASK_EXP_MASK_THREADS=$t timeout 300 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-extras > gpurun_out/r5c13_t${t}_$pass.json 2> gpurun_out/r5c13_t${t}_$pass.err
    echo "threads $t pass $pass: $(python -c "import json,sys; d=json.loads(open('gpurun_out/r5c13_t${t}_$pass.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")"
  done
done

# experiments build (make -C sipmask_amd/csrc EXPERIMENTS=1): the operand ring of conv_dma32_kernel, kernel level and step level
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "ring or loader_variants or igemm_vs_torch" > gpurun_out/r5c12_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r5c12_pytest.log
grep -v "^  File\|^$" gpurun_out/r5c12_pytest.log | tail -n 8
V="0,0x04000000,0x04080000,0x04000000:r3,0x04000000:r4,0x04000800:r4,0:r4,0:r3"
timeout 400 python tools/conv_bench.py --only "l3.conv1,l3.conv3,l4.conv3,l2.conv1,l1.conv1,mask_lat0,l2.conv3,l1.conv3" --variants "$V" > gpurun_out/r5c12_conv_ring.txt 2>&1
cat gpurun_out/r5c12_conv_ring.txt
for pass in 1 2; do
  for r in 2 4 3; do
    SIPMASK_EXP_K32_RING=$r timeout 300 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-extras > gpurun_out/r5c12_ring${r}_$pass.json 2> gpurun_out/r5c12_ring${r}_$pass.err
    echo "ring $r pass $pass: $(python -c "import json,sys; d=json.loads(open('gpurun_out/r5c12_ring${r}_$pass.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")"
  done
done
for r in 2 4; do
  SIPMASK_EXP_K32_RING=$r timeout 300 python bench.py --config r101 --steps 200 --warmup 20 --no-cpu-baseline --no-extras > gpurun_out/r5c12_r101_ring$r.json 2> gpurun_out/r5c12_r101_ring$r.err
  echo "r101 ring $r: $(python -c "import json,sys; d=json.loads(open('gpurun_out/r5c12_r101_ring$r.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")"
done

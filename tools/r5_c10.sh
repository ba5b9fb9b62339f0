set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "deform" > gpurun_out/r5c10_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r5c10_pytest.log
grep -v "^  File\|^$" gpurun_out/r5c10_pytest.log | tail -n 8
timeout 300 python tools/deform_bwd_bench.py 2>&1 | tail -n 12
timeout 600 python bench.py --config train --no-cpu-baseline > gpurun_out/r5c10_train.json 2> gpurun_out/r5c10_train.err; echo "train rc $?"
cut -c1-300 gpurun_out/r5c10_train.json

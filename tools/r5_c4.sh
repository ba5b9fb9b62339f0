set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_deform_x3.py -x -q -m gpu > gpurun_out/r5c4_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r5c4_pytest.log
tail -n 12 gpurun_out/r5c4_pytest.log
timeout 300 python tools/deform_fwd_bench.py 4 0.0,1.0,2.0,4.0 > gpurun_out/r5c4_deform.txt 2>&1
grep x3 gpurun_out/r5c4_deform.txt
bash tools/pmc_deform_x3.sh 2>&1 | grep x3_window

set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_deform_x3.py tests/test_gpu_x3.py -x -q -m gpu > gpurun_out/r5c1_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r5c1_pytest.log
tail -n 15 gpurun_out/r5c1_pytest.log
timeout 300 python tools/deform_fwd_bench.py 4 0.0,1.0,4.0 > gpurun_out/r5c1_deform.txt 2>&1
tail -n 8 gpurun_out/r5c1_deform.txt
timeout 600 python bench.py --precision head_x3 --no-extras --no-cpu-baseline --breakdown gpurun_out/r5c1_bd_x3.txt > gpurun_out/r5c1_bench_x3.json 2> gpurun_out/r5c1_bench_x3.err
cut -c1-600 gpurun_out/r5c1_bench_x3.json
grep -n "feat_align\|sip_mask_lat\|reg_ctr\|sum2\|# sum" gpurun_out/r5c1_bd_x3.txt

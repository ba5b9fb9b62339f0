cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_deform_patch.py tests/test_gpu_kernels.py -x -q -k deform 2>&1 | tail -2
timeout 200 python tools/deform_fwd_bench.py 2>&1 | grep -v "^$" | tail -10

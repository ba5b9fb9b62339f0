cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_deform_patch.py -x -q 2>&1 | tail -15
timeout 200 python tools/deform_fwd_bench.py 2>&1 | grep -v "^$" | tail -12

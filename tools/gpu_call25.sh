cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_patch_conv.py -x -q 2>&1 | tail -3
timeout 200 python tools/patch_bench.py 2>&1 | grep -v "^$" | cut -c1-400
timeout 200 python bench.py 2>&1 | tail -1 | cut -c1-600

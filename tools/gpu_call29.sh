cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_deform_patch.py tests/test_gpu_kernels.py tests/test_gpu_engine.py -x -q 2>&1 | tail -5
timeout 200 python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c1-330
timeout 200 python bench.py --no-cpu-baseline --config r101 2>&1 | tail -1 | cut -c1-200

#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_api.py::test_gradients_living_in_allreduce_buckets_match_plain_training tests/test_gpu_deform_patch.py tests/test_gpu_kernels.py -m gpu -q -k "bucket or deform_patch or nms or mask_assemble" > gpurun_out/r3c2_pytest.log 2>&1
echo "pytest rc $?" >> gpurun_out/r3c2_pytest.log
grep -n "RuntimeError\|passed\|failed\|^E  " gpurun_out/r3c2_pytest.log | head -40

#!/bin/bash
# round 3, call 10: transposing GN-statistics reduce in the patch / deform-window epilogues -- bit equality against the previous
# build, the failing determinism test, bench lines
mkdir -p gpurun_out; cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_engine.py tests/test_gpu_patch_conv.py tests/test_gpu_deform_patch.py tests/test_gpu_x3.py -x -q -m gpu > gpurun_out/r3c10_pytest.log 2>&1; tail -3 gpurun_out/r3c10_pytest.log
timeout 300 python bench.py --breakdown gpurun_out/r3c10_breakdown.txt > gpurun_out/r3c10_bench.json 2> gpurun_out/r3c10_bench.err; cat gpurun_out/r3c10_bench.json | cut -c1-400
timeout 300 python bench.py --precision head_x3 > gpurun_out/r3c10_bench_x3.json 2>> gpurun_out/r3c10_bench.err; cut -c1-200 gpurun_out/r3c10_bench_x3.json
grep -E "tower|reg_convs.3|feat_align|sum" gpurun_out/r3c10_breakdown.txt
timeout 300 python tools/hash_outputs.py > gpurun_out/r3c10_hash_new.txt 2> gpurun_out/r3c10_hash.err
cp sipmask_amd/libsipmask_hip_prev.so sipmask_amd/libsipmask_hip.so
timeout 300 python tools/hash_outputs.py > gpurun_out/r3c10_hash_old.txt 2>> gpurun_out/r3c10_hash.err
diff gpurun_out/r3c10_hash_new.txt gpurun_out/r3c10_hash_old.txt && echo "BIT-IDENTICAL to the previous build ($(wc -l < gpurun_out/r3c10_hash_new.txt) digests)"
tail -3 gpurun_out/r3c10_hash.err

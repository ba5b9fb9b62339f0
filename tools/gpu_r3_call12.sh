#!/bin/bash
# round 3, call 12: full GPU suite on the transposing GN reduce (explicit fma for the sum of squares), uniform vs mixed patch
# launches inside sub-plans (A/B, interleaved), bench lines
mkdir -p gpurun_out; cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -x -q -m gpu > gpurun_out/r3c12_pytest.log 2>&1; tail -3 gpurun_out/r3c12_pytest.log
for i in 1 2 3; do
  timeout 200 python bench.py --steps 40 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('uniform', d['value'], d['roofline']['ms_per_launch'], d['roofline']['frac'])" | tee -a gpurun_out/r3c12_ab.txt
  SIPMASK_PATCH_MIXED=1 timeout 200 python bench.py --steps 40 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mixed  ', d['value'], d['roofline']['ms_per_launch'], d['roofline']['frac'])" | tee -a gpurun_out/r3c12_ab.txt
done
timeout 300 python bench.py --precision head_x3 --steps 40 2>/dev/null | cut -c1-300 | tee gpurun_out/r3c12_x3.json
SIPMASK_PATCH_MIXED=1 timeout 300 python bench.py --precision head_x3 --steps 40 2>/dev/null | cut -c1-300 | tee gpurun_out/r3c12_x3_mixed.json
timeout 300 python tools/hash_outputs.py > gpurun_out/r3c12_hash_a.txt 2> gpurun_out/r3c12_hash.err
timeout 300 python tools/hash_outputs.py > gpurun_out/r3c12_hash_b.txt 2>> gpurun_out/r3c12_hash.err
diff gpurun_out/r3c12_hash_a.txt gpurun_out/r3c12_hash_b.txt && echo "two processes: BIT-IDENTICAL ($(wc -l < gpurun_out/r3c12_hash_a.txt) digests)"

#!/bin/bash
# NOTE (round 3): the A/B flags this tool toggles are experiments -- build the library with `make -C sipmask_amd/csrc EXPERIMENTS=1` first (csrc/experiments.h); the default build ignores them.
# One-call A/B of the conv K-loop variants on the GPU box: kernel parity, conv micro-benchmark, end-to-end bench per
# variant, then the engine/API parity tests under the fastest variant.  Everything lands in gpurun_out/ab_conv/.
#   FLAT_LOOP 0x00200000 | LEGACY_LOOP 0x00100000 | TILE256 0x00400000 (see include/sipmask_hip.h)
OUT=gpurun_out/ab_conv
mkdir -p $OUT
timeout 60 python -m pytest tests/test_gpu_kernels.py -q -x -k "loader_variants or groupnorm_statistics" > $OUT/parity.txt 2>&1
tail -2 $OUT/parity.txt
if ! grep -q " passed" $OUT/parity.txt || grep -q "failed" $OUT/parity.txt; then echo "PARITY FAILED"; tail -30 $OUT/parity.txt; fi
timeout 60 python tools/conv_bench.py --rounds 5 --iters 10 --only "tower,fpn.out0,cls_cof,l3.conv2,l4.conv2" \
    --variants 0,0x00100000,0x00200000,0x00400000 > $OUT/conv_bench.txt 2>&1
cat $OUT/conv_bench.txt
for f in 0 0x00100000 0x00200000 0x00400000; do
  SIPMASK_CONV_DEBUG_FLAGS=$f timeout 40 python bench.py --no-cpu-baseline > $OUT/bench_$f.txt 2>&1
done
BEST=$(python - <<'PY'
import json, glob
best, bv = "0", 0.0
for p in sorted(glob.glob("gpurun_out/ab_conv/bench_*.txt")):
    try:
        v = json.loads(open(p).read().strip().splitlines()[-1])["value"]
    except Exception:
        v = 0.0
    f = p.split("bench_")[1][:-4]
    print("#", f, v, flush=True)
    if v > bv:
        best, bv = f, v
print(best)
PY
)
echo "$BEST" > $OUT/best.txt
cat $OUT/best.txt
FLAG=$(tail -1 $OUT/best.txt)
if [ "$FLAG" != "0" ]; then
  SIPMASK_CONV_DEBUG_FLAGS=$FLAG timeout 100 python -m pytest tests/test_gpu_engine.py tests/test_gpu_api.py -q -x > $OUT/engine_tests_$FLAG.txt 2>&1
  tail -3 $OUT/engine_tests_$FLAG.txt
fi

#!/bin/bash
mkdir -p gpurun_out/prof3; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof3; cd $R
timeout 600 python -m pytest tests/test_gpu_engine.py tests/test_gpu_baseline_shape.py tests/test_gpu_api.py -x -q -m gpu 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
timeout 400 python $R/bench.py --breakdown $OUT/step_breakdown.txt > $OUT/bench_r50.json 2> $OUT/bench_r50.err
timeout 300 python $R/bench.py --no-cpu-baseline --precision head_x3 > $OUT/bench_r50_x3b.json 2>/dev/null
timeout 300 python $R/bench.py --no-cpu-baseline --config r101 > $OUT/bench_r101b.json 2>/dev/null
for f in r50 r50_x3b r101b; do python - <<PY
import json
d=json.loads([l for l in open("$OUT/bench_$f.json") if l.startswith("{")][-1])
print("$f", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"], d["config"].get("steps_in_flight"), (d.get("parity") or {}).get("common_dets"))
PY
done

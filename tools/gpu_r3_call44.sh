#!/bin/bash
mkdir -p gpurun_out/prof3; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof3; cd /tmp && export TMPDIR=/tmp
timeout 400 python $R/bench.py > $OUT/bench_r50.json 2> $OUT/bench_r50.err
timeout 300 python $R/bench.py --no-cpu-baseline --in-flight 2 > $OUT/bench_r50_inflight2.json 2>/dev/null
timeout 300 python $R/bench.py --no-cpu-baseline --precision head_x3 > $OUT/bench_r50_x3b.json 2>/dev/null
timeout 300 python $R/bench.py --no-cpu-baseline --config r101 > $OUT/bench_r101b.json 2>/dev/null
for f in r50 r50_inflight2 r50_x3b r101b; do python - <<PY
import json
d=json.loads([l for l in open("$OUT/bench_$f.json") if l.startswith("{")][-1])
print("$f", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"], d["config"].get("steps_in_flight"), (d.get("parity") or {}).get("common_dets"))
PY
done

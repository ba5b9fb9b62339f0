#!/bin/bash
mkdir -p gpurun_out/prof3; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof3; cd /tmp && export TMPDIR=/tmp
timeout 400 python $R/bench.py > $OUT/bench_r50.json 2> $OUT/bench_r50.err
python - <<PY
import json
d=json.loads([l for l in open("$OUT/bench_r50.json") if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], d["steps"], d["warmup"], d["roofline"]["frac"], d["roofline"]["traffic"], d["config"]["steps_in_flight"], d["parity"]["common_dets"], d["parity"]["mask_logit_max_abs"], d["cpu_baseline"]["value"])
PY

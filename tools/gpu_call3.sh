#!/bin/bash
set -x
export TMPDIR=/tmp
O=gpurun_out/c3
mkdir -p $O
timeout 600 python bench.py > $O/bench_r50.json 2> $O/bench_r50.err
timeout 600 python bench.py --precision f32 --no-cpu-baseline --breakdown $O/breakdown_f32.txt > $O/bench_r50_f32.json 2> $O/bench_r50_f32.err
timeout 600 python bench.py --config r101 --no-cpu-baseline > $O/bench_r101.json 2> $O/bench_r101.err
timeout 900 python bench.py --config train > $O/bench_train.json 2> $O/bench_train.err
timeout 600 python bench.py --config vis > $O/bench_vis.json 2> $O/bench_vis.err
for f in r50 r50_f32 r101 train vis; do echo "== $f"; cut -c1-600 $O/bench_$f.json; tail -n 3 $O/bench_$f.err; done

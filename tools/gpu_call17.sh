#!/bin/bash
# NOTE (round 3): the A/B flags this tool toggles are experiments -- build the library with `make -C sipmask_amd/csrc EXPERIMENTS=1` first (csrc/experiments.h); the default build ignores them.
# pipelined patch stage (SM_CONV_DBG_PATCH_PIPE): parity, micro-benchmark, whole-step A/B
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/call17
rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_patch_conv.py -x -q -m gpu > $OUT/pytest_patch.log 2>&1
tail -5 $OUT/pytest_patch.log
timeout 300 python tools/patch_bench.py > $OUT/patch_bench.txt 2>&1
cut -c1-330 $OUT/patch_bench.txt
B="timeout 300 python bench.py --no-cpu-baseline --steps 100 --warmup 10"
for rep in 1 2; do
  SIPMASK_CONV_DEBUG_FLAGS=0x4000 $B --breakdown $OUT/bd_uniform_$rep.txt > $OUT/uniform_$rep.json 2>$OUT/uniform_$rep.err
  SIPMASK_CONV_DEBUG_FLAGS=0x4800 $B --breakdown $OUT/bd_pipeuni_$rep.txt > $OUT/pipeuni_$rep.json 2>$OUT/pipeuni_$rep.err
  SIPMASK_CONV_DEBUG_FLAGS=0x800 $B --breakdown $OUT/bd_pipemix_$rep.txt > $OUT/pipemix_$rep.json 2>$OUT/pipemix_$rep.err
done
for f in $OUT/*.json; do echo $(basename $f) $(python -c "import json,sys;l=[x for x in open('$f') if x.startswith('{')];j=json.loads(l[-1]) if l else {};print(j.get('value'),j.get('ms_per_step'),(j.get('roofline') or {}).get('frac'))"); done
grep -E "tower|reg_convs.3|cls_cof|fpn.out0" $OUT/bd_pipeuni_1.txt
grep -E "tower|reg_convs.3|cls_cof|fpn.out0" $OUT/bd_pipemix_1.txt
